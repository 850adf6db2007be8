// chunk_reads -- TEST INFRASTRUCTURE ONLY (built and run by tests/test_read_chunker_cpu.py).
// Dumps what the product's read loader (s3host::ReadChunker) makes of a read file (or a pair of files) when it is asked for chunks of a given size:
// one line per read -- name, length, the read's packed words, codes and qualities as hex.  The test compares the dumps of different chunk sizes.
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include "../../soap3-dp_amd/host/s3_host.h"
using namespace s3host;

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: chunk_reads <chunk reads> <max read length> <out> <file1> [file2]\n"); return 1; }
    const uint32_t chunk = (uint32_t)strtoul(argv[1], nullptr, 10), L = (uint32_t)atoi(argv[2]);
    FILE *out = fopen(argv[3], "w");
    ReadChunker c;
    std::string err;
    // a file argument "bam:<file>" / "bampair:<file>": reads (pairs) from one BAM file
    const std::string f1 = argv[4];
    const bool ok = f1.compare(0, 4, "bam:") == 0 ? c.openBam(f1.substr(4), false, L, 33, 64, err)
                  : f1.compare(0, 8, "bampair:") == 0 ? c.openBam(f1.substr(8), true, L, 33, 64, err) : c.open(f1, argc > 5 ? argv[5] : "", L, 33, 64, err);
    if (!ok) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    ReadSet rs;
    unsigned chunks = 0;
    while (c.next(chunk, rs, err)) {
        chunks++;
        if (rs.n > chunk) { fprintf(stderr, "chunk of %u reads, asked for %u\n", rs.n, chunk); return 1; }
        for (uint32_t r = 0; r < rs.n; r++) {
            fprintf(out, "%s %u %d ", rs.names[r].c_str(), rs.lengths[r], rs.isFastq ? 1 : 0);
            for (uint32_t w = 0; w < rs.wpq; w++) fprintf(out, "%08x", rs.queries[(size_t)(r / 32) * 32 * rs.wpq + r % 32 + (size_t)32 * w]);
            fputc(' ', out);
            for (uint32_t i = 0; i < rs.maxReadLength; i++) fprintf(out, "%x", rs.codes[(size_t)r * rs.maxReadLength + i]);
            fputc(' ', out);
            for (uint32_t i = 0; i < rs.maxReadLength; i++) fprintf(out, "%02x", (unsigned char)rs.quals[(size_t)r * rs.maxReadLength + i]);
            fputc('\n', out);
        }
    }
    if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    fprintf(stderr, "%u chunks\n", chunks);
    fclose(out);
    return 0;
}
