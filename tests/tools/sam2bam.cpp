// sam2bam.cpp -- test tool: a SAM file through the command line's BAM writer (soap3-dp_amd/host/s3_bam.cpp), the way `soap3-dp -b 3` uses it:
// header block, records in pieces (whole lines) compressed on their own, end-of-file block.     sam2bam in.sam out.bam [bytes per piece [zlib level]]
#include <stdio.h>
#include <stdlib.h>
#include <fstream>
#include <sstream>
#include "s3_bam.h"
using namespace s3host;

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: sam2bam in.sam out.bam [piece]\n"); return 2; }
    std::ifstream in(argv[1], std::ios::binary);
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string all = ss.str();
    const size_t piece = argc > 3 ? (size_t)atol(argv[3]) : 100000;
    const int level = argc > 4 ? atoi(argv[4]) : -1;
    // header text and reference dictionary from the @SQ lines
    std::string header;
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
    size_t pos = 0;
    while (pos < all.size() && all[pos] == '@') {
        const size_t e = all.find('\n', pos);
        const std::string line = all.substr(pos, e - pos);
        header += line + "\n";
        if (line.compare(0, 3, "@SQ") == 0) {
            const size_t sn = line.find("SN:"), ln = line.find("LN:");
            names.push_back(line.substr(sn + 3, line.find('\t', sn) - sn - 3));
            lens.push_back((uint32_t)strtoul(line.c_str() + ln + 3, nullptr, 10));
        }
        pos = e + 1;
    }
    FILE *out = fopen(argv[2], "wb");
    if (!out) return 1;
    bamRegister(out, names, lens, level);
    if (!bamWriteHeader(out, header)) return 1;
    // half of the body through the buffered single-record path, half as independently compressed pieces (the parallel formatter's way)
    const size_t half = all.find('\n', pos + (all.size() - pos) / 2);
    size_t cur = pos;
    while (cur < all.size() && cur <= half) {
        const size_t e = all.find('\n', cur);
        if (!bamWriteLines(out, all.data() + cur, e + 1 - cur)) return 1;
        cur = e + 1;
    }
    if (!bamFlush(out)) return 1;
    while (cur < all.size()) {
        size_t e = cur + piece < all.size() ? all.find('\n', cur + piece) : std::string::npos;
        e = e == std::string::npos ? all.size() : e + 1;
        std::string raw, z, err;
        if (!samTextToBam(names, all.data() + cur, e - cur, raw, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        if (!bgzfCompress(raw.data(), raw.size(), z, level)) return 1;
        fwrite(z.data(), 1, z.size(), out);
        cur = e;
    }
    if (!bamClose(out)) return 1;
    fclose(out);
    return 0;
}
