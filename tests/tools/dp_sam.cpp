// dp_sam -- TEST INFRASTRUCTURE ONLY (built and run by tests/test_host_dp_sam_cpu.py).
// The product's host SAM writer for DP results driven by a result file (readID strand algnmt score editdist num_sameScore cigar), so that
// the output stage can be checked against the reference's own outputDPSingleResult2 without a GPU.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>
#include "../../soap3-dp_amd/host/s3_host.h"
using namespace s3host;

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: dp_sam <index> <reads> <results> <out prefix> [-L n] [-h n] [-p]\n"); return 1; }
    int maxReadLength = 120, type = 2;
    bool printMDNM = false;
    for (int i = 5; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-L") maxReadLength = atoi(argv[++i]);
        else if (a == "-h") type = atoi(argv[++i]);
        else if (a == "-p") printMDNM = true;
    }
    std::string err;
    HostIndex hix;
    if (!loadIndex(argv[1], hix, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    ReadSet rs;
    if (!loadSingleReads(argv[2], (uint32_t)maxReadLength, 33, 128, rs, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    FILE *out = fopen((std::string(argv[4]) + ".dpout.1").c_str(), "w");
    SamOptions so;
    so.readGroup = argv[2]; so.sampleName = "default"; so.alignmentType = type; so.minMAPQ = 1; so.maxMAPQ = 40; so.bwaLikeScore = 1; so.printMDNM = printMDNM;
    SamWriter sam(hix, so, out);
    sam.header();
    // cutoff of the single-read DP stage: 0.3 x the detected read length (first ten reads), alignment.cu:2580-2582
    uint32_t detected = 0;
    for (uint32_t i = 0; i < rs.n && i < 10; i++) detected = detected > rs.lengths[i] ? detected : rs.lengths[i];
    const int cutoff = (int)ceil(0.3 * (double)detected);
    FILE *f = fopen(argv[3], "r");
    if (!f) return 1;
    unsigned readID, algnmt;
    int strand, score, editdist, same;
    char cigar[4096];
    std::vector<DpHit> hits;
    long cur = -1;
    auto flush = [&]() { if (cur >= 0) sam.singleDP(rs, (uint32_t)cur, hits.data(), hits.size(), cutoff); hits.clear(); };
    while (fscanf(f, "%u %d %u %d %d %d %4095s", &readID, &strand, &algnmt, &score, &editdist, &same, cigar) == 7) {
        if ((long)readID != cur) { flush(); cur = readID; }
        hits.push_back(DpHit{algnmt, (uint8_t)strand, score, editdist, cigar[0] == '*' ? std::string() : std::string(cigar)});
    }
    flush();
    fclose(f);
    fclose(out);
    return 0;
}
