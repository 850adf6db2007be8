// dp_pair_sam -- TEST INFRASTRUCTURE ONLY (built and run by tests/test_host_dp_sam_cpu.py).
// The product's host SAM writer for the paired DP stages driven by result vectors (formats: tools/simulate_dp_pair_results.py), so that the
// output stage can be checked against the reference's own outputRead2 / outputDeepDPResult2 without a GPU.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <map>
#include <string>
#include <vector>
#include "../../soap3-dp_amd/host/s3_host.h"
using namespace s3host;

int main(int argc, char **argv)
{
    if (argc == 2 && std::string(argv[1]) == "--dedup") {
        // stdin: one deep-DP result per line, "pos1 tag"; stdout: the tags keepFirstResultPerPosition leaves, in its order
        std::vector<DeepDpHit> hits;
        unsigned pos; int tag;
        while (scanf("%u %d", &pos, &tag) == 2) { DeepDpHit h{}; h.pos1 = pos; h.score1 = tag; hits.push_back(h); }
        keepFirstResultPerPosition(hits);
        for (const DeepDpHit &h : hits) printf("%u %d\n", h.pos1, h.score1);
        return 0;
    }
    if (argc < 7) { fprintf(stderr, "usage: dp_pair_sam <index> <reads1> <reads2> <results> <stats> <out prefix> [-L n] [-h n] [-p] [-B 0|1]\n"); return 1; }
    int maxReadLength = 120, type = 2, bwaLike = 1;
    bool printMDNM = false;
    for (int i = 7; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-L") maxReadLength = atoi(argv[++i]);
        else if (a == "-h") type = atoi(argv[++i]);
        else if (a == "-p") printMDNM = true;
        else if (a == "-B") bwaLike = atoi(argv[++i]);
    }
    std::string err;
    HostIndex hix;
    if (!loadIndex(argv[1], hix, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    ReadSet rs;
    if (!loadPairReads(argv[2], argv[3], (uint32_t)maxReadLength, 33, 128, rs, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    FILE *out = fopen((std::string(argv[6]) + ".dpout.1").c_str(), "w");
    SamOptions so;
    so.readGroup = argv[2]; so.sampleName = "default"; so.alignmentType = type; so.minMAPQ = 1; so.maxMAPQ = 40; so.bwaLikeScore = bwaLike; so.printMDNM = printMDNM;
    SamWriter sam(hix, so, out);
    sam.header();
    std::map<unsigned, BwtEndStats> stats;
    FILE *f = fopen(argv[5], "r");
    if (!f) return 1;
    unsigned rid;
    int x0, x1, mm;
    while (fscanf(f, "%u %d %d %d", &rid, &x0, &x1, &mm) == 4) { BwtEndStats s; s.x0 = x0; s.x1 = x1; s.mismatch = mm; stats[rid] = s; }
    fclose(f);
    f = fopen(argv[4], "r");
    if (!f) return 1;
    std::vector<PairDpHit> dh;
    std::vector<DeepDpHit> ph;
    std::map<unsigned, std::vector<DpHit>> uh;                 // leftover pairs: hits per read
    long cur = -1;
    auto flush = [&]() {
        if (cur < 0) return;
        BwtEndStats st[2] = {stats.count((unsigned)cur) ? stats[(unsigned)cur] : BwtEndStats(), stats.count((unsigned)cur + 1) ? stats[(unsigned)cur + 1] : BwtEndStats()};
        if (!dh.empty()) sam.pairDP(rs, (uint32_t)cur, dh.data(), dh.size(), st);
        if (!ph.empty()) sam.pairDeepDP(rs, (uint32_t)cur, ph.data(), ph.size(), st);
        dh.clear(); ph.clear();
    };
    char kind[8], c1[4096], c2[4096];
    while (fscanf(f, "%7s", kind) == 1) {
        if (kind[0] == 'U') {
            unsigned pos;
            int fromDP, s1, score, ed;
            if (fscanf(f, "%u %d %u %d %d %d %4095s", &rid, &fromDP, &pos, &s1, &score, &ed, c1) != 7) return 1;
            std::vector<DpHit> &v = uh[rid];
            if (pos != 0xFFFFFFFFu) { DpHit h{pos, (uint8_t)s1, score, ed, c1}; h.fromDP = fromDP != 0; v.push_back(h); }
        } else if (kind[0] == 'D') {
            PairDpHit h;
            int which, s1, s2;
            if (fscanf(f, "%u %d %u %d %d %u %d %d %d %d %d %4095s", &rid, &which, &h.pos1, &s1, &h.score1, &h.pos2, &s2, &h.score2, &h.editdist, &h.insertSize, &h.numSameScore, c1) != 12) return 1;
            h.whichFromDP = (uint8_t)which; h.strand1 = (uint8_t)s1; h.strand2 = (uint8_t)s2; h.cigar = c1[0] == '*' ? "" : c1;
            if ((long)rid != cur || !ph.empty()) { flush(); cur = rid; }
            dh.push_back(h);
        } else {
            DeepDpHit h;
            int s1, s2;
            if (fscanf(f, "%u %u %d %d %d %d %4095s %u %d %d %d %d %4095s %d", &rid, &h.pos1, &s1, &h.score1, &h.editdist1, &h.numSame1, c1, &h.pos2, &s2, &h.score2, &h.editdist2,
                       &h.numSame2, c2, &h.insertSize) != 14) return 1;
            h.strand1 = (uint8_t)s1; h.strand2 = (uint8_t)s2; h.cigar1 = c1[0] == '*' ? "" : c1; h.cigar2 = c2[0] == '*' ? "" : c2;
            if ((long)rid != cur || !dh.empty()) { flush(); cur = rid; }
            ph.push_back(h);
        }
    }
    flush();
    fclose(f);
    // cutoff of the single-read DP stage on the detected read length (first ten reads), as tests/tools/dp_sam.cpp
    uint32_t detected = 0;
    for (uint32_t i = 0; i < rs.n && i < 10; i++) detected = detected > rs.lengths[i] ? detected : rs.lengths[i];
    const int cutoff = (int)ceil(0.3 * (double)detected);
    for (auto it = uh.begin(); it != uh.end(); ++it) {
        if (it->first & 1u) continue;
        const std::vector<DpHit> &a = it->second, &b = uh[it->first + 1];
        sam.unpairedDP(rs, it->first, a.data(), a.size(), b.data(), b.size(), cutoff);
    }
    fclose(out);
    return 0;
}
