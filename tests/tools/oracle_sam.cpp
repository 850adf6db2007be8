// oracle_sam -- TEST INFRASTRUCTURE ONLY (built and run by tests/test_host_sam_cpu.py).
// The product's host code (soap3-dp_amd/host: index loader, read loader, SAM writer) driven by the CPU ORACLE instead of the device, so
// that the host logic can be checked against the reference's golden SAM without a GPU.  Same options as the product's command line.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../soap3-dp_amd/host/s3_host.h"
extern "C" {
#include "../../oracle/s3_oracle.h"
typedef struct {
    uint8_t  phase, min1, min2, minTotal, secMinTotal, bestS1, bestS2, bestM1, bestM2, pad[3];
    uint32_t x0_1, x1_1, x0_2, x1_2, numPairs, numMin, bestA1, bestA2;
} s3o_pair;
typedef struct { uint32_t a1, a2; uint8_t s1, s2, m1, m2; } s3o_pair_xa;
long s3o_align_pair(const s3o_index *ix, const uint32_t *queries, const uint32_t *readLengths, uint32_t nReads, uint32_t wordPerQuery,
                    int numMismatch, uint32_t maxHitsPerEnd, int reportType, uint32_t insertLow, uint32_t insertHigh, uint32_t peMaxOutputPerPair,
                    s3o_pair *pairs, uint64_t *xaOffsets, s3o_pair_xa *xa, size_t xaCap,
                    uint64_t *occOffsets, uint32_t *occPos, uint8_t *occStrand, uint8_t *occMis, size_t occCap, s3o_stats *st);
}
using namespace s3host;

static int die(const std::string &m) { fprintf(stderr, "%s\n", m.c_str()); return 1; }

int main(int argc, char **argv)
{
    if (argc < 4) return die("usage: oracle_sam single|pair <index> <reads...> [options]");
    const bool paired = strcmp(argv[1], "pair") == 0;
    const std::string indexName = argv[2], q1 = argv[3], q2 = paired ? argv[4] : "";
    int maxReadLength = 120, numMismatch = 3, type = 2, insertHigh = 500, insertLow = 1;
    unsigned maxOut = 1000, maxHits = 8000, peMax = 1000;
    bool printMDNM = false;
    std::string prefix = "out";
    for (int i = paired ? 5 : 4; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-L") maxReadLength = atoi(argv[++i]);
        else if (a == "-h") type = atoi(argv[++i]);
        else if (a == "-s") numMismatch = atoi(argv[++i]);
        else if (a == "-o") prefix = argv[++i];
        else if (a == "-u") insertHigh = atoi(argv[++i]);
        else if (a == "-v") insertLow = atoi(argv[++i]);
        else if (a == "-b") ++i;
        else if (a == "-p") printMDNM = true;
        else return die("unknown option " + a);
    }
    std::string err;
    HostIndex hix;
    if (!loadIndex(indexName, hix, err)) return die(err);
    const s3_index_desc d = hix.desc();
    s3o_index *oix = s3o_index_create(d.textLength, d.fwd.bwt, d.fwd.bwtWords, d.fwd.occ, d.fwd.inverseSa0, d.fwd.cumFreq, d.rev.bwt, d.rev.bwtWords, d.rev.occ,
                                      d.rev.inverseSa0, d.rev.cumFreq, d.sa, d.packedDNA, d.packedWords, d.lkt, d.revLkt, d.lktSize);
    if (!oix) return die("oracle index");
    ReadSet rs;
    if (paired ? !loadPairReads(q1, q2, (uint32_t)maxReadLength, 33, 128, rs, err) : !loadSingleReads(q1, (uint32_t)maxReadLength, 33, 128, rs, err)) return die(err);
    FILE *out = fopen((prefix + ".gout.1").c_str(), "w");
    SamOptions so;
    so.readGroup = q1; so.sampleName = "default"; so.alignmentType = type; so.minMAPQ = 1; so.maxMAPQ = 40; so.bwaLikeScore = 1; so.printMDNM = printMDNM;
    SamWriter sam(hix, so, out);
    sam.header();
    const uint32_t n = rs.n;
    if (paired) {
        std::vector<s3o_pair> pr(n / 2 + 1);
        std::vector<uint64_t> xoff(n / 2 + 1), ooff((size_t)n + 1);
        size_t cap = (size_t)n * 64 + 4096;
        std::vector<s3o_pair_xa> xa;
        std::vector<uint32_t> pos;
        std::vector<uint8_t> strand, mis;
        for (;;) {
            xa.resize(cap); pos.resize(cap); strand.resize(cap); mis.resize(cap);
            if (s3o_align_pair(oix, rs.queries.data(), rs.lengths.data(), n, rs.wpq, numMismatch, maxHits, type, (uint32_t)insertLow, (uint32_t)insertHigh, peMax,
                               pr.data(), xoff.data(), xa.data(), cap, ooff.data(), pos.data(), strand.data(), mis.data(), cap, nullptr) == 0) break;
            cap *= 4;
        }
        static_assert(sizeof(s3o_pair) == sizeof(s3_pair_result) && sizeof(s3o_pair_xa) == sizeof(s3_pair_xa), "layouts");
        for (uint32_t p = 0; p < n / 2; p++) {
            if (pr[p].phase) sam.properPair(rs, 2 * p, 2 * p + 1, *(const s3_pair_result *)&pr[p], (const s3_pair_xa *)xa.data() + xoff[p], (size_t)(xoff[p + 1] - xoff[p]));
            else {
                const size_t o1 = ooff[2 * p], n1 = ooff[2 * p + 1] - o1, o2 = ooff[2 * p + 1], n2 = ooff[2 * p + 2] - o2;
                sam.improperPair(rs, 2 * p, 2 * p + 1, pos.data() + o1, strand.data() + o1, mis.data() + o1, n1, pos.data() + o2, strand.data() + o2, mis.data() + o2, n2, maxHits);
            }
        }
    } else {
        std::vector<uint32_t> seeds;
        if (maxReadLength > 120) { seeds.resize(n); for (uint32_t k = 0; k < n; k++) seeds[k] = rs.lengths[k] > 120 ? 100u : rs.lengths[k]; }
        std::vector<uint8_t> phase(n + 1);
        std::vector<uint32_t> we((size_t)n * 8 + 8);
        std::vector<uint64_t> offs((size_t)n + 1);
        size_t cap = (size_t)n * 64 + 4096;
        std::vector<uint32_t> pos;
        std::vector<uint8_t> strand, mis;
        for (;;) {
            pos.resize(cap); strand.resize(cap); mis.resize(cap);
            if (s3o_align_single_long(oix, rs.queries.data(), rs.lengths.data(), seeds.empty() ? nullptr : seeds.data(), n, rs.wpq, numMismatch, maxOut, type, phase.data(),
                                      we.data(), offs.data(), pos.data(), strand.data(), mis.data(), cap, nullptr) >= 0) break;
            cap *= 4;
        }
        for (uint32_t k = 0; k < n; k++) {
            const size_t o = offs[k], m = offs[k + 1] - offs[k];
            if (type >= 3) {
                if (m == 0 || (type == 3 && m > 1)) sam.unaligned(rs, k);
                else sam.singleAnswer(rs, k, pos[o], strand[o], mis[o]);
            } else sam.singleEnd(rs, k, pos.data() + o, strand.data() + o, mis.data() + o, m, seeds.empty() ? 0u : seeds[k]);
        }
    }
    fclose(out);
    s3o_index_free(oix);
    return 0;
}
