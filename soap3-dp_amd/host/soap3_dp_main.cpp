// soap3-dp -- drop-in command line for the alignment path (single-end and paired-end, DP stages on or off), same arguments, .ini keys and index files as
// the reference program (SOAP3-DP.cu main, IniParam.cpp:237-767), driving the MI355X kernels through the C-ABI of libs3dp.so.
//
//   soap3-dp single <index prefix> <reads.fa|fq[.gz]> [-L maxReadLen] [-s k] [-h 1..4] [-b 2|3] [-bam] [-o prefix] [-c gpu] [-D rg] [-A sample] [-R opts] [-p]
//   soap3-dp pair   <index prefix> <reads 1> <reads 2> [-u maxInsert] [-v minInsert] [-s k] [the same options]
// Without -s the DP stages are on (the reference's default): single reads the BWT stage leaves unaligned go through seeding + semi-global DP
// and are written to <prefix>.dpout.1; pairs without a proper pairing go through the default DP (mate rescue next to an aligned end) or the deep
// DP (both ends seeded) into <prefix>.dpout.1, the rest into <prefix>.unpair (pairDpStages below says which reference stages are missing).
//
// Output: <prefix>.gout.1 (SAM), <prefix>.dpout.1, <prefix>.unpair, <prefix>.done, like SOAP3-DP.cu:251-258 with NumOfCpuThreads = 1.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <unistd.h>
#include <sched.h>
#include <thread>
#include <future>
#include <vector>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <memory>
#include "s3_host.h"
#include "s3_bam.h"

using namespace s3host;

struct Ini {
    unsigned maxOutputPerRead = 1000, peMaxOutputPerPair = 1000, maxHitsEachEndForPairing = 10000;
    int minMAPQ = 1, maxMAPQ = 40, bwaLike = 0, maxLenReadName = 64;     // code defaults of IniParam.cpp:237-362
    int matchScore = 1, mismatchScore = -2, gapOpenScore = -3, gapExtendScore = -1;
    int soap3MisMatchAllow = 2, dpScoreThreshold = -1 /* DEFAULT */, maxFrontLenClipped = 3, maxEndLenClipped = 8;   // IniParam.cpp:300-362
};

static void parseIni(const std::string &file, Ini &ini)
{
    FILE *f = fopen(file.c_str(), "r");
    if (!f) return;
    char line[1024];
    while (fgets(line, sizeof line, f)) {
        char key[256], val[256];
        if (line[0] == '#' || line[0] == '[') continue;
        if (sscanf(line, " %255[^= ] = %255s", key, val) != 2) continue;
        if (!strcmp(key, "MaxOutputPerRead")) ini.maxOutputPerRead = (unsigned)strtoul(val, nullptr, 10);
        else if (!strcmp(key, "MinMAPQ")) ini.minMAPQ = atoi(val);
        else if (!strcmp(key, "MaxMAPQ")) ini.maxMAPQ = atoi(val);
        else if (!strcmp(key, "BWALikeScore")) ini.bwaLike = atoi(val);
        else if (!strcmp(key, "MaxLenReadName")) ini.maxLenReadName = atoi(val);
        else if (!strcmp(key, "MaxOutputPerPair")) ini.peMaxOutputPerPair = (unsigned)strtoul(val, nullptr, 10);
        else if (!strcmp(key, "MaxHitsEachEndForPairing")) ini.maxHitsEachEndForPairing = (unsigned)strtoul(val, nullptr, 10);
        else if (!strcmp(key, "MatchScore")) ini.matchScore = atoi(val);
        else if (!strcmp(key, "MismatchScore")) ini.mismatchScore = atoi(val);
        else if (!strcmp(key, "GapOpenScore")) ini.gapOpenScore = atoi(val);
        else if (!strcmp(key, "GapExtendScore")) ini.gapExtendScore = atoi(val);
        else if (!strcmp(key, "Soap3MisMatchAllow")) ini.soap3MisMatchAllow = atoi(val);
        else if (!strcmp(key, "DPScoreThreshold")) ini.dpScoreThreshold = strcmp(val, "DEFAULT") == 0 ? -1 : atoi(val);
        else if (!strcmp(key, "MaxFrontLenClipped")) ini.maxFrontLenClipped = atoi(val);
        else if (!strcmp(key, "MaxEndLenClipped")) ini.maxEndLenClipped = atoi(val);
    }
    fclose(f);
}

// SAM records of `count` units (reads or pairs) formatted by all host threads, each into its own buffer, written in unit order.  The buffers of a
// call go to the file in the background -- a few pwrite streams at their final offsets -- while the next call formats into a second set: formatting
// 4 Mi reads takes all host threads about half a second and writing their 1.3 GB about as long, and the two used to take turns.  Per output file: two
// generations of buffers and the write that is still under way (drained before the file is closed: drainOutput).
// (one cache line pair per buffer: the headers of neighbouring std::strings share a line, every append updates a length, and sixteen formatting threads
//  appending a dozen fields per record to neighbouring strings ran at a quarter of their speed -- 0.28 us per record per thread with the padding, 1.24 without)
struct alignas(128) OutBuffer { std::string s; };
struct OutState {
    std::mutex use;                                   // one formatting call (or direct writer) per file at a time: two DP workers share .dpout / .unpair
    std::vector<OutBuffer> buf[2];
    std::thread pending[2];
    int gen = 0;
};
static std::mutex g_outMu;
static std::map<FILE *, std::unique_ptr<OutState>> g_outStates;
static OutState &outState(FILE *f)
{
    std::lock_guard<std::mutex> g(g_outMu);
    std::unique_ptr<OutState> &p = g_outStates[f];
    if (!p) p.reset(new OutState());
    return *p;
}
static void drainOutput(FILE *f)                                              // before anything else touches the file, and before it is closed
{
    std::unique_ptr<OutState> st;
    {
        std::lock_guard<std::mutex> g(g_outMu);
        auto it = g_outStates.find(f);
        if (it == g_outStates.end()) return;
        st = std::move(it->second);
        g_outStates.erase(it);
    }
    for (std::thread &t : st->pending) if (t.joinable()) t.join();
}

// where a call of parallelFormat spends its time (summed over the run, printed with S3DP_VERBOSE): waiting for the write that still holds its buffers,
// the formatting threads (wall, and the sum over the threads), the rest
static std::mutex g_fmtMu;
static double g_fmtSeconds[4] = {0, 0, 0, 0};

template <class F>
static void parallelFormat(const HostIndex &hix, const SamOptions &so, FILE *out, uint32_t count, F &&one, uint64_t &aligned)
{
    const auto tEnter = std::chrono::steady_clock::now();
    auto secondsSince = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    const int T = hostThreads();
    const uint32_t per = (count + (uint32_t)T - 1) / (uint32_t)T;
    // the buffers live as long as the output file: a chunk of 4 Mi reads is more than a gigabyte of SAM, and touching that much fresh memory from all
    // host threads at once for every chunk is page faults on one address space's lock
    OutState &os = outState(out);
    std::lock_guard<std::mutex> oneAtATime(os.use);
    const int gen = os.gen;
    os.gen ^= 1;
    if (os.pending[gen].joinable()) os.pending[gen].join();                   // the write of the call before the last one used these buffers
    const double waited = secondsSince(tEnter);
    const auto tWork = std::chrono::steady_clock::now();
    std::vector<double> busy((size_t)T, 0);
    std::vector<OutBuffer> &buf = os.buf[gen];
    if (buf.size() < (size_t)T) { std::vector<OutBuffer> bigger((size_t)T); for (size_t i = 0; i < buf.size(); i++) bigger[i].s.swap(buf[i].s); buf.swap(bigger); }
    for (auto &b : buf) b.s.clear();
    std::vector<uint64_t> al((size_t)T, 0);
    const BamFile *bam = bamFileOf(out);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t] {
        const uint32_t lo = (uint32_t)t * per, hi = lo + per < count ? lo + per : count;
        if (lo >= hi) return;
        const auto tMine = std::chrono::steady_clock::now();
        struct Stop { double &d; std::chrono::steady_clock::time_point t0; ~Stop() { d = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } stop{busy[(size_t)t], tMine};
        std::string &mine = buf[(size_t)t].s;
        mine.reserve((size_t)(hi - lo) * 320);
        SamWriter w(hix, so, &mine);
        uint64_t alignedHere = 0;                                            // (not al[t] in the loop: the counters of eight threads share a cache line)
        for (uint32_t k = lo; k < hi; k++) one(w, k, alignedHere);
        al[(size_t)t] = alignedHere;
        if (bam) {
            // -b 3: this thread's records as BAM, in BGZF blocks of their own (blocks are independent: the parts are written behind each other)
            std::string raw, err;
            if (!samTextToBam(bam->refNames, mine.data(), mine.size(), raw, err)) { fprintf(stderr, "BAM output: %s\n", err.c_str()); fflush(stdout); fflush(stderr); _exit(1); }
            mine.clear();
            if (!bgzfCompress(raw.data(), raw.size(), mine, bam->level)) { fprintf(stderr, "BAM output: compression failed\n"); fflush(stdout); fflush(stderr); _exit(1); }
        }
    });
    for (auto &x : th) x.join();
    {
        double sum = 0;
        for (double b : busy) sum += b;
        std::lock_guard<std::mutex> g(g_fmtMu);
        g_fmtSeconds[0] += waited; g_fmtSeconds[1] += secondsSince(tWork); g_fmtSeconds[2] += sum; g_fmtSeconds[3] += 1;
    }
    // the buffers go to the file at their final offsets, a few writers at a time; the file position moves behind them at once, so whatever is
    // written through the FILE next (records of stages that do not come through here) lands behind this call's records
    if (!bamFlush(out)) { fprintf(stderr, "write error on the output file\n"); fflush(stdout); fflush(stderr); _exit(1); }
    fflush(out);
    const int fd = fileno(out);
    std::vector<long long> off((size_t)T + 1, 0);
    off[0] = (long long)ftello(out);
    for (int t = 0; t < T; t++) { off[(size_t)t + 1] = off[(size_t)t] + (long long)buf[(size_t)t].s.size(); aligned += al[(size_t)t]; }
    fseeko(out, (off_t)off[(size_t)T], SEEK_SET);
    os.pending[gen] = std::thread([fd, T, off, &buf] {
        static const int maxWriters = getenv("S3DP_WRITE_THREADS") && atoi(getenv("S3DP_WRITE_THREADS")) > 0 ? atoi(getenv("S3DP_WRITE_THREADS")) : 8;
        const int W = T < maxWriters ? T : maxWriters;
        std::vector<std::thread> wr;
        std::atomic<bool> bad(false);
        for (int w = 0; w < W; w++) wr.emplace_back([&, w] {
            for (int t = w; t < T; t += W) {
                const std::string &b = buf[(size_t)t].s;
                size_t done = 0;
                while (done < b.size()) {
                    const ssize_t k = pwrite(fd, b.data() + done, b.size() - done, (off_t)(off[(size_t)t] + (long long)done));
                    if (k <= 0) { bad = true; return; }
                    done += (size_t)k;
                }
            }
        });
        for (auto &x : wr) x.join();
        if (bad) { fprintf(stderr, "write error on the output file\n"); fflush(stdout); fflush(stderr); _exit(1); }
    });
}

// ---- single-end DP of a list of reads (DPForUnalignSingle2, DV-DPForSingleReads.cu): seeding in batches of at most 256 Ki seeds, DP at the candidates;
// `emit` gets one call per run of results of the same read (DPSOutputThread, DV-DPfunctions.cu:1776 -- results arrive in (read, position) order, both
// strands of a read in one run: s3dp_seed.hip, singleMergeKernel) after OutputBuffer::ready (DV-DPfunctions.h:367-412) with the DP module's type
// (all-best + SAM runs it as all-valid, alignment.cu:2563): sorted by position, one result per position; then best-only for -h 3 / 4 --------------
// parallelOut != NULL: the runs of a batch are handed to `emit` by all host threads at once, each with a writer whose records land in parallelOut in run
// order; NULL: one call after the other (the writer argument is then a dummy that must not be used).
// number of seeds of a read of a given length in a DP stage (s3_seed_positions, asked per read when the seed batches are cut): the answer depends on the
// stage and the length only, and a chunk asks a hundred thousand times
static int seedCountOf(int stage, int readLength)
{
    thread_local int cachedStage = -1, cachedLength = -1, cachedCount = 0;
    if (stage != cachedStage || readLength != cachedLength) {
        int sl = 0, sp[128];
        cachedCount = 0;
        s3_seed_positions(stage, readLength, &sl, sp, &cachedCount);
        cachedStage = stage; cachedLength = readLength;
    }
    return cachedCount;
}

template <class Emit>
static bool singleDpRuns(const HostIndex &hix, const SamOptions &so, FILE *parallelOut, s3_batch *b, s3_batch *seedBatch, s3_dp *dp, const uint32_t *lens,
                         const std::vector<uint32_t> &todo, const s3_dp_stage_params &dpPar, int alignmentType, Emit &&emit, std::string &err)
{
    size_t at = 0;
    while (at < todo.size()) {
        size_t end = at, nSeeds = 0;
        while (end < todo.size()) {
            const int num = seedCountOf(S3_STAGE_SINGLE_DP, (int)lens[todo[end]]);
            if (nSeeds + (size_t)num > 262144) break;
            nSeeds += (size_t)num;
            end++;
        }
        if (end == at) end = at + 1;
        s3_seed_candidate *cand = nullptr;
        size_t nCand = 0;
        if (s3_seed_candidates(b, seedBatch, lens, todo.data() + at, end - at, S3_STAGE_SINGLE_DP, (uint32_t)dpPar.maxHitNum, &cand, &nCand, nullptr, nullptr) != S3_OK) {
            err = s3_last_error();
            return false;
        }
        s3_dp_single_result *res = nullptr;
        size_t nRes = 0;
        char *pool = nullptr;
        if (s3_dp_single_end(dp, b, lens, cand, nCand, dpPar.cutoffThreshold, dpPar.softClipLeft, dpPar.softClipRight, &res, &nRes, &pool) != S3_OK) {
            err = s3_last_error();
            s3_free(cand);
            return false;
        }
        std::vector<std::pair<size_t, size_t>> groups;
        for (size_t i = 0; i < nRes;) {
            size_t j = i;
            while (j < nRes && res[j].readID == res[i].readID) j++;
            groups.push_back({i, j});
            i = j;
        }
        auto one = [&](SamWriter &w, uint32_t gk) {
            const size_t i = groups[gk].first, j = groups[gk].second;
            std::vector<DpHit> hits;
            for (size_t q = i; q < j; q++) hits.push_back(DpHit{res[q].algnmt, res[q].strand, res[q].score, (int)res[q].editdist, std::string(pool + res[q].cigarOffset)});
            std::sort(hits.begin(), hits.end(), [](const DpHit &x, const DpHit &y) { return x.pos < y.pos; });
            size_t u = 0;
            for (size_t q = 1; q < hits.size(); q++) if (hits[u].pos < hits[q].pos) hits[++u] = hits[q];
            hits.resize(hits.empty() ? 0 : u + 1);
            if (alignmentType >= 3 && !hits.empty()) {
                std::vector<DpHit> bestOnes;
                int bestScore = hits[0].score;
                for (const DpHit &hh : hits) bestScore = std::max(bestScore, hh.score);
                for (const DpHit &hh : hits) if (hh.score == bestScore) bestOnes.push_back(hh);
                if (alignmentType == 3 && bestOnes.size() != 1) bestOnes.clear();
                if (alignmentType == 4) bestOnes.resize(1);
                hits.swap(bestOnes);
            }
            if (!hits.empty()) emit(w, res[i].readID, hits);
        };
        if (parallelOut) {
            uint64_t unused = 0;
            parallelFormat(hix, so, parallelOut, (uint32_t)groups.size(), [&](SamWriter &w, uint32_t gk, uint64_t &) { one(w, gk); }, unused);
        } else {
            std::string sink;
            SamWriter dummy(hix, so, &sink);
            for (uint32_t gk = 0; gk < groups.size(); gk++) one(dummy, gk);
        }
        s3_free(cand); s3_free(res); s3_free(pool);
        at = end;
    }
    return true;
}

static std::mutex g_pairStageMu;                                 // (several devices run the stages at once)
static double g_pairStageSeconds[5] = {0, 0, 0, 0, 0};          // routing, default DP, new default DP, deep DP, single-end DP of the leftovers
static double g_chunkSeconds[6] = {0, 0, 0, 0, 0, 0};              // wall clock inside the chunk loop: alignment call, fetch of the results, formatting + writing, DP stages (on their own thread), waiting for them, upload
static void addChunkSeconds(int k, double sec) { std::lock_guard<std::mutex> g(g_pairStageMu); g_chunkSeconds[k] += sec; }
static double g_streamSeconds[6] = {0, 0, 0, 0, 0, 0};             // streamed single reads: waiting for the loader, for a free slot, copying a chunk in; writer: waiting for results, formatting + writing
static void addStreamSeconds(int k, double sec) { std::lock_guard<std::mutex> g(g_pairStageMu); g_streamSeconds[k] += sec; }
static double g_deviceMs[4] = {0, 0, 0, 0};                        // device time: BWT stages (ladder), their search kernels; DP kernels; (spare)
static uint64_t g_dpProblems = 0, g_dpCells = 0;
static void addDeviceMs(int k, double ms) { std::lock_guard<std::mutex> g(g_pairStageMu); g_deviceMs[k] += ms; }
static void addBatchTiming(s3_batch *b)
{
    float tot = 0, search = 0; uint32_t launches = 0;
    if (s3_batch_last_timing(b, &tot, &search, &launches) == S3_OK) { addDeviceMs(0, tot); addDeviceMs(1, search); }
}
static void addDpTotals(s3_dp *h)
{
    double ms = 0; uint64_t pr = 0, cells = 0;
    if (h && s3_dp_totals(h, &ms, &pr, &cells) == S3_OK) { std::lock_guard<std::mutex> g(g_pairStageMu); g_deviceMs[2] += ms; g_dpProblems += pr; g_dpCells += cells; }
}
static double pairStageSeconds(int k) { std::lock_guard<std::mutex> g(g_pairStageMu); return g_pairStageSeconds[k]; }

// ---- DP stages of a paired-end batch (soap3_dp_pair_align after the BWT ladder, alignment.cu:2021-2420) --------------------------------------
struct PairDpCtx {
    const HostIndex &hix; const SamOptions &so; const ReadSet &rs; const Ini &ini;
    s3_batch *b, *seedBatch; s3_dp *dp, *dpHalf; FILE *dpOut, *unpairOut;
    uint32_t first, cnt; int alignmentType, insertLow, insertHigh;
    s3_pair_routing routing; uint64_t chunkNo, readBase;      // of the chunk: the device-side routing's counts; its number and its first read in the input (routing dump)
    const s3_dp_stage_params *halfPar; const s3_dp_stage_params (*deepPar)[2]; const s3_dp_stage_params &singlePar; const s3_dp_stage_params *newPar;
};

// Routing of the pairs the ladder left without a proper pairing (hostKernel with needProceedDP, CPUfunctions.cpp:2153-2262, 2444-2543), the default
// DP on the half-aligned ones (semiGlobalDP2 -> HalfEndAlignmentEngine -> DPOutputThread -> outputRead2), the deep DP on the rest
// (DPForUnalignPairs2: seeding round 1, round 2 for the pairs that only lost seeds to the hit limit, PairEndAlignmentEngine ->
// outputDeepDPResult2), and for the pairs every stage failed on the single-end DP of their ends (DPForUnalignSingle2 + outputSingleResultForPairEnds)
// into <prefix>.unpair; pairs with an end aligned to too many places for the default DP go through the new default DP first (newSemiGlobalDP).
static bool pairDpStages(const PairDpCtx &c, const std::vector<s3_pair_result> &pr, uint64_t &defaultDpPairs,
                         uint64_t &newDefaultPairs, uint64_t &deepDpPairs, uint64_t &leftoverPairs, std::string &err)
{
    const uint32_t np = c.cnt / 2;
    auto tStage = std::chrono::steady_clock::now();
    auto lap = [&](int k) {
        const auto now = std::chrono::steady_clock::now();
        { std::lock_guard<std::mutex> g(g_pairStageMu); g_pairStageSeconds[k] += std::chrono::duration<double>(now - tStage).count(); }
        tStage = now;
    };
    const int moduleType = c.alignmentType == 2 ? 1 : c.alignmentType;          // all-best + SAM runs the DP module as all-valid (alignment.cu:2023-2030)
    const uint32_t *lens = c.rs.lengths.data() + c.first;
    // The routing itself (which list of which end goes to which stage, cut at the stage's hit limit) ran on the device behind the ladder
    // (s3_batch_route_pairs, called by the chunk loop): here come one byte per pair and the default DP's list; the new default DP's list -- the thousands of
    // hits of ends inside repeat families -- stays where it is, the kept lists of the few pairs every stage fails on are fetched at the end.
    std::vector<uint8_t> route(np ? np : 1, 0);
    std::vector<s3_occ> dpOcc((size_t)c.routing.nDefault);
    if (s3_batch_fetch_routing(c.b, route.data(), dpOcc.data(), dpOcc.size(), nullptr, 0, nullptr, 0) != S3_OK) { err = s3_last_error(); return false; }
    std::vector<uint32_t> deepPairs;                                             // first read (batch index) of the pairs for the deep DP
    for (uint32_t p = 0; p < np; p++) if (route[p] & S3_ROUTE_DEEP) deepPairs.push_back(2 * p);
    if (const char *dump = getenv("S3DP_DUMP_ROUTING")) {                        // the routing, for tests/test_gpu_cli_sam.py (format: oracle/ref_harness.cpp)
        std::vector<s3_occ> newOcc((size_t)c.routing.nNewDefault), otherOcc((size_t)c.routing.nOther);
        if (s3_batch_fetch_routing(c.b, nullptr, nullptr, 0, newOcc.data(), newOcc.size(), otherOcc.data(), otherOcc.size()) != S3_OK) { err = s3_last_error(); return false; }
        FILE *f = fopen(dump, c.first || c.chunkNo ? "a" : "w");
        if (f) {
            const std::vector<s3_occ> *lists[3] = {&dpOcc, &newOcc, &otherOcc};
            for (int l = 0; l < 3; l++)
                for (const s3_occ &x : *lists[l]) fprintf(f, "%c %llu %d %d %u\n", "DNO"[l], (unsigned long long)(c.readBase + x.readID), (int)x.strand, (int)x.mismatchCount, x.pos);
            for (uint32_t r1 : deepPairs) fprintf(f, "U %llu\n", (unsigned long long)(c.readBase + r1));
            fclose(f);
        }
    }
    auto statsOf = [&](uint32_t p, BwtEndStats st[2]) {
        st[0] = BwtEndStats(); st[1] = BwtEndStats();
        if (pr[p].min1 != 255) { st[0].x0 = (int)pr[p].x0_1; st[0].x1 = (int)pr[p].x1_1; st[0].mismatch = pr[p].min1; }
        if (pr[p].min2 != 255) { st[1].x0 = (int)pr[p].x0_2; st[1].x1 = (int)pr[p].x1_2; st[1].mismatch = pr[p].min2; }
    };
    SamWriter w(c.hix, c.so, c.dpOut);
    lap(0);
    // ---- default DP ----
    if (!dpOcc.empty()) {
        s3_dp_half_result *res = nullptr;
        size_t nRes = 0;
        char *pool = nullptr;
        const int cut[2] = {c.halfPar[0].cutoffThreshold, c.halfPar[1].cutoffThreshold};
        if (s3_dp_half_end(c.dpHalf, c.b, lens, dpOcc.data(), dpOcc.size(), c.insertLow, c.insertHigh, cut, c.halfPar[0].softClipLeft, c.halfPar[0].softClipRight, &res, &nRes,
                           &pool) != S3_OK) { err = s3_last_error(); return false; }
        std::vector<uint8_t> done(np, 0);
        std::vector<std::pair<size_t, size_t>> groups;                              // runs of results of the same pair
        for (size_t i = 0; i < nRes;) {
            size_t j = i;
            while (j < nRes && res[j].readID == res[i].readID) j++;
            groups.push_back({i, j});
            i = j;
        }
        w.flush();
        parallelFormat(c.hix, c.so, c.dpOut, (uint32_t)groups.size(), [&](SamWriter &w, uint32_t gk, uint64_t &al) {
            const size_t i = groups[gk].first, j = groups[gk].second;
            // OutputBuffer::ready(OUTPUT_AS_INPUT_ORDER | DO_NOT_OUTPUT_HALF_ALIGNED), DV-DPfunctions.h:367-412 with ScoreCompare<AlgnmtDPResult> (:886)
            std::vector<PairDpHit> hits;
            auto value = [](const s3_dp_half_result &x) -> uint64_t {
                uint64_t al, mm, sc;
                if (x.whichFromDP == 0) { al = 1; mm = (uint32_t)x.score_2; sc = (uint32_t)x.score_1; }
                else if (x.whichFromDP == 1) { al = 1; mm = (uint32_t)x.score_1; sc = (uint32_t)x.score_2; }
                else { al = 0; mm = (uint32_t)(x.algnmt_1 != 0xFFFFFFFFu ? x.score_1 : x.score_2); sc = 0; }
                return al << 63 | (uint64_t)(uint32_t)(0x1FFFFFFFu - (uint32_t)mm) << 32 | (uint32_t)((uint32_t)sc + 0x1FFFFFFFu);
            };
            uint64_t bestV = 0;
            if (moduleType != 1) for (size_t q = i; q < j; q++) bestV = std::max(bestV, value(res[q]));
            for (size_t q = i; q < j; q++) {
                if (moduleType == 1 ? res[q].whichFromDP >= 2 : value(res[q]) != bestV) continue;
                PairDpHit h;
                h.pos1 = res[q].algnmt_1; h.pos2 = res[q].algnmt_2; h.strand1 = res[q].strand_1; h.strand2 = res[q].strand_2; h.whichFromDP = res[q].whichFromDP;
                h.score1 = res[q].score_1; h.score2 = res[q].score_2; h.editdist = (int)res[q].editdist; h.insertSize = (int)res[q].insertSize;
                h.numSameScore = (int)res[q].num_sameScore;
                if (res[q].cigarOffset != 0xFFFFFFFFu) h.cigar = pool + res[q].cigarOffset;
                hits.push_back(h);
            }
            if (moduleType == 3 && hits.size() != 1) hits.clear();
            if (moduleType == 4 && hits.size() > 1) hits.resize(1);
            const uint32_t p = res[i].readID / 2;
            if (!hits.empty() && hits[0].whichFromDP < 2) {
                BwtEndStats st[2];
                statsOf(p, st);
                w.pairDP(c.rs, c.first + res[i].readID, hits.data(), hits.size(), st);
                done[p] = 1;
                al++;
            }
        }, defaultDpPairs);
        s3_free(res); s3_free(pool);
        for (uint32_t p = 0; p < np; p++) if ((route[p] & S3_ROUTE_DEFAULT) && !done[p]) deepPairs.push_back(2 * p);
        std::sort(deepPairs.begin(), deepPairs.end());
    }
    // deep-DP results of a run of pairs -> records (DP2OutputThread :3846-3858: OutputBuffer::ready(), then outputDeepDPResult2)
    std::vector<uint8_t> deepDone(np, 0);
    auto deepRecords = [&](const s3_dp_deep_result *res, size_t nRes, const char *pool, uint64_t &counter) {
    std::vector<std::pair<size_t, size_t>> groups;
    for (size_t i = 0; i < nRes;) {
        size_t j = i;
        while (j < nRes && res[j].readID / 2 == res[i].readID / 2) j++;
        groups.push_back({i, j});
        i = j;
    }
    w.flush();
    parallelFormat(c.hix, c.so, c.dpOut, (uint32_t)groups.size(), [&](SamWriter &w, uint32_t gk, uint64_t &al) {
        const size_t i = groups[gk].first, j = groups[gk].second;
        // OutputBuffer::ready(): sorted by the first end's position, one result per position; best only for -h 3 / 4 (ScoreCompare: score sum)
        std::vector<DeepDpHit> hits;
        for (size_t q = i; q < j; q++) {
            DeepDpHit h;
            h.pos1 = res[q].algnmt_1; h.pos2 = res[q].algnmt_2; h.strand1 = res[q].strand_1; h.strand2 = res[q].strand_2;
            h.score1 = res[q].score_1; h.score2 = res[q].score_2; h.editdist1 = (int)res[q].editdist_1; h.editdist2 = (int)res[q].editdist_2;
            h.insertSize = (int)res[q].insertSize; h.numSame1 = (int)res[q].num_sameScore_1; h.numSame2 = (int)res[q].num_sameScore_2;
            if (res[q].cigarOffset_1 != 0xFFFFFFFFu) h.cigar1 = pool + res[q].cigarOffset_1;
            if (res[q].cigarOffset_2 != 0xFFFFFFFFu) h.cigar2 = pool + res[q].cigarOffset_2;
            hits.push_back(h);
        }
        // (std::sort, not a stable sort -- s3_host.h tells why; found by the every-route scale test, where pairs come with hundreds of deep-DP results)
        keepFirstResultPerPosition(hits);
        if (moduleType != 1) {
            int best = hits[0].score1 + hits[0].score2;
            for (const DeepDpHit &h : hits) best = std::max(best, h.score1 + h.score2);
            size_t w2 = 0;
            for (const DeepDpHit &h : hits) if (h.score1 + h.score2 == best) hits[w2++] = h;
            hits.resize(w2);
            if (moduleType == 3 && hits.size() != 1) hits.clear();
            if (moduleType == 4) hits.resize(1);
        }
        const uint32_t p = res[i].readID / 2;
        if (!hits.empty()) {
            BwtEndStats st[2];
            statsOf(p, st);
            w.pairDeepDP(c.rs, c.first + 2 * p, hits.data(), hits.size(), st);
            deepDone[p] = 1;
            al++;
        }
    }, counter);
    };
    lap(1);
    // ---- new default DP (the reference runs it before the default DP; the stages are independent of each other) ----
    if (c.routing.nNewDefault) {
        s3_dp_deep_result *res = nullptr;
        size_t nRes = 0, nNewPairs = 0;
        char *pool = nullptr;
        uint32_t *newPairs = nullptr;
        const int cut[2] = {c.newPar[0].cutoffThreshold, c.newPar[1].cutoffThreshold};
        if (s3_dp_new_default_routed(c.dp, c.b, c.seedBatch, lens, (uint32_t)c.newPar[0].maxHitNum, c.insertLow, c.insertHigh, cut,
                                     c.newPar[0].softClipLeft, c.newPar[0].softClipRight, &res, &nRes, &pool, nullptr, &newPairs, &nNewPairs) != S3_OK) { err = s3_last_error(); return false; }
        deepRecords(res, nRes, pool, newDefaultPairs);
        for (size_t k = 0; k < nNewPairs; k++) if (!deepDone[newPairs[k] / 2]) deepPairs.push_back(newPairs[k]);   // passUnalignedToDeepDP
        std::sort(deepPairs.begin(), deepPairs.end());
        s3_free(res); s3_free(pool); s3_free(newPairs);
    }
    lap(2);
    // ---- deep DP ----
    if (!deepPairs.empty()) {
        std::vector<uint32_t> todo = deepPairs;
        for (int round = 0; round < 2 && !todo.empty(); round++) {
            s3_dp_deep_result *res = nullptr;
            size_t nRes = 0;
            char *pool = nullptr;
            uint8_t *flags = nullptr;
            const s3_dp_stage_params *par = c.deepPar[round];
            const int cut[2] = {par[0].cutoffThreshold, par[1].cutoffThreshold};
            // seeds go through the seed batch in pieces (at most 12 seeds per read below 400 bases)
            std::vector<uint32_t> next;
            const int stageId = round == 0 ? S3_STAGE_DEEP_DP_ROUND1 : S3_STAGE_DEEP_DP_ROUND2;
            for (size_t at = 0; at < todo.size();) {
                size_t m = 0, nSeeds = 0;                                         // as many pairs as the seed batch holds seeds for
                while (at + m < todo.size()) {
                    const int n1 = seedCountOf(stageId, (int)lens[todo[at + m]]), n2 = seedCountOf(stageId, (int)lens[todo[at + m] + 1]);
                    if (nSeeds + (size_t)(n1 + n2) > 262144) break;
                    nSeeds += (size_t)(n1 + n2);
                    m++;
                }
                if (m == 0) m = 1;
                if (s3_dp_deep(c.dp, c.b, c.seedBatch, lens, todo.data() + at, m, stageId, (uint32_t)par[0].maxHitNum,
                               c.insertLow, c.insertHigh, cut, par[0].softClipLeft, par[0].softClipRight, &res, &nRes, &pool, &flags) != S3_OK) { err = s3_last_error(); return false; }
                for (size_t k = 0; k < m; k++) if (round == 0 && flags[k] == 1) next.push_back(todo[at + k]);
                deepRecords(res, nRes, pool, deepDpPairs);
                s3_free(res); s3_free(pool); s3_free(flags);
                at += m;
            }
            todo.swap(next);
        }
    }
    lap(3);
    // ---- what every stage failed on: each end on its own (outputDeepDPResult2 :624-650: an end without BWT hits goes through the single-end DP;
    //      with hits the FIRST end keeps them, the second end keeps its own only when the first had none) ----
    std::vector<uint32_t> left, todo;
    for (uint32_t r1 : deepPairs) if (!deepDone[r1 / 2]) left.push_back(r1);
    // keepSoap3Results: what the BWT stage left of the leftover pairs' ends (the lists as they were routed, cut at the hit limit)
    std::vector<uint32_t> leftReads;
    for (uint32_t r1 : left) { leftReads.push_back(r1); leftReads.push_back(r1 + 1); }
    std::vector<uint64_t> keptOff(leftReads.size() + 1, 0);
    s3_occ *keptOcc = nullptr;
    if (!leftReads.empty() && s3_batch_fetch_kept(c.b, leftReads.data(), leftReads.size(), keptOff.data(), &keptOcc, nullptr) != S3_OK) { err = s3_last_error(); return false; }
    struct KeptFree { s3_occ *&p; ~KeptFree() { if (p) s3_free(p); } } keptFree{keptOcc};
    std::vector<uint32_t> keptSlot(np, 0xFFFFFFFFu);
    for (size_t lk = 0; lk < left.size(); lk++) keptSlot[left[lk] >> 1] = (uint32_t)lk;
    struct KeptView { const s3_occ *b, *e; const s3_occ *begin() const { return b; } const s3_occ *end() const { return e; } bool empty() const { return b == e; } };
    auto kept = [&](uint32_t r) -> KeptView {
        const size_t i = (size_t)2 * keptSlot[r >> 1] + (r & 1u);
        return KeptView{keptOcc + keptOff[i], keptOcc + keptOff[i + 1]};
    };
    std::vector<uint8_t> viaDP(c.cnt, 0);
    for (uint32_t r1 : left) {
        const bool no1 = kept(r1).empty(), no2 = kept(r1 + 1).empty();
        if (no1) { viaDP[r1] = 1; todo.push_back(r1); }
        if (no2 || !no1) { viaDP[r1 + 1] = 1; todo.push_back(r1 + 1); }
    }
    std::vector<uint32_t> leftSlot(np, 0xFFFFFFFFu);                             // (likewise: lists for the ends of the leftover pairs only)
    for (size_t lk = 0; lk < left.size(); lk++) leftSlot[left[lk] >> 1] = (uint32_t)lk;
    std::vector<std::vector<DpHit>> ownLists(2 * left.size());
    auto own = [&](uint32_t r) -> std::vector<DpHit> & { return ownLists[(size_t)2 * leftSlot[r >> 1] + (r & 1u)]; };
    // (one run per read: a read's DP hits on both strands are one list in position order, as the reference's store holds them)
    if (!singleDpRuns(c.hix, c.so, nullptr, c.b, c.seedBatch, c.dp, lens, todo, c.singlePar, moduleType, [&](SamWriter &, uint32_t k, std::vector<DpHit> &hits) {
            own(k).insert(own(k).end(), hits.begin(), hits.end());
        }, err)) return false;
    parallelFormat(c.hix, c.so, c.unpairOut, (uint32_t)left.size(), [&](SamWriter &up, uint32_t lk, uint64_t &al) {
        const uint32_t r1 = left[lk];
        for (uint32_t r = r1; r < r1 + 2; r++) {
            if (viaDP[r]) continue;
            const int L = (int)lens[r];
            for (const s3_occ &x : kept(r)) {                                      // inputSoap3AnsToArray, PEAlgnmt.cpp:1180
                DpHit h{x.pos, x.strand, L * c.so.dpMatchScore + (int)x.mismatchCount * c.so.dpMismatchScore, (int)x.mismatchCount, std::to_string(L) + "M"};
                h.fromDP = false;
                own(r).push_back(h);
            }
        }
        up.unpairedDP(c.rs, c.first + r1, own(r1).data(), own(r1).size(), own(r1 + 1).data(), own(r1 + 1).size(), c.singlePar.cutoffThreshold);
        al++;
    }, leftoverPairs);
    lap(4);
    return true;
}

static int die(const std::string &m) { fprintf(stderr, "%s\n", m.c_str()); return 1; }
// what a user can do about a chunk whose hit lists outgrow the device
static std::string withHint(const std::string &m)
{
    if (m.find("out of memory") == std::string::npos && m.find("2^32") == std::string::npos) return m;
    return m + "\n(the hit lists of one chunk of reads do not fit: run with fewer reads per chunk -- S3DP_CHUNK_READS=262144 or less -- or a lower hit limit)";
}

// ---- placement: the calling thread (and every thread it starts afterwards) goes to the CPUs of the host NUMA node(s) the given devices hang off, so that
// the chunk buffers, the pinned staging buffers and the threads that fill them sit on the device's side of the socket link.  Measured on a two-socket host
// (2 x 64 cores, MI355X on node 0), ten million 2 x 150 pairs: 4.03-4.17 s with the threads wherever the scheduler put them, 3.63-3.83 s on either single
// node -- what costs is a run spread over both.  An affinity the caller already narrowed (taskset, a cgroup) is only ever narrowed further; a system that
// does not tell the node is left alone; S3DP_NUMA_BIND=0 turns this off.  (The reference leaves placement to the operating system.)
static bool cpusOfNode(int node, cpu_set_t &set)
{
    CPU_ZERO(&set);
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    char text[4096] = {0};
    const bool got = fgets(text, sizeof text, f) != nullptr;
    fclose(f);
    if (!got) return false;
    for (const char *q = text; *q && *q != '\n';) {                           // "0-63,128-191"
        char *e = nullptr;
        const long lo = strtol(q, &e, 10);
        if (e == q) return false;
        long hi = lo;
        if (*e == '-') { q = e + 1; hi = strtol(q, &e, 10); if (e == q) return false; }
        for (long c = lo; c <= hi && c < CPU_SETSIZE; c++) CPU_SET((int)c, &set);
        q = *e == ',' ? e + 1 : e;
    }
    return CPU_COUNT(&set) > 0;
}
static void bindNearDevices(const int *devices, int nDevices)
{
    if (const char *e = getenv("S3DP_NUMA_BIND")) if (atoi(e) == 0) return;
    // (what the program was started with, taken once: a second job of a *-multi run, or a member on another node, is placed within that, not within
    //  what an earlier call left the main thread with)
    static cpu_set_t have;
    static const bool known = sched_getaffinity(0, sizeof have, &have) == 0;
    if (!known) return;
    cpu_set_t want;
    CPU_ZERO(&want);
    for (int k = 0; k < nDevices; k++) {
        int node = -1;
        cpu_set_t c;
        if (s3_device_host_node(devices[k], &node) != S3_OK || node < 0 || !cpusOfNode(node, c)) return;
        CPU_OR(&want, &want, &c);
    }
    CPU_AND(&want, &want, &have);
    if (CPU_COUNT(&want) == 0) return;
    if (sched_setaffinity(0, sizeof want, &want) == 0 && getenv("S3DP_VERBOSE"))
        fprintf(stderr, "[s3dp] host threads on the %d CPUs of the device's NUMA node (of %d)\n", CPU_COUNT(&want), CPU_COUNT(&have));
}

// what the jobs of one program run share: the index in host memory and on the devices (single-multi / pair-multi load it once, SOAP3-DP.cu:869-1140)
struct SharedIndex {
    HostIndex hix;
    s3_group *grp = nullptr;
    std::vector<int> devices;
    bool loaded = false;
};

static int runJob(int argc, char **argv, SharedIndex &shared)
{
    { std::lock_guard<std::mutex> g(g_pairStageMu); for (double &x : g_pairStageSeconds) x = 0; for (double &x : g_deviceMs) x = 0; for (double &x : g_chunkSeconds) x = 0; for (double &x : g_streamSeconds) x = 0; g_dpProblems = g_dpCells = 0; }   // (per job: the *-multi modes run several)
    if (argc < 4) return die("usage: soap3-dp single <index> <reads> [options]\n       soap3-dp pair <index> <reads 1> <reads 2> [-u max insert] [-v min insert] [options]\n"
                             "       soap3-dp single-multi <index> <list file> [options]\n       soap3-dp pair-multi <index> <list file> [options]");
    const bool paired = strcmp(argv[1], "pair") == 0;
    if (!paired && strcmp(argv[1], "single") != 0) return die("the first argument must be 'single', 'pair', 'single-multi' or 'pair-multi'");
    // -bam: the reads come from ONE BAM file, the mates of a pair behind each other (isReadBAM, IniParam.cpp:449-459)
    bool readBam = false;
    for (int i = 2; i < argc; i++) if (!strcmp(argv[i], "-bam") || !strcmp(argv[i], "-BAM")) readBam = true;
    if (paired && !readBam && argc < 5) return die("pair mode needs two read files");
    Ini ini;
    parseIni(std::string(argv[0]) + ".ini", ini);
    const std::string indexName = argv[2], queryFile = argv[3], queryFile2 = paired && !readBam ? argv[4] : "";
    int insertHigh = 500, insertLow = 1;                                       // IniParam.cpp:482-483
    int maxReadLength = 120, numMismatch = -1, alignmentType = 2, outputFormat = 2;             // DEFAULT_MAX_READ_LEN, all-best, SAM
    std::vector<int> devices(1, 0);                                            // -c: the reference takes one GPU id (SOAP3-DP.cu:219-220); here also a list "0,1,2"
    bool enableDP = true, skipSoap3 = false;
    std::string prefix = queryFile, readGroup, sampleName, readGrpOption;
    bool printMDNM = false, outputBam = false;
    for (int i = paired && !readBam ? 5 : 4; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](const char *what) -> const char * { if (i + 1 >= argc) { fprintf(stderr, "Please specify %s after '%s'\n", what, a.c_str()); exit(1); } return argv[++i]; };
        if (a == "-L" || a == "-l") maxReadLength = atoi(need("the length"));
        else if (a == "-h") alignmentType = atoi(need("the output option"));
        else if (a == "-b") {
            // 2 = SAM through the samtools API; 3 = BAM: the same API with isOutputBinary (IniParam.cpp:598-604; samopen(.., "wb"), SOAP3-DP.cu:413-416)
            outputFormat = atoi(need("the output file format"));
            if (outputFormat == 3) { outputFormat = 2; outputBam = true; }
        } else if (a == "-bam" || a == "-BAM") { /* seen above */ }
        else if (a == "-o") prefix = need("the output prefix");
        else if (a == "-c") {
            devices.clear();
            for (const char *q = need("the GPU device ID"); *q;) {
                char *e = nullptr;
                devices.push_back((int)strtol(q, &e, 10));
                if (e == q) return die("The GPU device ID should be a number or a comma-separated list of numbers");
                q = *e == ',' ? e + 1 : e;
            }
            if (devices.empty()) return die("Please specify the GPU device ID after '-c'");
        }
        else if (a == "-D") readGroup = need("the read group");
        else if (a == "-A") sampleName = need("the sample name");
        else if (a == "-R") readGrpOption = need("the read group option");
        else if (a == "-p") printMDNM = true;
        else if (a == "-u") insertHigh = atoi(need("the maximum insert size"));
        else if (a == "-v") insertLow = atoi(need("the minimum insert size"));
        else if (a == "-s") {
            enableDP = false;
            if (i + 1 < argc && argv[i + 1][0] >= '0' && argv[i + 1][0] <= '9') numMismatch = atoi(argv[++i]);
        } else return die("unknown option " + a);
    }
    if (outputFormat != 2) return die("only SAM (-b 2) and BAM (-b 3) output are built");
    if (alignmentType < 1 || alignmentType > 4) return die("The output option should be 1, 2, 3 or 4");
    if (maxReadLength > S3_MAX_READ_LENGTH) return die("The length should not be greater than 1024");
    if (readGroup.empty()) readGroup = queryFile;
    if (sampleName.empty()) sampleName = "default";
    bindNearDevices(devices.data(), (int)devices.size());                       // before the loader thread and the index buffers
    std::string err;
    HostIndex &hix = shared.hix;
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
    // The read files come in chunks (ReadChunker): a loader thread parses chunk i+1 -- with all host threads -- while chunk i is aligned and
    // chunk i-1 is formatted and written, as the reference's loader thread fills the next batch buffer (aio_thread.cpp:582, SOAP3-DP.cu:607-800).
    // It starts before the index is read: neither needs the other.  S3DP_CHUNK_READS: reads per chunk.  The default is 1 Mi, the reference's own batch
    // (SOAP3-DP.cu:394), and it is the measured optimum of the whole program, not of the device: a ladder of 1 Mi reads costs the device 2.2 times as much per
    // read as one of 10 M (13.7 against 30 M reads/s resident -- the tails of its launches), but twenty million reads in chunks of 1 / 2 / 4 Mi with three in
    // flight take 1.41 / 1.62 / 2.06 s to /dev/null (gpurun_out/r05e2): with few chunks the first output waits for a long first search and the last chunk's
    // formatting has nothing to hide behind.
    size_t chunkReadsEnv = 0;
    if (const char *e = getenv("S3DP_CHUNK_READS")) chunkReadsEnv = (size_t)strtoull(e, nullptr, 10);
    // (all valid alignments of reads above 120 bases: the 100-base seeds are searched without a hit limit, a read inside a repeat family brings thousands
    //  of hits.  Until round 5 four batches of 1 Mi such reads grew past the HBM the index leaves on the 3.1 Gbp repeat-rich genome; the decode works in
    //  slices now (S3DP_LONG_BUDGET) and 1 Mi / 2 Mi chunks run -- 8.8 / 12.1 s for ten million reads where a quarter of the chunk takes 6.7 s
    //  (gpurun_out/r05h2): the small chunk stays the default of this mode because it is the faster one)
    const size_t chunkDefault = (alignmentType == 1 && maxReadLength > 120) ? (size_t)1 << 18 : (size_t)1 << 20;
    const uint32_t chunkReads = (uint32_t)((chunkReadsEnv ? chunkReadsEnv : chunkDefault) / 64 * 64 + (chunkReadsEnv && chunkReadsEnv < 64 ? 64 : 0));
    // (Large chunks first, small ones at the end -- the search kernels' cost per read falls with the batch, 2.95 / 2.14 / 1.66 s of BWT stages for ten million
    //  pairs in chunks of 1 / 2 / 4 Mi reads -- was measured in round 5: pairs 3.55 s without, 3.65-3.78 s with; single reads 1.41 s without, 1.70-2.2 s with:
    //  the DP stages and the output of the large chunks take what the search wins.  Taken out in round 6; profiles/README_r05.md section 6.)
    ReadChunker chunker;
    if (readBam ? !chunker.openBam(queryFile, paired, (uint32_t)maxReadLength, 33, ini.maxLenReadName, err)
                : !chunker.open(queryFile, queryFile2, (uint32_t)maxReadLength, 33, ini.maxLenReadName, err)) return die(err);
    struct ChunkQueue {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<std::unique_ptr<ReadSet>> q;
        bool done = false;
        std::string err;
        double seconds = 0;
    } loaded;
    // chunks that have been written out come back here: a chunk of 1 Mi reads is 230 MB in five arrays, and giving them back to the system and asking for
    // them again cost a munmap in the writer thread and sixty thousand page faults in the loader per chunk
    struct { std::mutex mu; std::vector<std::unique_ptr<ReadSet>> free; } pool;
    auto recycle = [&](std::unique_ptr<ReadSet> rs) {
        if (!rs) return;
        std::lock_guard<std::mutex> g(pool.mu);
        if (pool.free.size() < 8) pool.free.push_back(std::move(rs));
    };
    std::thread loader([&] {
        for (;;) {
            std::unique_ptr<ReadSet> rs;
            {
                std::lock_guard<std::mutex> g(pool.mu);
                if (!pool.free.empty()) { rs = std::move(pool.free.back()); pool.free.pop_back(); }
            }
            if (!rs) rs.reset(new ReadSet());
            std::string e;
            const auto t = std::chrono::steady_clock::now();
            const bool ok = chunker.next(chunkReads, *rs, e);
            std::unique_lock<std::mutex> lk(loaded.mu);
            loaded.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
            if (!ok) { loaded.err = e; loaded.done = true; loaded.cv.notify_all(); return; }
            loaded.cv.wait(lk, [&] { return loaded.q.size() < 2; });
            loaded.q.push_back(std::move(rs));
            loaded.cv.notify_all();
        }
    });
    auto nextChunk = [&]() -> std::unique_ptr<ReadSet> {
        std::unique_lock<std::mutex> lk(loaded.mu);
        loaded.cv.wait(lk, [&] { return !loaded.q.empty() || loaded.done; });
        if (loaded.q.empty()) return nullptr;
        std::unique_ptr<ReadSet> r = std::move(loaded.q.front());
        loaded.q.pop_front();
        loaded.cv.notify_all();
        return r;
    };
    // (a failure below leaves through exit(): the loader thread must not outlive main's locals silently)
    auto dieJoin = [&](const std::string &m) -> int { fflush(stdout); fprintf(stderr, "%s\n", withHint(m).c_str()); fflush(stderr); _exit(1); return 1; };
    std::unique_ptr<ReadSet> chunk = nextChunk();
    if (!chunk) {
        if (!loaded.err.empty()) return dieJoin(loaded.err);
        chunk.reset(new ReadSet());                                            // an empty input still gets its (header-only) output files
        uint32_t w = 1;
        while (w < (uint32_t)maxReadLength) w *= 2;
        chunk->wpq = w / 16 ? w / 16 : 1; chunk->maxReadLength = (uint32_t)maxReadLength;
    }
    printf("[Main] Loaded %u short reads from the query file (first chunk; %d host threads, alongside the index).\n", chunk->n, hostThreads());
    if (numMismatch < 0) numMismatch = maxReadLength > 50 ? 3 : 2;          // getDefaultMismatchNum, CPUfunctions.cpp:26-35
    // DP enabled (the default): the BWT stage runs with Soap3MisMatchAllow mismatches (SOAP3-DP.cu:208-213); reads it leaves unaligned go
    // through seeding + DP and are reported in <prefix>.dpout.1.  Too short reads switch the DP off (SOAP3-DP.cu:643-652).
    uint32_t detectedReadLength = 0, detectedReadLength2 = 0;              // GetReadLength: the longest of the first ten reads (of each end)
    if (!paired) for (uint32_t i = 0; i < chunk->n && i < 10; i++) detectedReadLength = std::max(detectedReadLength, chunk->lengths[i]);
    else
        for (uint32_t i = 0, k = 0; i + 1 < chunk->n && k < 10; i += 2, k++) {
            detectedReadLength = std::max(detectedReadLength, chunk->lengths[i]);
            detectedReadLength2 = std::max(detectedReadLength2, chunk->lengths[i + 1]);
        }
    if (paired && insertLow < (int)detectedReadLength2) insertLow = (int)detectedReadLength2;   // SOAP3-DP.cu:632-634
    if (enableDP) {
        numMismatch = ini.soap3MisMatchAllow;
        if (detectedReadLength < 30 || (paired && detectedReadLength2 < 30)) { enableDP = false; printf("Dynamic programming is disabled because read length < %i\n", 30); }
        else if (paired && (detectedReadLength > 150 || detectedReadLength2 > 150)) {
            skipSoap3 = true;                                                  // SOAP3-DP.cu:658-662, alignment.cu:1928-1944: every pair goes straight to the deep DP
            printf("All reads are directly processed by DP, because the read length > 150\n");
        }
        else if (paired && (detectedReadLength <= 50 || detectedReadLength2 <= 50)) numMismatch = 1;   // SOAP3-DP.cu:663-666
        if (!enableDP && numMismatch < 0) numMismatch = maxReadLength > 50 ? 3 : 2;
    }
    // the DP aligners' traceback tables are a few gigabytes per device (20 ms per GB to allocate): made while the index is read and uploaded
    constexpr int kDpWorkers = 4;                                             // at most: DP stage threads per device (each with its own aligners and seed batch)
    // two by default (S3DP_DP_THREADS=1 for one).  A stage thread's work is a chain of small launches and fetches, each of which waits behind the search
    // kernels of the chunk loop -- persistent waves that hold the device for tens of milliseconds at a time: one thread needed 5.1 s for the stages of 10 M
    // pairs while the chunk loop needed 3.5 s; two threads on alternate chunks hide each other's waits (10 M pairs: 5.1 -> 4.1 s).  Round 3 found no gain in
    // a second thread because the stages were host work then (routing of the hit lists: on the device since round 4) and the box grants sixteen CPUs.
    // With two threads the records of different chunks reach .dpout / .unpair in the order the threads finish (the reference's own output threads write
    // in no fixed order either, SOAP3-DP.cu:251-258): the SET of records is the result; S3DP_DP_THREADS=1 gives a reproducible file order.
    // (S3DP_DUMP_ROUTING, the tests' routing dump, is appended chunk by chunk by the stage thread: one thread, or chunk 1 could write before chunk 0 truncates the file)
    const int dpThreadsEnv = getenv("S3DP_DP_THREADS") ? atoi(getenv("S3DP_DP_THREADS")) : 0;
    // (with large chunks a chunk's DP stages last longer than the search of the next one: three threads keep up with them)
    // (pairs: three since the search kernels of round 5 -- a chunk's search is over before two threads are through the stages of the two chunks before it;
    //  10 M pairs, outputs to /dev/null: 3.40 -> 3.25 s with three threads, 3.18 s with a search grid that leaves a workgroup per compute unit to their kernels)
    int nDpWorkers = getenv("S3DP_DUMP_ROUTING") ? 1 : (dpThreadsEnv >= 1 && dpThreadsEnv <= kDpWorkers ? dpThreadsEnv : (paired ? 3 : 2));
    // (every stage thread brings its own aligners -- up to 8 GB of traceback tables each, twice for pairs: when the device has no room for a second or a third
    //  set beside a large index, the run goes on with the threads that got theirs: `workers`, the smallest count over the devices)
    struct PreDp { s3_dp *dp[kDpWorkers] = {}, *dpHalf[kDpWorkers] = {}; int workers = 0; std::string err; };
    const std::vector<int> &dpDevices = shared.loaded ? shared.devices : devices;
    std::vector<PreDp> preDp(dpDevices.size());
    std::vector<std::thread> preDpThreads;
    auto dpShapes = [&](int &maxRL, int &maxDNA, int &maxDNAHalf) {
        maxRL = (maxReadLength / 4 + 1) * 4;
        const int margin = maxReadLength > 100 ? maxReadLength >> 2 : 25;
        maxDNA = maxRL + 2 * margin + 8;                                      // DV-DPfunctions.cu:1580
        maxDNAHalf = insertHigh - insertLow + maxReadLength + 1;              // HalfEndAlignmentEngine::performAlignment, DV-DPfunctions.cu:2221-2224
    };
    if (enableDP)
        for (size_t m = 0; m < dpDevices.size(); m++) preDpThreads.emplace_back([&, m] {
            int maxRL, maxDNA, maxDNAHalf;
            dpShapes(maxRL, maxDNA, maxDNAHalf);
            // SemiGlobalAligner::decideConfiguration (DV-DPfunctions.cu:568-612) takes scheme 1 (one table over the whole window) whenever 64 blocks of full
            // tables fit the device, which on 288 GB they always do.  S3DP_DP_SCHEME=2 asks for what the reference does on a device too small for that: scheme 2
            // (score pass over the window, table of L + L/2 + 8 columns) wherever the window is at least three read lengths (coefConf[0]), scheme 1 elsewhere.
            const int askScheme = getenv("S3DP_DP_SCHEME") && atoi(getenv("S3DP_DP_SCHEME")) == 2 ? 2 : 1;
            auto wantScheme = [&](int rl, int dna) { return askScheme == 2 && dna >= rl * 3 ? 2 : 1; };
            // problems per DP launch: the traceback tables of one launch stay under 8 GB (ten thousand problems fill the chip as well as thirty thousand do)
            auto problems = [&](int rl, int tl) -> uint32_t {
                const double perProblem = (double)((rl + 63) / 64) * (tl + 64) * 64 * 4;
                const double n = 8e9 / perProblem;
                return n >= 32768 ? 32768u : n < 256 ? 256u : (uint32_t)n;
            };
            int scheme = 1, tableLen = 0, patLen = 0;
            s3_dp_scores sc{ini.matchScore, ini.mismatchScore, ini.gapOpenScore, ini.gapExtendScore};
            s3_dp_config(maxRL, maxDNA, wantScheme(maxRL, maxDNA), &scheme, &tableLen, &patLen);
            PreDp &P = preDp[m];
            const int want = nDpWorkers;
            P.workers = want;
            for (int w = 0; w < want; w++)
                if (s3_dp_create(dpDevices[m], problems(maxRL, tableLen), (uint32_t)maxRL, (uint32_t)maxDNA, (uint32_t)tableLen, scheme, &sc, &P.dp[w]) != S3_OK) {
                    if (w == 0) { P.err = s3_last_error(); return; }
                    P.dp[w] = nullptr; P.workers = w;
                    break;
                }
            if (paired) {
                s3_dp_config(maxRL, maxDNAHalf, wantScheme(maxRL, maxDNAHalf), &scheme, &tableLen, &patLen);
                for (int w = 0; w < P.workers; w++)
                    if (s3_dp_create(dpDevices[m], problems(maxRL, tableLen), (uint32_t)maxRL, (uint32_t)maxDNAHalf, (uint32_t)tableLen, scheme, &sc, &P.dpHalf[w]) != S3_OK) {
                        if (w == 0) { P.err = s3_last_error(); return; }
                        P.dpHalf[w] = nullptr;
                        for (int v = w; v < P.workers; v++) { if (P.dp[v]) s3_dp_free(P.dp[v]); P.dp[v] = nullptr; }
                        P.workers = w;
                        break;
                    }
            }
        });
    // single reads with the DP stages off go through a stream of batches (below): the pinned host staging of its slots is made NOW, while the index files are
    // read -- the runtime starts no kernel while it pins, and three slots pinning their 256 MB of occurrence lists in the first batches of a run kept the
    // device idle for 0.2 s (S3DP_STAGING_OCC_PER_READ, default 48 entries per read and list; 0: the slots pin what they need when they need it)
    const int streamDepth = getenv("S3DP_STREAM_DEPTH") && atoi(getenv("S3DP_STREAM_DEPTH")) > 0 ? atoi(getenv("S3DP_STREAM_DEPTH")) : 3;
    if (!paired && !enableDP && !shared.loaded) {
        const long occPerRead = getenv("S3DP_STAGING_OCC_PER_READ") ? atol(getenv("S3DP_STAGING_OCC_PER_READ")) : 48;
        uint32_t w16 = 1;
        while (w16 < (uint32_t)maxReadLength) w16 *= 2;
        const uint32_t wpqGuess = w16 / 16 ? w16 / 16 : 1;                          // (words per read as ReadChunker::next packs them)
        if (occPerRead > 0 && (size_t)chunkReads * (size_t)occPerRead * 8 * (size_t)streamDepth <= ((size_t)3 << 30))
            preDpThreads.emplace_back([=] { (void)s3_host_staging_reserve(chunkReads, wpqGuess, streamDepth, (uint32_t)occPerRead); });
    }
    if (!shared.loaded) {
        printf("[Main] loading index into host...\n");
        if (!loadIndex(indexName, hix, err)) return dieJoin(err);
        printf("[Main] Finished loading index into host.\n[Main] Loading time : %9.4f seconds\n", since(t0));
        printf("[Main] Reference sequence length : %u\n", hix.n);
        // the index is replicated: one copy per device, uploaded at the same time (s3_group_create); the read chunks are dealt to the devices as they
        // come free, each device writes its own set of output files (as the reference writes one .gout.N per CPU thread, SOAP3-DP.cu:251-258)
        s3_index_desc d = hix.desc();
        if (s3_group_create(devices.data(), (int)devices.size(), &d, &shared.grp) != S3_OK) return dieJoin(s3_last_error());
        shared.devices = devices;
        shared.loaded = true;
        printf("[Main] Finished copying index into device (GPU).\n");
        // (the long lookup tables are sized by the HBM that is free: 16-mers want 103 GB; a run beside another process's allocations may get 15- or 14-mers --
        //  the same records, more rank queries per read)
        printf("[Main] Lookup tables in device memory : %d characters\n", s3_index_lookup_chars(s3_group_index(shared.grp, 0)));
        printf("[Main] Flank tables of check-and-extend in device memory : %d of 2\n", s3_index_flank_tables(s3_group_index(shared.grp, 0)));
    } else devices = shared.devices;                                           // (later jobs of a list run on the devices the index is on)
    s3_group *grp = shared.grp;
    const int nMembers = (int)devices.size();
    for (std::thread &t : preDpThreads) t.join();
    for (const PreDp &x : preDp) if (!x.err.empty()) return dieJoin(x.err);
    if (enableDP) {
        int got = nDpWorkers;
        for (const PreDp &x : preDp) got = std::min(got, x.workers);
        if (got < nDpWorkers) {
            fprintf(stderr, "[Main] device memory for %d DP stage thread%s per device only (asked: %d)\n", got, got == 1 ? "" : "s", nDpWorkers);
            for (PreDp &x : preDp)
                for (int w = got; w < kDpWorkers; w++) {
                    if (x.dp[w]) s3_dp_free(x.dp[w]);
                    if (x.dpHalf[w]) s3_dp_free(x.dpHalf[w]);
                    x.dp[w] = x.dpHalf[w] = nullptr;
                }
            nDpWorkers = got;
        }
    }
    const auto tAlign = std::chrono::steady_clock::now();
    // the first chunk goes back to the head of the queue: the members take their chunks from there
    if (chunk->n) { std::lock_guard<std::mutex> g(loaded.mu); loaded.q.push_front(std::move(chunk)); }
    chunk.reset();
    enum { C_ALIGNED, C_READS, C_DP_READS, C_DEFAULT_DP, C_NEW_DEFAULT_DP, C_DEEP_DP, C_LEFTOVER, C_COUNTERS };
    std::vector<uint64_t> counters((size_t)nMembers * C_COUNTERS, 0);
    std::vector<double> memberDpSeconds((size_t)nMembers, 0);
    auto runMember = [&](int member) -> int {
    if (nMembers > 1) bindNearDevices(&devices[(size_t)member], 1);           // (each member's threads to its own device's node)
    s3_index *ix = s3_group_index(grp, member);
    const std::string fileNo = std::to_string(member + 1);
    double dpStageSeconds = 0;
    std::unique_ptr<ReadSet> chunk = nextChunk();
    if (!chunk) {
        if (member > 0) {                                                      // fewer chunks than devices: nothing for this one (member 0 still writes the headers)
            // its aligners were made while the index loaded: several GB of traceback tables per device that no later job would ever free
            for (int w = 0; w < kDpWorkers; w++) {
                if (preDp[(size_t)member].dp[w]) s3_dp_free(preDp[(size_t)member].dp[w]);
                if (preDp[(size_t)member].dpHalf[w]) s3_dp_free(preDp[(size_t)member].dpHalf[w]);
                preDp[(size_t)member].dp[w] = preDp[(size_t)member].dpHalf[w] = nullptr;
            }
            return 0;
        }
        chunk.reset(new ReadSet());
        uint32_t w = 1;
        while (w < (uint32_t)maxReadLength) w *= 2;
        chunk->wpq = w / 16 ? w / 16 : 1; chunk->maxReadLength = (uint32_t)maxReadLength;
    }
    static const bool verboseTimeline = getenv("S3DP_VERBOSE") != nullptr;
    auto mark = [&](const char *what) { if (verboseTimeline) fprintf(stderr, "[s3dp] member %d: %-44s at %8.3f s after the index was resident\n", member, what, since(tAlign)); };
    mark("first chunk in hand");
    const std::string outName = prefix + ".gout." + fileNo;
    FILE *out = fopen(outName.c_str(), "w");
    if (!out) return dieJoin("Could not open the output file " + outName);
    // -b 3: the same records as BAM (s3_bam.h); the header's reference dictionary is the index's
    std::vector<uint32_t> refLengths;
    for (int i = 0; i < hix.numOfSeq; i++) refLengths.push_back(hix.actualEnd[i] - hix.actualStart[i]);
    // S3DP_BAM_LEVEL: zlib level of the BGZF blocks (default 1: 7 us per record on one core; samtools' default, -1 = level 6, costs 13 us and saves a fifth)
    const int bamLevel = getenv("S3DP_BAM_LEVEL") ? atoi(getenv("S3DP_BAM_LEVEL")) : 1;
    auto asBam = [&](FILE *f) { if (outputBam) bamRegister(f, hix.seqName, refLengths, bamLevel); };
    asBam(out);
    SamOptions so;
    so.readGroup = readGroup; so.sampleName = sampleName; so.readGrpOption = readGrpOption; so.alignmentType = alignmentType;
    so.minMAPQ = ini.minMAPQ; so.maxMAPQ = ini.maxMAPQ; so.bwaLikeScore = ini.bwaLike; so.printMDNM = printMDNM;
    so.dpMatchScore = ini.matchScore; so.dpMismatchScore = ini.mismatchScore;
    SamWriter sam(hix, so, out);
    sam.header();
    FILE *dpOut = nullptr;
    s3_dp *dpSet[kDpWorkers] = {}, *halfSet[kDpWorkers] = {};                 // one set of aligners and one seed batch per DP stage thread
    s3_batch *seedSet[kDpWorkers] = {};
    s3_dp_stage_params dpPar{}, halfPar[2] = {}, deepPar[2][2] = {}, newPar[2] = {};
    FILE *unpairOut = nullptr;
    uint64_t dpAlignedReads = 0, defaultDpPairs = 0, newDefaultPairs = 0, deepDpPairs = 0, leftoverPairs = 0;
    if (enableDP) {
        const std::string dpName = prefix + ".dpout." + fileNo;
        dpOut = fopen(dpName.c_str(), "w");
        if (!dpOut) return dieJoin("Could not open the output file " + dpName);
        asBam(dpOut);
        SamWriter dpHdr(hix, so, dpOut);
        dpHdr.header();
        s3_dp_ini di{ini.dpScoreThreshold, ini.maxFrontLenClipped, ini.maxEndLenClipped};
        s3_dp_stage_parameters(S3_STAGE_SINGLE_DP, detectedReadLength, &di, &dpPar);      // getParameterForSingleDP on the detected length (alignment.cu:2580)
        for (int w = 0; w < nDpWorkers; w++) {                                // (the aligners were made while the index was loaded, above)
            dpSet[w] = preDp[(size_t)member].dp[w];
            if (s3_batch_create(ix, 262144 + 64, 8, &seedSet[w]) != S3_OK) return dieJoin(s3_last_error());
        }
        if (paired) {
            for (int w = 0; w < nDpWorkers; w++) halfSet[w] = preDp[(size_t)member].dpHalf[w];
            s3_dp_stage_parameters(S3_STAGE_DEFAULT_DP, detectedReadLength, &di, &halfPar[0]);
            s3_dp_stage_parameters(S3_STAGE_DEFAULT_DP, detectedReadLength2, &di, &halfPar[1]);
            for (int e = 0; e < 2; e++) {
                s3_dp_stage_parameters(S3_STAGE_NEW_DEFAULT_DP, e ? detectedReadLength2 : detectedReadLength, &di, &newPar[e]);
                if (s3_dp_stage_parameters(S3_STAGE_DEEP_DP_ROUND1, e ? detectedReadLength2 : detectedReadLength, &di, &deepPar[0][e]) != S3_OK ||
                    s3_dp_stage_parameters(S3_STAGE_DEEP_DP_ROUND2, e ? detectedReadLength2 : detectedReadLength, &di, &deepPar[1][e]) != S3_OK)
                    return dieJoin("s3_dp_stage_parameters refused a deep-DP stage");
            }
            const std::string upName = prefix + (member == 0 ? ".unpair" : ".unpair." + fileNo);
            unpairOut = fopen(upName.c_str(), "w");
            if (!unpairOut) return dieJoin("Could not open the output file " + upName);
            asBam(unpairOut);
            SamWriter upHdr(hix, so, unpairOut);
            upHdr.header();
        }
    }
    // ---- single reads with the DP stages off: streamed -- three batches in flight (upload / kernels / download on their own HIP streams,
    // s3_stream_*), this thread feeds them, a second one formats and writes what comes back ---------------------------------------------------
    uint64_t aligned = 0, totalReads = 0;
    std::thread streamFreer;
    if (!paired && !enableDP) {
        s3_stream *st = nullptr;
        const uint32_t cap = chunk->n == chunkReads ? chunkReads : (chunk->n ? chunk->n : 1);
        // three batches in flight (S3DP_STREAM_DEPTH).  The library alone runs 1 Mi-read batches faster six deep (0.85 of the resident rate against 0.75,
        // bench.py's streamed figure); behind the loader and the formatter three are faster (20 M reads, repeat-rich genome: 1.6 s against 2.2; uniform:
        // 0.87 against 1.1 -- profiles/r03_e2e_depth_queues.txt)
        if (s3_stream_create(ix, cap, chunk->wpq, streamDepth, &st) != S3_OK) return dieJoin(s3_last_error());
        mark("stream created");
        struct InFlight { std::mutex mu; std::condition_variable cv; std::deque<std::unique_ptr<ReadSet>> q; bool done = false; } fl;
        std::string werr;
        std::thread writer([&] {
            for (;;) {
                std::unique_ptr<ReadSet> rsP;
                {
                    std::unique_lock<std::mutex> lk(fl.mu);
                    fl.cv.wait(lk, [&] { return !fl.q.empty() || fl.done; });
                    if (fl.q.empty()) return;
                    rsP = std::move(fl.q.front());
                    fl.q.pop_front();
                }
                fl.cv.notify_all();
                const ReadSet &rs = *rsP;
                s3_stream_result res;
                const auto tN = std::chrono::steady_clock::now();
                if (s3_stream_next(st, &res) != S3_OK) { werr = s3_last_error(); fprintf(stderr, "%s\n", werr.c_str()); fflush(stderr); _exit(1); }
                const uint32_t cnt = res.n;
                addStreamSeconds(3, since(tN));
                if (res.batch) addBatchTiming(res.batch);                       // (device time of the batch's ladder: batches in flight share the device, the sum is above the wall clock)
                const auto tFm = std::chrono::steady_clock::now();
                parallelFormat(hix, so, out, cnt, [&](SamWriter &w, uint32_t k, uint64_t &al) {
                    const size_t o = res.occOffsets[k], m = res.occOffsets[k + 1] - res.occOffsets[k];
                    uint8_t sbuf[64], mbuf[64];
                    std::vector<uint8_t> sv, mv;
                    uint8_t *sp = sbuf, *mp = mbuf;
                    if (m > 64) { sv.resize(m); mv.resize(m); sp = sv.data(); mp = mv.data(); }
                    for (size_t j = 0; j < m; j++) { sp[j] = (uint8_t)(res.occMeta[o + j] & 0xFF); mp[j] = (uint8_t)((res.occMeta[o + j] >> 8) & 0xFF); }
                    if (alignmentType >= 3) {
                        // unique best: exactly one hit, otherwise a no-answer record; random best: the first hit (hostKernel, CPUfunctions.cpp:1861-1930)
                        if (m == 0 || (alignmentType == 3 && m > 1)) w.unaligned(rs, k);
                        else { al++; w.singleAnswer(rs, k, res.occPos[o], sp[0], mp[0]); }
                        return;
                    }
                    if (m) al++;
                    w.singleEnd(rs, k, res.occPos + o, sp, mp, m, maxReadLength > 120 ? (rs.lengths[k] > 120 ? 100u : rs.lengths[k]) : 0u);
                }, aligned);
                addStreamSeconds(4, since(tFm));
                if (s3_stream_release(st) != S3_OK) { fprintf(stderr, "%s\n", s3_last_error()); fflush(stderr); _exit(1); }
                recycle(std::move(rsP));
            }
        });
        while (chunk && chunk->n) {
            uint32_t *q = nullptr, *l = nullptr, *sl = nullptr;
            const auto tAq = std::chrono::steady_clock::now();
            if (s3_stream_acquire(st, &q, &l, &sl) != S3_OK) return dieJoin(s3_last_error());
            addStreamSeconds(1, since(tAq));
            const uint32_t cnt = chunk->n;
            const auto tCp = std::chrono::steady_clock::now();
            memcpy(q, chunk->queries.data(), chunk->queries.size() * 4);
            memcpy(l, chunk->lengths.data(), (size_t)cnt * 4);
            addStreamSeconds(2, since(tCp));
            // reads longer than 120 bases: only the first 100 are searched, the rest is verified on the device (alignment.cu:2473-2491)
            if (maxReadLength > 120) for (uint32_t k = 0; k < cnt; k++) sl[k] = chunk->lengths[k] > 120 ? 100u : chunk->lengths[k];
            if (s3_stream_submit(st, cnt, maxReadLength > 120, (uint32_t)numMismatch, ini.maxOutputPerRead, alignmentType, nullptr) != S3_OK) return dieJoin(s3_last_error());
            totalReads += cnt;
            {
                std::unique_lock<std::mutex> lk(fl.mu);
                fl.q.push_back(std::move(chunk));
            }
            fl.cv.notify_all();
            const auto tLd = std::chrono::steady_clock::now();
            chunk = nextChunk();
            addStreamSeconds(0, since(tLd));
        }
        {
            std::lock_guard<std::mutex> g(fl.mu);
            fl.done = true;
        }
        fl.cv.notify_all();
        writer.join();
        mark("stream loop done");
        // (the slots' pinned staging and device buffers take 0.2-0.4 s to give back: done while the last records are written)
        streamFreer = std::thread([st] { s3_stream_free(st); });
        if (!loaded.err.empty()) return dieJoin(loaded.err);
        chunk.reset();
    }
    mark("DP aligners and seed batch created");
    s3_batch *b = nullptr;
    const uint32_t maxBatch = chunkReads;
    if (chunk && s3_batch_create(ix, chunk->n == chunkReads ? chunkReads : (chunk->n ? chunk->n : 1), chunk->wpq, &b) != S3_OK) return dieJoin(s3_last_error());
    mark("batch created");
    // ---- pairs with the DP stages on: the stages of chunk k run on a thread of their own while this one uploads, searches and writes the paired
    // records of chunk k + 1 -- two batches take turns (the DP packers read a chunk's reads where the search left them); the stages are host work around
    // a quarter second of kernels per ten million pairs and used to run between two chunks' searches ------------------------------------------------
    struct PairJob {
        std::unique_ptr<ReadSet> rs; s3_batch *b = nullptr; uint32_t cnt = 0;
        std::vector<s3_pair_result> pr; s3_pair_routing routing{}; uint64_t chunkNo = 0, readBase = 0;
        std::vector<uint32_t> todo; bool hadSeedLengths = false;                  // single reads (default mode): the reads the BWT stage left without a hit
        std::vector<uint32_t> seeds;                                              // single reads above 120 bases: the searched length of every read
    };
    struct { std::mutex mu; std::condition_variable cv; std::deque<std::unique_ptr<PairJob>> q; std::vector<s3_batch *> freeBatches; bool stop = false; } dpq;
    std::vector<s3_batch *> stageBatches;                                         // the batches that take turns: one under the search, one with each DP stage thread
    std::thread dpWorkers[kDpWorkers];
    uint64_t workerCounts[kDpWorkers][5] = {};                                    // per DP stage thread: reads aligned by the single-end DP, default / new default / deep DP pairs, leftovers
    double workerSeconds[kDpWorkers] = {};
    const bool pipelinedDp = enableDP && chunk && chunk->n;                      // (pairs and, in the default mode, single reads)
    if (pipelinedDp) {
        stageBatches.push_back(b);
        for (int w = 0; w < nDpWorkers; w++) {
            s3_batch *nb = nullptr;
            if (s3_batch_create(ix, chunk->n == chunkReads ? chunkReads : chunk->n, chunk->wpq, &nb) != S3_OK) return dieJoin(s3_last_error());
            stageBatches.push_back(nb);
        }
        dpq.freeBatches = stageBatches;
        // the searches of a paired run share the device with the DP stages of the chunks before: their grids leave a workgroup per compute unit to the DP kernels
        if (paired) for (s3_batch *sb : stageBatches) if (s3_batch_set_grid(sb, 3) != S3_OK) return dieJoin(s3_last_error());
        for (int wk = 0; wk < nDpWorkers; wk++) dpWorkers[wk] = std::thread([&, wk] {
            s3_dp *dpW = dpSet[wk], *halfW = halfSet[wk];
            s3_batch *seedW = seedSet[wk];
            uint64_t *cntW = workerCounts[wk];
            for (;;) {
                std::unique_ptr<PairJob> j;
                {
                    std::unique_lock<std::mutex> lk(dpq.mu);
                    dpq.cv.wait(lk, [&] { return !dpq.q.empty() || dpq.stop; });
                    if (dpq.q.empty()) return;
                    j = std::move(dpq.q.front());
                    dpq.q.pop_front();
                }
                std::string e2;
                const auto tDp = std::chrono::steady_clock::now();
                auto fatal = [&](const std::string &m) { fflush(stdout); fprintf(stderr, "%s\n", withHint(m).c_str()); fflush(stderr); _exit(1); };
                if (paired) {
                    PairDpCtx c{hix, so, *j->rs, ini, j->b, seedW, dpW, halfW, dpOut, unpairOut, 0u, j->cnt, alignmentType, insertLow, insertHigh, j->routing, j->chunkNo,
                                j->readBase, halfPar, deepPar, dpPar, newPar};
                    if (!pairDpStages(c, j->pr, cntW[1], cntW[2], cntW[3], cntW[4], e2)) fatal(e2);
                } else {
                    // DPForUnalignSingle2 (DV-DPForSingleReads.cu): seeding in batches of at most 256 Ki seeds, DP at the candidates, one record per run of
                    // results of the same read (DPSOutputThread, DV-DPfunctions.cu:1776), the rest as unaligned records
                    const ReadSet &rs = *j->rs;
                    if (j->hadSeedLengths && s3_batch_set_seed_lengths(j->b, nullptr) != S3_OK) fatal(s3_last_error());
                    std::vector<uint8_t> doneFlag(j->cnt, 0);
                    // (one run per read; the count of aligned reads is taken from the flags afterwards)
                    if (!singleDpRuns(hix, so, dpOut, j->b, seedW, dpW, rs.lengths.data(), j->todo, dpPar, alignmentType, [&](SamWriter &w, uint32_t k, std::vector<DpHit> &hits) {
                            w.singleDP(rs, k, hits.data(), hits.size(), dpPar.cutoffThreshold);
                            doneFlag[k] = 1;
                        }, e2)) fatal(e2);
                    for (uint32_t k : j->todo) cntW[0] += doneFlag[k];
                    // (records written through the FILE: under the file's lock, like a formatting call -- the other DP stage thread writes there too)
                    std::lock_guard<std::mutex> oneAtATime(outState(dpOut).use);
                    SamWriter w(hix, so, dpOut);
                    for (uint32_t k : j->todo) if (!doneFlag[k]) w.unaligned(rs, k);
                    w.flush();
                }
                workerSeconds[wk] += since(tDp);
                addChunkSeconds(3, since(tDp));
                recycle(std::move(j->rs));
                s3_batch *back = j->b;
                j.reset();
                { std::lock_guard<std::mutex> g(dpq.mu); dpq.freeBatches.push_back(back); }
                dpq.cv.notify_all();
            }
        });
    }
    // ---- pairs with the DP stages on: a third stage between the search and the DP stages.  The chunk loop uploads, searches and routes; the POST thread
    // fetches the chunk's pairings, formats and writes the properly paired records and hands the chunk to the DP stage threads -- a quarter second of host
    // work per ten million pairs during which the device used to wait for the next chunk's search (round 4) ------------------------------------------------
    struct { std::mutex mu; std::condition_variable cv; std::deque<std::unique_ptr<PairJob>> q; bool stop = false; } postq;
    std::thread postThread;
    uint64_t postAligned = 0;
    const bool postWanted = !skipSoap3 && !(getenv("S3DP_POST_THREAD") && atoi(getenv("S3DP_POST_THREAD")) == 0);
    // pairs with the DP stages off (-s): no DP stage threads, but the same hand-over -- two batches take turns between the chunk loop (upload, search,
    // pairing) and the post thread (fetch, format, write), which gives its batch back itself
    const bool postOnly = postWanted && paired && !enableDP && chunk && chunk->n;
    const bool postOn = (pipelinedDp && postWanted) || postOnly;
    if (postOnly) { stageBatches.push_back(b); dpq.freeBatches = stageBatches; }
    // single reads: what follows the search of a chunk -- the hits fetched, the records of the reads that have some formatted and written, the others listed
    // for the DP stage (`todo`, DP on).  Called by the chunk loop, or by the post thread while the loop searches the next chunk.
    auto finishSingle = [&](s3_batch *bb, const ReadSet &rs, uint32_t cnt, const std::vector<uint32_t> &seeds, uint64_t &al, std::vector<uint32_t> *todo, std::string &e) -> bool {
        std::vector<uint8_t> phase(cnt);
        std::vector<uint64_t> offs((size_t)cnt + 1);
        uint64_t nOcc = 0;
        const auto tF = std::chrono::steady_clock::now();
        if (s3_batch_fetch_alignments(bb, phase.data(), nullptr, offs.data(), nullptr, nullptr, nullptr, 0, &nOcc) != S3_OK) { e = s3_last_error(); return false; }
        addBatchTiming(bb);
        std::vector<uint32_t> pos(nOcc ? nOcc : 1);
        std::vector<uint8_t> strand(nOcc ? nOcc : 1), mis(nOcc ? nOcc : 1);
        if (s3_batch_fetch_alignments(bb, nullptr, nullptr, nullptr, pos.data(), strand.data(), mis.data(), pos.size(), nullptr) != S3_OK) { e = s3_last_error(); return false; }
        addChunkSeconds(1, since(tF));
        const auto tW = std::chrono::steady_clock::now();
        parallelFormat(hix, so, out, cnt, [&](SamWriter &w, uint32_t k, uint64_t &a1) {
            const size_t o = offs[k], m = offs[k + 1] - offs[k];
            if (alignmentType >= 3) {
                // unique best: exactly one hit, otherwise a no-answer record; random best: the first hit (hostKernel, CPUfunctions.cpp:1861-1930)
                if (m == 0 && enableDP) return;                                   // goes to the DP stage
                if (m == 0 || (alignmentType == 3 && m > 1)) w.unaligned(rs, k);
                else { a1++; w.singleAnswer(rs, k, pos[o], strand[o], mis[o]); }
                return;
            }
            if (m) a1++;
            else if (enableDP) return;                                            // goes to the DP stage
            w.singleEnd(rs, k, pos.data() + o, strand.data() + o, mis.data() + o, m, seeds.empty() ? 0u : seeds[k]);
        }, al);
        addChunkSeconds(2, since(tW));
        if (todo) for (uint32_t k = 0; k < cnt; k++) if (offs[k + 1] == offs[k]) todo->push_back(k);
        return true;
    };
    if (postOn) {
        s3_batch *nb = nullptr;                                                   // (one more batch takes turns: the one the post thread holds)
        if (s3_batch_create(ix, chunk->n == chunkReads ? chunkReads : chunk->n, chunk->wpq, &nb) != S3_OK) return dieJoin(s3_last_error());
        if (paired && pipelinedDp && s3_batch_set_grid(nb, 3) != S3_OK) return dieJoin(s3_last_error());
        stageBatches.push_back(nb);
        { std::lock_guard<std::mutex> g(dpq.mu); dpq.freeBatches.push_back(nb); }
        postThread = std::thread([&] {
            auto fatal = [&](const std::string &m) { fflush(stdout); fprintf(stderr, "%s\n", withHint(m).c_str()); fflush(stderr); _exit(1); };
            for (;;) {
                std::unique_ptr<PairJob> j;
                {
                    std::unique_lock<std::mutex> lk(postq.mu);
                    postq.cv.wait(lk, [&] { return !postq.q.empty() || postq.stop; });
                    if (postq.q.empty()) return;
                    j = std::move(postq.q.front());
                    postq.q.pop_front();
                }
                const ReadSet &rs = *j->rs;
                if (!paired) {
                    std::string e2;
                    if (!finishSingle(j->b, rs, j->cnt, j->seeds, postAligned, &j->todo, e2)) fatal(e2);
                    { std::lock_guard<std::mutex> g(dpq.mu); dpq.q.push_back(std::move(j)); }
                    dpq.cv.notify_all();
                    continue;
                }
                const uint32_t np = j->cnt / 2;
                std::vector<s3_pair_result> &pr = j->pr;
                pr.resize(np ? np : 1);
                std::vector<uint64_t> xoff((size_t)np + 1, 0);
                std::vector<s3_pair_xa> xa(1);
                uint64_t nXa = 0;
                if (!enableDP) {
                    // DP stages off: the pairs without a proper pairing are written with their ends' own hits
                    std::vector<uint64_t> ooff((size_t)j->cnt + 1, 0);
                    uint64_t nOcc = 0;
                    const auto tF0 = std::chrono::steady_clock::now();
                    if (s3_batch_fetch_pairs(j->b, pr.data(), xoff.data(), nullptr, 0, &nXa, ooff.data(), nullptr, nullptr, nullptr, 0, &nOcc) != S3_OK) fatal(s3_last_error());
                    addBatchTiming(j->b);
                    xa.resize(nXa ? nXa : 1);
                    std::vector<uint32_t> pos(nOcc ? nOcc : 1);
                    std::vector<uint8_t> strand(nOcc ? nOcc : 1), mis(nOcc ? nOcc : 1);
                    if (s3_batch_fetch_pairs(j->b, nullptr, nullptr, xa.data(), xa.size(), nullptr, nullptr, pos.data(), strand.data(), mis.data(), pos.size(), nullptr) != S3_OK)
                        fatal(s3_last_error());
                    addChunkSeconds(1, since(tF0));
                    const auto tW0 = std::chrono::steady_clock::now();
                    parallelFormat(hix, so, out, np, [&](SamWriter &w, uint32_t p, uint64_t &al) {
                        const uint32_t r1 = 2 * p, r2 = r1 + 1;
                        if (pr[p].phase) {
                            if (!(alignmentType == 3 && pr[p].numMin != 1)) al += 2;
                            w.properPair(rs, r1, r2, pr[p], xa.data() + xoff[p], (size_t)(xoff[p + 1] - xoff[p]));
                        } else {
                            const size_t o1 = ooff[2 * p], n1 = ooff[2 * p + 1] - o1, o2 = ooff[2 * p + 1], n2 = ooff[2 * p + 2] - o2;
                            w.improperPair(rs, r1, r2, pos.data() + o1, strand.data() + o1, mis.data() + o1, n1, pos.data() + o2, strand.data() + o2, mis.data() + o2, n2,
                                           ini.maxHitsEachEndForPairing);
                        }
                    }, postAligned);
                    addChunkSeconds(2, since(tW0));
                    recycle(std::move(j->rs));
                    s3_batch *back = j->b;
                    j.reset();
                    { std::lock_guard<std::mutex> g(dpq.mu); dpq.freeBatches.push_back(back); }
                    dpq.cv.notify_all();
                    continue;
                }
                const auto tF = std::chrono::steady_clock::now();
                if (s3_batch_fetch_pairs(j->b, pr.data(), xoff.data(), nullptr, 0, &nXa, nullptr, nullptr, nullptr, nullptr, 0, nullptr) != S3_OK) fatal(s3_last_error());
                addBatchTiming(j->b);
                xa.resize(nXa ? nXa : 1);
                if (nXa && s3_batch_fetch_pairs(j->b, nullptr, nullptr, xa.data(), xa.size(), nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr) != S3_OK) fatal(s3_last_error());
                addChunkSeconds(1, since(tF));
                const auto tW = std::chrono::steady_clock::now();
                parallelFormat(hix, so, out, np, [&](SamWriter &w, uint32_t p, uint64_t &al) {
                    if (!pr[p].phase) return;                                      // goes to the DP stages
                    if (!(alignmentType == 3 && pr[p].numMin != 1)) al += 2;
                    w.properPair(rs, 2 * p, 2 * p + 1, pr[p], xa.data() + xoff[p], (size_t)(xoff[p + 1] - xoff[p]));
                }, postAligned);
                addChunkSeconds(2, since(tW));
                { std::lock_guard<std::mutex> g(dpq.mu); dpq.q.push_back(std::move(j)); }
                dpq.cv.notify_all();
            }
        });
    }
    uint64_t chunkNo = 0;
    for (; chunk && chunk->n; recycle(std::move(chunk)), chunk = nextChunk(), chunkNo++) {
        ReadSet &rs = *chunk;
        const uint32_t first = 0, cnt = rs.n;
        totalReads += cnt;
        (void)maxBatch;
        if (pipelinedDp || postOnly) {
            // a batch whose chunk is through its DP stages (or has not been used yet)
            const auto tWt = std::chrono::steady_clock::now();
            std::unique_lock<std::mutex> lk(dpq.mu);
            // (S3DP_SERIAL=1, a measurement aid: the next chunk's search starts when every earlier chunk is through all of its stages -- one host thread's
            //  kernels on the device at a time, so that a kernel trace shows each kernel's own duration and not its wait behind another thread's)
            static const bool serialStages = getenv("S3DP_SERIAL") && atoi(getenv("S3DP_SERIAL")) == 1;
            dpq.cv.wait(lk, [&] { return serialStages ? dpq.freeBatches.size() == stageBatches.size() : !dpq.freeBatches.empty(); });
            b = dpq.freeBatches.back();
            dpq.freeBatches.pop_back();
            addChunkSeconds(4, since(tWt));
        }
        const auto tUp = std::chrono::steady_clock::now();
        if (s3_batch_upload(b, rs.queries.data() + (size_t)first * rs.wpq, rs.lengths.data() + first, cnt) != S3_OK) return dieJoin(s3_last_error());
        addChunkSeconds(5, since(tUp));
        // reads longer than 120 bases: only the first 100 are searched, the rest is verified on the device (alignment.cu:2473-2491)
        if (paired) {
            // pairs keep to the whole read (seedLengths is NULL for pairs, alignment.cu:1965-2013); DP stages are off (-s)
            const uint32_t np = cnt / 2;
            std::vector<s3_pair_result> pr(np ? np : 1);
            std::vector<uint64_t> xoff((size_t)np + 1, 0), ooff((size_t)cnt + 1, 0);
            uint64_t nXa = 0, nOcc = 0;
            std::vector<s3_pair_xa> xa(1);
            std::vector<uint32_t> pos(1);
            std::vector<uint8_t> strand(1), mis(1);
            s3_pair_routing routing{};
            const bool mapqMode = alignmentType == 1 || alignmentType == 2;
            const uint32_t routeMaxHit[2] = {(uint32_t)halfPar[0].maxHitNum, (uint32_t)halfPar[1].maxHitNum};
            if (skipSoap3) {
                // no BWT stage: nothing paired, no hits -- an empty ladder result (numMismatch 0 over nothing would still search; the batch is told directly)
                for (s3_pair_result &x : pr) { memset(&x, 0, sizeof x); x.min1 = x.min2 = 255; }
                if (s3_batch_no_pair_result(b) != S3_OK || s3_batch_route_pairs(b, routeMaxHit, mapqMode, &routing) != S3_OK) return dieJoin(s3_last_error());
            } else if (enableDP) {
                // DP stages on: the occurrence lists of the unpaired pairs stay on the device -- routed there (s3dp_route.hip), read there by the DP stages
                const auto tA = std::chrono::steady_clock::now();
                if (s3_batch_align_pair(b, (uint32_t)numMismatch, ini.maxHitsEachEndForPairing, alignmentType, (uint32_t)insertLow, (uint32_t)insertHigh,
                                        ini.peMaxOutputPerPair) != S3_OK) return dieJoin(s3_last_error());
                if (s3_batch_route_pairs(b, routeMaxHit, mapqMode, &routing) != S3_OK) return dieJoin(s3_last_error());
                addChunkSeconds(0, since(tA));
                if (postOn) {
                    std::unique_ptr<PairJob> j(new PairJob());
                    j->b = b; j->cnt = cnt; j->routing = routing; j->chunkNo = chunkNo; j->readBase = totalReads - cnt;
                    j->rs = std::move(chunk);                                      // (the loop's recycle() gets nothing: the DP stage thread gives the set back)
                    { std::lock_guard<std::mutex> g(postq.mu); postq.q.push_back(std::move(j)); }
                    postq.cv.notify_all();
                    continue;
                }
                const auto tF = std::chrono::steady_clock::now();
                if (s3_batch_fetch_pairs(b, pr.data(), xoff.data(), nullptr, 0, &nXa, nullptr, nullptr, nullptr, nullptr, 0, nullptr) != S3_OK) return dieJoin(s3_last_error());
                addBatchTiming(b);
                xa.resize(nXa ? nXa : 1);
                if (nXa && s3_batch_fetch_pairs(b, nullptr, nullptr, xa.data(), xa.size(), nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr) != S3_OK) return dieJoin(s3_last_error());
                addChunkSeconds(1, since(tF));
            } else {
                const auto tA = std::chrono::steady_clock::now();
                if (s3_batch_align_pair(b, (uint32_t)numMismatch, ini.maxHitsEachEndForPairing, alignmentType, (uint32_t)insertLow, (uint32_t)insertHigh,
                                        ini.peMaxOutputPerPair) != S3_OK) return dieJoin(s3_last_error());
                if (postOnly) {
                    addChunkSeconds(0, since(tA));
                    std::unique_ptr<PairJob> j(new PairJob());
                    j->b = b; j->cnt = cnt;
                    j->rs = std::move(chunk);                                      // (the loop's recycle() gets nothing: the post thread gives the set back)
                    { std::lock_guard<std::mutex> g(postq.mu); postq.q.push_back(std::move(j)); }
                    postq.cv.notify_all();
                    continue;
                }
                if (s3_batch_sync(b) != S3_OK) return dieJoin(s3_last_error());
                addChunkSeconds(0, since(tA));
                const auto tF = std::chrono::steady_clock::now();
                if (s3_batch_fetch_pairs(b, pr.data(), xoff.data(), nullptr, 0, &nXa, ooff.data(), nullptr, nullptr, nullptr, 0, &nOcc) != S3_OK) return dieJoin(s3_last_error());
                addBatchTiming(b);
                xa.resize(nXa ? nXa : 1); pos.resize(nOcc ? nOcc : 1); strand.resize(nOcc ? nOcc : 1); mis.resize(nOcc ? nOcc : 1);
                if (s3_batch_fetch_pairs(b, nullptr, nullptr, xa.data(), xa.size(), nullptr, nullptr, pos.data(), strand.data(), mis.data(), pos.size(), nullptr) != S3_OK)
                    return dieJoin(s3_last_error());
                addChunkSeconds(1, since(tF));
            }
            const auto tW = std::chrono::steady_clock::now();
            parallelFormat(hix, so, out, np, [&](SamWriter &w, uint32_t p, uint64_t &al) {
                const uint32_t r1 = first + 2 * p, r2 = r1 + 1;
                if (pr[p].phase) {
                    if (!(alignmentType == 3 && pr[p].numMin != 1)) al += 2;
                    w.properPair(rs, r1, r2, pr[p], xa.data() + xoff[p], (size_t)(xoff[p + 1] - xoff[p]));
                } else if (enableDP) {
                    return;                                                          // goes to the DP stages
                } else {
                    const size_t o1 = ooff[2 * p], n1 = ooff[2 * p + 1] - o1, o2 = ooff[2 * p + 1], n2 = ooff[2 * p + 2] - o2;
                    w.improperPair(rs, r1, r2, pos.data() + o1, strand.data() + o1, mis.data() + o1, n1, pos.data() + o2, strand.data() + o2, mis.data() + o2, n2,
                                   ini.maxHitsEachEndForPairing);
                }
            }, aligned);
            addChunkSeconds(2, since(tW));
            if (enableDP) {
                std::unique_ptr<PairJob> j(new PairJob());
                j->b = b; j->cnt = cnt;
                j->pr.swap(pr); j->routing = routing; j->chunkNo = chunkNo; j->readBase = totalReads - cnt;
                j->rs = std::move(chunk);                                          // (the loop's recycle() gets nothing: the worker gives the set back)
                { std::lock_guard<std::mutex> g(dpq.mu); dpq.q.push_back(std::move(j)); }
                dpq.cv.notify_all();
            }
            continue;
        }
        std::vector<uint32_t> seeds;
        if (maxReadLength > 120) {
            seeds.resize(cnt);
            for (uint32_t k = 0; k < cnt; k++) seeds[k] = rs.lengths[first + k] > 120 ? 100u : rs.lengths[first + k];
            if (s3_batch_set_seed_lengths(b, seeds.data()) != S3_OK) return dieJoin(s3_last_error());
        }
        const auto tA = std::chrono::steady_clock::now();
        if (s3_batch_align_single(b, (uint32_t)numMismatch, ini.maxOutputPerRead, alignmentType) != S3_OK) return dieJoin(s3_last_error());
        if (postOn) {
            // DP on: the post thread fetches and writes this chunk's hits while the next chunk is searched (the search returns with its last kernels still
            // under way: the wait for them is the post thread's)
            addChunkSeconds(0, since(tA));
            std::unique_ptr<PairJob> j(new PairJob());
            j->b = b; j->cnt = cnt; j->hadSeedLengths = !seeds.empty(); j->seeds.swap(seeds);
            j->rs = std::move(chunk);                                              // (the loop's recycle() gets nothing: the DP stage thread gives the set back)
            { std::lock_guard<std::mutex> g(postq.mu); postq.q.push_back(std::move(j)); }
            postq.cv.notify_all();
            continue;
        }
        std::unique_ptr<PairJob> j(enableDP ? new PairJob() : nullptr);
        std::string e2;
        if (!finishSingle(b, rs, cnt, seeds, aligned, j ? &j->todo : nullptr, e2)) return dieJoin(e2);
        if (j) {
            j->b = b; j->cnt = cnt; j->hadSeedLengths = !seeds.empty();
            j->rs = std::move(chunk);                                              // (the loop's recycle() gets nothing: the worker gives the set back)
            { std::lock_guard<std::mutex> g(dpq.mu); dpq.q.push_back(std::move(j)); }
            dpq.cv.notify_all();
        }
    }
    mark("last chunk searched");
    if (postThread.joinable()) {
        { std::lock_guard<std::mutex> g(postq.mu); postq.stop = true; }
        postq.cv.notify_all();
        postThread.join();
        aligned += postAligned;
    }
    if (postOnly) {
        b = stageBatches[0];
        for (size_t k = 1; k < stageBatches.size(); k++) s3_batch_free(stageBatches[k]);
    }
    if (pipelinedDp) {
        { std::lock_guard<std::mutex> g(dpq.mu); dpq.stop = true; }
        dpq.cv.notify_all();
        for (std::thread &t : dpWorkers) if (t.joinable()) t.join();
        for (int w = 0; w < nDpWorkers; w++) {
            dpAlignedReads += workerCounts[w][0]; defaultDpPairs += workerCounts[w][1]; newDefaultPairs += workerCounts[w][2]; deepDpPairs += workerCounts[w][3];
            leftoverPairs += workerCounts[w][4];
            dpStageSeconds += workerSeconds[w];
        }
        mark("DP stage threads done");
        b = stageBatches[0];
        for (size_t k = 1; k < stageBatches.size(); k++) s3_batch_free(stageBatches[k]);
    }
    auto closeOut = [&](FILE *f) { drainOutput(f); if (!bamClose(f)) { fprintf(stderr, "write error on the output file\n"); fflush(stdout); fflush(stderr); _exit(1); } fclose(f); };
    closeOut(out);
    mark("first output file closed");
    if (streamFreer.joinable()) { streamFreer.join(); mark("stream freed"); }
    if (dpOut) {
        closeOut(dpOut);
        mark("chunk loop done");
        if (paired) {
            closeOut(unpairOut);
            dpAlignedReads = 2 * (defaultDpPairs + newDefaultPairs + deepDpPairs);
            for (int w = 0; w < nDpWorkers; w++) { addDpTotals(halfSet[w]); s3_dp_free(halfSet[w]); }
        }
        aligned += dpAlignedReads;
        for (int w = 0; w < nDpWorkers; w++) { addDpTotals(dpSet[w]); s3_dp_free(dpSet[w]); s3_batch_free(seedSet[w]); }
    }
    if (b) s3_batch_free(b);
    mark("aligners and batches freed");
    uint64_t *C = counters.data() + (size_t)member * C_COUNTERS;
    C[C_ALIGNED] = aligned; C[C_READS] = totalReads; C[C_DP_READS] = dpAlignedReads; C[C_DEFAULT_DP] = defaultDpPairs; C[C_NEW_DEFAULT_DP] = newDefaultPairs;
    C[C_DEEP_DP] = deepDpPairs; C[C_LEFTOVER] = leftoverPairs;
    memberDpSeconds[(size_t)member] = dpStageSeconds;
    return 0;
    };   // runMember
    {
        std::vector<std::thread> mt;
        for (int m = 1; m < nMembers; m++) mt.emplace_back([&, m] { runMember(m); });
        runMember(0);
        for (auto &t : mt) t.join();
    }
    // the one exchange step of the path: the counters of the devices summed (one ncclAllReduce over the group, s3_group_allreduce)
    if (s3_group_allreduce(grp, counters.data(), C_COUNTERS) != S3_OK) return dieJoin(s3_last_error());
    const uint64_t aligned = counters[C_ALIGNED], totalReads = counters[C_READS];
    double dpStageSeconds = 0;
    for (double x : memberDpSeconds) dpStageSeconds = std::max(dpStageSeconds, x);
    if (enableDP) {
        if (paired)
            printf("[Main] Number of pairs aligned by new default DP: %llu\n[Main] Number of pairs aligned by DP: %llu\n[Main] Number of pairs aligned by deep DP: %llu\n"
                   "[Main] Pairs left without a proper pairing: %llu\n[Main] DP stages time : %9.4f seconds (routing %.3f, default DP %.3f, new default DP %.3f, deep DP %.3f, leftovers %.3f)\n",
                   (unsigned long long)counters[C_NEW_DEFAULT_DP], (unsigned long long)counters[C_DEFAULT_DP], (unsigned long long)counters[C_DEEP_DP],
                   (unsigned long long)counters[C_LEFTOVER], dpStageSeconds, pairStageSeconds(0), pairStageSeconds(1), pairStageSeconds(2), pairStageSeconds(3), pairStageSeconds(4));
        else printf("[Main] Number of reads aligned by DP: %llu\n", (unsigned long long)counters[C_DP_READS]);
    }
    // device time of the run (HIP events): the BWT stages with their search kernels, and every launch of the DP kernels with the cells they filled
    if (getenv("S3DP_VERBOSE")) {
        std::lock_guard<std::mutex> g(g_fmtMu);
        fprintf(stderr, "[s3dp] formatter: %.0f calls, waited %.3f s for buffers still being written, formatting threads %.3f s of wall clock (%.3f s summed over %d threads)\n",
                g_fmtSeconds[3], g_fmtSeconds[0], g_fmtSeconds[1], g_fmtSeconds[2], hostThreads());
    }
    if (!paired && !enableDP) printf("[Main] Stream loop time : feeder waited %.3f for the loader, %.3f for a free slot, copied chunks in %.3f; writer waited %.3f for results, formatted + wrote %.3f seconds\n",
                                     g_streamSeconds[0], g_streamSeconds[1], g_streamSeconds[2], g_streamSeconds[3], g_streamSeconds[4]);
    if (paired) printf("[Main] Chunk loop time : alignment calls %.3f, fetching results %.3f, formatting + writing %.3f, DP stages %.3f seconds (on a thread of their own; the loop waited %.3f for them), uploads %.3f\n",
                       g_chunkSeconds[0], g_chunkSeconds[1], g_chunkSeconds[2], g_chunkSeconds[3], g_chunkSeconds[4], g_chunkSeconds[5]);
    printf("[Main] Device time : BWT stages %.1f ms (search kernels %.1f ms)", g_deviceMs[0], g_deviceMs[1]);
    if (enableDP) printf(", DP kernels %.1f ms, %llu problems, %llu cells, %.1f GCUPS", g_deviceMs[2], (unsigned long long)g_dpProblems, (unsigned long long)g_dpCells,
                         g_deviceMs[2] > 0 ? (double)g_dpCells / (g_deviceMs[2] * 1e-3) / 1e9 : 0.0);
    printf("\n");
    if (nMembers > 1) printf("[Main] Devices: %d (index replicated, chunks of %u reads dealt as the devices come free, counters summed by one all-reduce)\n", nMembers, chunkReads);
    printf("[Main] Alignment + output time : %9.4f seconds\n[Main] Total elapsed time : %9.4f seconds\n", since(tAlign), since(t0));
    printf("[Main] Overall number of reads aligned: %llu\n", (unsigned long long)aligned);
    printf("[Main] Overall number of unaligned reads: %llu\n", (unsigned long long)(totalReads - aligned));
    printf("[Main] Read load time : %9.4f seconds (loader thread, overlapped)\n", loaded.seconds);
    loader.join();
    if (!loaded.err.empty()) return die(loaded.err);
    FILE *done = fopen((prefix + ".done").c_str(), "w");
    if (done) fclose(done);
    return 0;
}

// single / pair: one job.  single-multi / pair-multi <index> <list file> [options]: one job per line of the list, the index loaded once
// (loadMultiInputFile, IniParam.cpp:28-170; README.md "Multiple sets of ..."): columns separated by tabs --
//   single-multi: read file, output prefix [, read group ID, sample name [, read group options]]
//   pair-multi:   read file 1, read file 2, minimum insert, maximum insert, output prefix [, read group ID, sample name [, options]]
//   with -bam a pair is ONE file: read file, minimum insert, maximum insert, output prefix, ...
int main(int argc, char **argv)
{
    // sixteen hardware queues instead of the HIP runtime's four, before its first call (s3dp_api.hip: batches in flight on their own streams otherwise wait
    // for each other); a value the user set is left alone
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    SharedIndex shared;
    int rc = 0;
    const bool multi = argc >= 2 && (!strcmp(argv[1], "single-multi") || !strcmp(argv[1], "pair-multi"));
    if (!multi) rc = runJob(argc, argv, shared);
    else {
        if (argc < 4) return die("usage: soap3-dp single-multi|pair-multi <index> <list file> [options]");
        const bool paired = !strcmp(argv[1], "pair-multi");
        bool readBam = false;
        for (int i = 4; i < argc; i++) if (!strcmp(argv[i], "-bam") || !strcmp(argv[i], "-BAM")) readBam = true;
        FILE *f = fopen(argv[3], "r");
        if (!f) return die(std::string("cannot open the list file ") + argv[3]);
        std::vector<std::vector<std::string>> lines;
        char buf[8192];
        while (fgets(buf, sizeof buf, f)) {
            std::vector<std::string> cols;
            std::string cur;
            for (const char *p = buf; *p && *p != '\n' && *p != '\r'; p++) { if (*p == '\t') { cols.push_back(cur); cur.clear(); } else cur.push_back(*p); }
            cols.push_back(cur);
            while (!cols.empty() && cols.back().empty()) cols.pop_back();
            if (!cols.empty()) lines.push_back(cols);
        }
        fclose(f);
        for (size_t ln = 0; ln < lines.size() && rc == 0; ln++) {
            const std::vector<std::string> &c = lines[ln];
            const size_t nFiles = paired && !readBam ? 2 : 1, need = nFiles + (paired ? 2 : 0) + 1;
            if (c.size() < need) return die("line " + std::to_string(ln + 1) + " of " + argv[3] + ": too few columns");
            std::vector<std::string> a = {argv[0], paired ? "pair" : "single", argv[2]};
            for (size_t k = 0; k < nFiles; k++) a.push_back(c[k]);
            for (int i = 4; i < argc; i++) a.push_back(argv[i]);               // the options of the command line hold for every job
            size_t k = nFiles;
            if (paired) { a.push_back("-v"); a.push_back(c[k++]); a.push_back("-u"); a.push_back(c[k++]); }
            a.push_back("-o"); a.push_back(c[k++]);
            if (k < c.size()) { a.push_back("-D"); a.push_back(c[k++]); }
            if (k < c.size()) { a.push_back("-A"); a.push_back(c[k++]); }
            if (k < c.size()) { a.push_back("-R"); a.push_back(c[k++]); }
            std::vector<char *> av;
            for (std::string &x : a) av.push_back(&x[0]);
            printf("[Main] ---- job %zu of %zu: %s\n", ln + 1, lines.size(), c[0].c_str());
            rc = runJob((int)av.size(), av.data(), shared);
        }
    }
    // (giving 200 GB of device memory back takes seconds -- hipFree of a 50 GB table: 0.9 s -- and ending the process without it only moves the time into the
    //  start of the next process on the device: tools/exit_fast_probe.sh, round 6)
    if (shared.grp) s3_group_free(shared.grp);
    return rc;
}
