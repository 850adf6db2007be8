// s3dp_seed.hip -- seeding for the DP stages (s3_seed_candidates): seeds are packed from the resident reads into a second batch, searched with
// the ordinary search passes, and their hits sorted and merged into candidate positions on the device.
#include <hipcub/hipcub.hpp>
#include "s3dp_internal.h"

using namespace s3;
using namespace s3i;

// ---------------------------------------------------------------------------------------------
// Seeding for the DP stages of single reads: seeds cut from the reads at the tabulated offsets, searched exactly, the seeds
// without a hit once more with one mismatch, every hit turned into an estimated read position, positions of one read and strand merged.
// Replaces, of the reference: SingleEndSeedingBatch::pack / packSeeds (DV-DPfunctions.cu:1058-1100), single_1_mismatch_alignment2
// (alignment.cu:1839-1894) + the hit limit of hostKernelSingle (CPUfunctions.cpp:2843-2876: a seed with more than maxHitNum hits is
// dropped), decodePositions (:1143-1219: sorted by read and position, the strands together -- see singleMergeKernel) and singleMerge (:1102-1141, positions closer than
// DPS_DIVIDE_GAP = 50 collapse into the first).  Everything runs on the device next to the resident reads; the seed batch is an
// ordinary s3_batch.  Pinned through the reference's own seeding engines in the oracle harness (tests/test_gpu_dp_pipeline.py); also tested against a restatement built on the pinned search oracle.
// ---------------------------------------------------------------------------------------------

__global__ void seedPackKernel(uint32_t nSeeds, const SeedDesc *seeds, const uint32_t *queries, uint32_t wpq, uint32_t *seedQueries, uint32_t seedWpq,
                               uint32_t *seedLens)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (uint64_t)nSeeds * seedWpq) return;
    const uint32_t j = (uint32_t)(g / seedWpq), w = (uint32_t)(g % seedWpq);
    const SeedDesc d = seeds[j];
    const uint32_t *src = queries + (size_t)(d.readId >> 5) * 32u * wpq + (d.readId & 31u);
    uint32_t out = 0;
    for (uint32_t c = 0; c < 16u; c++) {
        const uint32_t i = w * 16u + c;
        if (i < d.len) {
            const uint32_t q = d.off + i;
            out |= ((src[(size_t)(q >> 4) * 32u] >> ((q & 15u) * 2u)) & 3u) << (2u * c);
        }
    }
    seedQueries[(size_t)(j >> 5) * 32u * seedWpq + (j & 31u) + (size_t)w * 32u] = out;
    if (w == 0) seedLens[j] = d.len;
}

// state per seed after a pass: 1 = hits kept, 0 = none / dropped; `retry` marks seeds without any hit (second pass with one mismatch)
__global__ void seedClassifyKernel(uint32_t n, const uint32_t *active, const uint32_t *totalOcc, uint32_t maxHit, uint8_t *state, uint8_t *retry, uint8_t *status)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = totalOcc[i], j = active ? active[i] : i;
    state[i] = (t > 0 && t <= maxHit) ? 1 : 0;
    retry[i] = t == 0;
    status[j] = t == 0 ? 0 : (t <= maxHit ? 1 : 4);
}

// estimated read position of every seed hit (MC_EstimatedPos, DV-DPfunctions.cu:1163): '+': pos - offset, '-': pos + seedLen + offset - readLen;
// key = strand << 63 | readID << 32 | position, the order decodePositions sorts into
__global__ void seedKeysKernel(uint32_t nSeeds, const SeedDesc *seeds, const uint32_t *readLens, const uint32_t *occStart, const uint32_t *occCount,
                               const uint32_t *outOffsets, const uint32_t *occPos, const uint32_t *occMeta, uint32_t strandBit, uint64_t *keys)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nSeeds) return;
    const SeedDesc d = seeds[j];
    const uint32_t cnt = occCount[j], src = occStart[j], dst = outOffsets[j], readLen = readLens[d.readId];
    for (uint32_t k = 0; k < cnt; k++) {
        const uint32_t pos = occPos[src + k], strand = occMeta[src + k] & 0xFFu;
        const uint32_t est = strand == 1u ? pos - d.off : pos + d.len + d.off - readLen;
        // strandBit 0xFFFFFFFF (the single-end DP): the strand is no part of the order -- (read, position), the strand rides in the lowest bit
        if (strandBit == 0xFFFFFFFFu) keys[dst + k] = (uint64_t)d.keyHi << 33 | (uint64_t)est << 1 | (strand == 2u ? 1u : 0u);
        else keys[dst + k] = (uint64_t)(d.keyHi | (strand == 2u ? 1u << strandBit : 0u)) << 32 | est;
    }
}

// ---- candidate merges on the device --------------------------------------------------------------------------------------------
// singleMerge (DV-DPfunctions.cu:1102-1141): inside one READ's run of the sorted positions the first position is kept, then every position more than
// DPS_DIVIDE_GAP = 50 beyond the last KEPT one -- a chain along the run, so the lane that finds itself at the head of a run walks it (runs are a few
// seeds' worth of hits); a counting walk, a scan, a writing walk.
// The order is (read, position) with BOTH strands in one run, a kept position keeping its own strand: decodePositions (:1214-1216) sorts by position, by read
// and then by strand -- but that third pass (MC_RadixSort_8_8, DV-DPfunctions.h:90: ONE round, from the array into the auxiliary array) leaves its result in
// the buffer that is freed on the next line; what singleMerge walks is the order of the second pass.  Until round 6 this file sorted by (strand, read,
// position) as the source reads: a read with DP hits on both strands then came as two runs, its result list '+' first instead of by position -- the
// primary record of such a read differed from the reference's whenever the two strands' best scores tied (found by bench.py's DP parity leg on the
// benchmark genome, whose repeat families have inverted copies; tests/test_gpu_cli_vs_oracle_scale.py::test_reads_with_dp_hits_on_both_strands).
// key = read << 33 | position << 1 | (strand == '-')
__global__ void singleMergeKernel(const uint64_t *keys, uint32_t n, const uint32_t *offsets, uint32_t *counts, s3_seed_candidate *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t run = keys[i] >> 33;
    if (i && (keys[i - 1] >> 33) == run) { if (!out) counts[i] = 0; return; }
    uint32_t prev = (uint32_t)(keys[i] >> 1), kept = 1;
    const uint32_t at = out ? offsets[i] : 0u;
    if (out) out[at] = s3_seed_candidate{(uint32_t)run, prev, (uint8_t)((keys[i] & 1u) ? 2 : 1), {0, 0, 0}};
    for (uint32_t k = i + 1; k < n && (keys[k] >> 33) == run; k++) {
        const uint32_t cur = (uint32_t)(keys[k] >> 1);
        if ((uint64_t)prev + 50u < cur) {
            if (out) out[at + kept] = s3_seed_candidate{(uint32_t)run, cur, (uint8_t)((keys[k] & 1u) ? 2 : 1), {0, 0, 0}};
            kept++; prev = cur;
        }
    }
    if (!out) counts[i] = kept;
}

// pairEndMerge (DV-DPfunctions.cu:2780-2883) for one orientation: the '+' positions of one end (the LEFT run: keys [l0, l1)) against the '-' positions of the
// other end (the RIGHT run: keys [r0, r1)), both sorted by (pair, position), a key = (... | pair's first read) << 32 | position.  Per pair: the left positions
// compressed by DP2_DIVIDE_GAP (a position more than 50 beyond the last KEPT one is kept), then the reference's two-pointer sweep -- a left position takes the
// first right position at or beyond readLoc + lengthLow (the right pointer never goes back: the bounds rise with the left positions) if that is not beyond
// readLoc + lengthHigh.  The lane at the head of a pair's left run walks it.  out == NULL: count into counts[2 * pairRank + leftIsMate]; else write behind
// offsets[...].  pairRank = the pair's place in pairFirstReads (ascending), so the candidates come out sorted by the left read, as the reference sorts them.
struct PairCandDev { uint32_t posLeft, posRight, readIDLeft; };
#define S3_PAIRMERGE_WAVE 24u       // left runs longer than this are walked by the whole wave of their head lane (S3DP_PAIRMERGE_WAVE; 0xFFFFFFFF: never)
// A pair whose end sits in a repeat family comes with thousands of positions (the new default DP's given hits: up to the hit limit per end), and the lane at
// the head of such a run kept 63 lanes and the launch waiting while it made two dependent loads per position.  The sweep is not as serial as it looks: which
// left positions survive the compression is a chain (the next kept one is the first more than 50 beyond the last kept one) -- solved for 64 sorted
// positions at a time in registers with a ballot per kept position --, and a kept position's partner is the lower bound of readLoc + lengthLow in the right
// run (what the never-returning right pointer arrives at), found by every kept lane on its own.  Candidates leave in left order through a ballot prefix.
__global__ void pairMergeKernel(const uint64_t *keys, uint32_t l0, uint32_t l1, uint32_t r0, uint32_t r1, uint32_t leftIsMate, const uint32_t *readLengths,
                                const uint32_t *pairFirstReads, uint32_t nPairs, int insertLow, int insertHigh, uint32_t waveMin, uint32_t *counts,
                                const uint32_t *offsets, PairCandDev *out)
{
    const uint32_t i = l0 + blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    auto idOf = [&](uint32_t k) -> uint32_t { return (uint32_t)(keys[k] >> 32) & 0x3FFFFFFFu; };
    uint32_t id = 0, rp = 0, slot = 0, at = 0;
    int lengthLow = 0, lengthHigh = 0;
    bool head = false;
    if (i < l1) {
        id = idOf(i);
        head = !(i > l0 && idOf(i - 1) == id);
    }
    if (head) {
        // the pair's right run
        uint32_t lo = r0, hi = r1;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (idOf(mid) < id) lo = mid + 1; else hi = mid; }
        rp = lo;
        head = rp < r1 && idOf(rp) == id;
    }
    bool big = false;
    if (head) {
        uint32_t a = 0, b = nPairs;
        while (a < b) { const uint32_t mid = (a + b) >> 1; if (pairFirstReads[mid] < id) a = mid + 1; else b = mid; }
        slot = 2u * a + leftIsMate;
        const int readLength = (int)readLengths[id];                                // readLengths[readID] with the pair's first read, as the reference
        const int margin = readLength > 100 ? readLength >> 2 : 25;                // DP2_MARGIN
        lengthLow = insertLow - readLength - margin;
        if (lengthLow < 0) lengthLow = 0;
        lengthHigh = insertHigh - readLength + margin;
        at = out ? offsets[slot] : 0u;
        big = waveMin != 0xFFFFFFFFu && (uint64_t)i + waveMin < l1 && idOf(i + waveMin) == id;
    }
    if (head && !big) {
        uint32_t n = 0, prev = 0;
        bool first = true;
        for (uint32_t lp = i; lp < l1 && idOf(lp) == id; lp++) {
            const uint32_t readLoc = (uint32_t)keys[lp];
            if (!first && !(prev + 50u < readLoc)) continue;                        // MC_Compress
            first = false; prev = readLoc;
            while (rp < r1 && idOf(rp) == id && readLoc + (uint32_t)lengthLow > (uint32_t)keys[rp]) rp++;
            if (rp >= r1 || idOf(rp) != id) break;                                 // the right run is used up: nothing pairs with the later left positions either
            const uint32_t mateLoc = (uint32_t)keys[rp];
            if (readLoc + (uint32_t)lengthHigh < mateLoc) continue;
            if (out) out[at + n] = PairCandDev{readLoc, mateLoc, id + leftIsMate};
            n++;
        }
        if (!out) counts[slot] = n;
    }
    // the long runs of this wave's heads, one after the other, by all of its lanes
    unsigned long long todo = __ballot(head && big);
    while (todo) {
        const int h = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const uint32_t hId = (uint32_t)__shfl((int)id, h), hI = (uint32_t)__shfl((int)i, h), hAt = (uint32_t)__shfl((int)at, h), hSlot = (uint32_t)__shfl((int)slot, h);
        const uint32_t hLow = (uint32_t)__shfl(lengthLow, h), hHigh = (uint32_t)__shfl(lengthHigh, h);
        uint32_t rLo = (uint32_t)__shfl((int)rp, h);
        // ends of the two runs (upper bounds of the pair's id)
        uint32_t lo = hI, hi = l1;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (idOf(mid) <= hId) lo = mid + 1; else hi = mid; }
        const uint32_t lEnd = lo;
        lo = rLo; hi = r1;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (idOf(mid) <= hId) lo = mid + 1; else hi = mid; }
        const uint32_t rHi = lo;
        uint32_t n = 0, prev = 0;
        bool first = true;
        for (uint32_t c0 = hI; c0 < lEnd; c0 += 64u) {
            const uint32_t k = c0 + lane;
            const bool in = k < lEnd;
            const uint32_t readLoc = in ? (uint32_t)keys[k] : 0u;
            // the chain of kept positions inside these 64: the qualifying lanes are a suffix (the positions ascend), its first lane is kept
            bool kept = false;
            unsigned long long after = ~0ULL;                                       // lanes behind the last kept one
            for (;;) {
                const unsigned long long q = __ballot(in && (first || prev + 50u < readLoc)) & after;
                if (!q) break;
                const int w = __ffsll((long long)q) - 1;
                if ((int)lane == w) kept = true;
                prev = (uint32_t)__shfl((int)readLoc, w);
                first = false;
                after = w == 63 ? 0ULL : ~0ULL << (w + 1);
            }
            // the partner: the first right position at or beyond readLoc + lengthLow, if it is not beyond readLoc + lengthHigh
            // (the bounds rise with the left positions: the search starts where the previous 64 positions' first partner was found and gallops from there --
            //  in a tandem array the partner moves a few entries per left position, and thirteen dependent loads become three)
            uint32_t mateLoc = 0, a = rLo;
            bool cand = false;
            if (kept) {
                const uint32_t want = readLoc + hLow;
                uint32_t b = a, span = 1;
                while (b < rHi && (uint32_t)keys[b] < want) { a = b + 1u; b = span < rHi - b ? b + span : rHi; span <<= 1; }
                while (a < b) { const uint32_t mid = (a + b) >> 1; if ((uint32_t)keys[mid] < want) a = mid + 1; else b = mid; }
                if (a < rHi) { mateLoc = (uint32_t)keys[a]; cand = !(readLoc + hHigh < mateLoc); }
            }
            {
                const unsigned long long km = __ballot(kept);
                if (km) rLo = (uint32_t)__shfl((int)a, __ffsll((long long)km) - 1);          // the lowest kept position's bound: no later one lies before it
            }
            const unsigned long long cm = __ballot(cand);
            if (cand && out) out[hAt + n + (uint32_t)__popcll(cm & ((1ULL << lane) - 1ULL))] = PairCandDev{readLoc, mateLoc, hId + leftIsMate};
            n += (uint32_t)__popcll(cm);
        }
        if (!out && (int)lane == h) counts[hSlot] = n;
    }
}

// keys of occurrences given from outside (the new default DP: the BWT stage's hits of one end), in the format of the seed keys
__global__ void givenKeysKernel(const s3_occ *occ, uint32_t n, uint64_t *keys)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t hi = (occ[i].readID & 1u) << 31 | (occ[i].strand == 2 ? 1u << 30 : 0u) | (occ[i].readID & ~1u);
    keys[i] = (uint64_t)hi << 32 | occ[i].pos;
}

// Search the seeds and return their hits as sorted 64-bit keys: (keyHi | strand bit) << 32 | estimated read position.  devKeysOut != NULL: the sorted keys
// stay on the device (the caller gives *devKeysOut back with scratchFree) and keysOut stays empty.
int s3i::seedSearchKeys(s3_batch *reads, s3_batch *seedBatch, const std::vector<SeedDesc> &seeds, uint32_t maxHitNum, uint32_t strandBit,
                        std::vector<uint64_t> &keysOut, std::vector<uint8_t> &statusOut, uint64_t **devKeysOut, uint32_t *nKeysOut)
{
    HIPCHK(hipSetDevice(reads->ix->dev));
    const uint32_t nSeeds = (uint32_t)seeds.size();
    if (nSeeds > seedBatch->maxReads) return fail(S3_EINVAL, "s3_seed_candidates: seed batch too small");
    if (seedBatch->wpq * 16u < 70u) return fail(S3_EINVAL, "s3_seed_candidates: the seed batch needs wordPerQuery >= 8");
    s3_batch *b = seedBatch;
    hipStream_t st = b->stream;
    SeedDesc *dSeeds = nullptr;
    uint8_t *dRetry = nullptr, *dStatus = nullptr;
    uint64_t *dKeys = nullptr, *dKeysOut = nullptr;
    char *dSortTmp = nullptr;
    std::vector<void *> tmp;
    auto talloc = [&](void **p, size_t bytes) -> int {
        hipError_t e = scratchAlloc(p, bytes ? bytes : 16, st);
        if (e != hipSuccess) return fail(S3_ENOMEM, std::string("hipMalloc(seeding): ") + hipGetErrorString(e));
        tmp.push_back(*p);
        return S3_OK;
    };
    auto cleanup = [&]() { for (void *p : tmp) if (p) scratchFree(p, st); };
    int rc;
    if ((rc = talloc((void **)&dSeeds, (size_t)nSeeds * sizeof(SeedDesc))) || (rc = talloc((void **)&dRetry, nSeeds)) || (rc = talloc((void **)&dStatus, nSeeds))) { cleanup(); return rc; }
#define SCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(S3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    SCHK(hipMemcpyAsync(dSeeds, seeds.data(), (size_t)nSeeds * sizeof(SeedDesc), hipMemcpyHostToDevice, st));
    {
        const size_t roundUp = ((size_t)nSeeds + 31) / 32 * 32;
        SCHK(hipMemsetAsync(b->dQueries, 0, roundUp * b->wpq * 4, st));
        const uint64_t threads = (uint64_t)nSeeds * b->wpq;
        if (threads) hipLaunchKernelGGL(seedPackKernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, nSeeds, dSeeds, reads->dQueries, reads->wpq,
                                        b->dQueries, b->wpq, b->dLens);
        launchOrientReads(b->dQueries, b->dLens, b->dReadsT, nSeeds, b->wpq, b->wpq, st);
        SCHK(hipMemsetAsync(b->dWasHeavy, 0, b->maxReads, st));
        b->unitGroup = 1;
        b->n = nSeeds; b->longMode = false; b->lastKind = 0;
        b->uniformLen = nSeeds ? seeds[0].len : 0;
        for (uint32_t j = 1; j < nSeeds; j++) if (seeds[j].len != seeds[0].len) { b->uniformLen = 0; break; }
        b->fullUniformLen = b->uniformLen;
    }
    b->searchLaunches = 0; b->evSearchUsed = 0; b->evCeUsed = 0; b->occUsed = 0; b->recordsTotal = 0;
    SCHK(hipMemsetAsync(b->dFlags, 0, 4, st));
    SCHK(hipMemsetAsync(b->dOccStart, 0, (size_t)(nSeeds ? nSeeds : 1) * 4, st));
    SCHK(hipMemsetAsync(b->dOccCount, 0, ((size_t)nSeeds + 1) * 4, st));
    SCHK(hipMemsetAsync(b->dNumSel, 0, 16, st));
    const LongArgs noLong{false, 0, 0, 0, 0, 0};
    const uint32_t *active = nullptr;
    uint32_t nA = nSeeds;
    for (uint32_t mm = 0; mm <= 1 && nA > 0; mm++) {                       // exact first, then one mismatch for the seeds without a hit
        if ((rc = searchPass(b, active, nA, mm, 0xFFFFFFFFu))) { cleanup(); return rc; }
        hipLaunchKernelGGL(seedClassifyKernel, dim3((nA + 255) / 256), dim3(256), 0, st, nA, active, b->dTotalOcc, maxHitNum, b->dState, dRetry, dStatus);
        if ((rc = decodeAndPublish(b, active, nA, 1, 0, 0xFFFFFFFFu, 0, noLong))) { cleanup(); return rc; }
        if (mm == 1) break;
        const uint32_t *idsIn = active;
        if (!idsIn) { launchIota(b->dIota, nA, st); idsIn = b->dIota; }
        size_t tb = b->tempBytes;
        SCHK(hipcub::DeviceSelect::Flagged(b->dTemp, tb, idsIn, dRetry, b->dActive[0], b->dNumSel, (int)nA, st));
        uint32_t nNext = 0;
        SCHK(hipMemcpyAsync(&nNext, b->dNumSel, 4, hipMemcpyDeviceToHost, st));
        SCHK(hipStreamSynchronize(st));
        active = b->dActive[0];
        nA = nNext;
    }
    // all kept hits -> keys -> sort -> merge
    size_t tb = b->tempBytes;
    SCHK(hipcub::DeviceScan::ExclusiveSum(b->dTemp, tb, b->dOccCount, b->dOutOffsets, (int)(nSeeds + 1), st));
    uint32_t total = 0;
    SCHK(hipMemcpyAsync(&total, b->dOutOffsets + nSeeds, 4, hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    keysOut.clear();
    if (total) {
        size_t sortBytes = 0;
        SCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, sortBytes, dKeys, dKeysOut, (int)total, 0, 64, st));
        if ((rc = talloc((void **)&dKeys, (size_t)total * 8)) || (rc = talloc((void **)&dKeysOut, (size_t)total * 8))||
            (rc = talloc((void **)&dSortTmp, sortBytes))) { cleanup(); return rc; }
        hipLaunchKernelGGL(seedKeysKernel, dim3((nSeeds + 255) / 256), dim3(256), 0, st, nSeeds, dSeeds, reads->dLens, b->dOccStart, b->dOccCount, b->dOutOffsets,
                           b->dOccPos, b->dOccMeta, strandBit, dKeys);
        SCHK(hipcub::DeviceRadixSort::SortKeys(dSortTmp, sortBytes, dKeys, dKeysOut, (int)total, 0, 64, st));
        if (devKeysOut) {
            for (void *&p : tmp) if (p == (void *)dKeysOut) p = nullptr;        // handed to the caller
            *devKeysOut = dKeysOut;
        } else {
            keysOut.resize(total);
            SCHK(hipMemcpyAsync(keysOut.data(), dKeysOut, (size_t)total * 8, hipMemcpyDeviceToHost, st));
        }
        SCHK(hipStreamSynchronize(st));
    } else if (devKeysOut) *devKeysOut = nullptr;
    if (nKeysOut) *nKeysOut = total;
    statusOut.resize(nSeeds);
    if (nSeeds) SCHK(d2h(statusOut.data(), dStatus, nSeeds, st));
#undef SCHK
    cleanup();
    return checkFlags(b);
}

__global__ void runBoundsKernel(const uint64_t *keys, uint32_t n, uint32_t *bounds)
{
    const uint32_t t = threadIdx.x;                                            // 3 lanes: first key with (end, strand) bits >= 1, 2, 3
    if (t >= 3) return;
    const uint64_t want = (uint64_t)(t + 1u) << 62;
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid] < want) lo = mid + 1; else hi = mid; }
    bounds[t] = lo;
}

int s3i::pairMergeOnDevice(s3_batch *b, uint64_t *dKeys, uint32_t nKeys, const s3_occ *given, size_t nGiven, bool givenOnDevice, const uint32_t *pairFirstReads,
                           size_t nPairs, int insertLow, int insertHigh, std::vector<PairCandHost> &cands, std::vector<uint32_t> &perPair)
{
    static_assert(sizeof(PairCandHost) == sizeof(PairCandDev), "candidate layouts");
    cands.clear();
    perPair.assign(nPairs, 0);
    hipStream_t st = b->stream;
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *p : tmp) if (p) scratchFree(p, st); };
    auto talloc = [&](void **p, size_t bytes) -> int {
        hipError_t e = scratchAlloc(p, bytes ? bytes : 16, st);
        if (e != hipSuccess) { cleanup(); return fail(S3_ENOMEM, std::string("hipMalloc(pair merge): ") + hipGetErrorString(e)); }
        tmp.push_back(*p);
        return S3_OK;
    };
#define PCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(S3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    int rc;
    uint64_t *keys = dKeys;
    uint32_t n = nKeys;
    if (nGiven) {
        // the given occurrences join the seed positions: one list sorted by (end, strand, pair, position) again (decodePositions, DV-DPfunctions.cu:2896-2899)
        if (nGiven + (size_t)nKeys >= 0xFFFFFFF0ull) { cleanup(); return fail(S3_EOVERFLOW, "pair merge: too many positions"); }
        s3_occ *dOcc = nullptr;
        uint64_t *all = nullptr, *sorted = nullptr;
        char *sortTmp = nullptr;
        const uint32_t m = (uint32_t)(nKeys + nGiven);
        size_t sortBytes = 0;
        PCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, sortBytes, all, sorted, (int)m, 0, 64, st));
        if ((!givenOnDevice && (rc = talloc((void **)&dOcc, nGiven * sizeof(s3_occ)))) || (rc = talloc((void **)&all, (size_t)m * 8)) || (rc = talloc((void **)&sorted, (size_t)m * 8)) ||
            (rc = talloc((void **)&sortTmp, sortBytes))) return rc;
        if (!givenOnDevice) PCHK(hipMemcpyAsync(dOcc, given, nGiven * sizeof(s3_occ), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(givenKeysKernel, dim3((unsigned)((nGiven + 255) / 256)), dim3(256), 0, st, givenOnDevice ? given : dOcc, (uint32_t)nGiven, all);
        if (nKeys) PCHK(hipMemcpyAsync(all + nGiven, dKeys, (size_t)nKeys * 8, hipMemcpyDeviceToDevice, st));
        PCHK(hipcub::DeviceRadixSort::SortKeys(sortTmp, sortBytes, all, sorted, (int)m, 0, 64, st));
        keys = sorted; n = m;
    }
    if (n == 0 || nPairs == 0) { cleanup(); return S3_OK; }
    uint32_t *dBounds = nullptr, *dPairs = nullptr, *dCounts = nullptr, *dOffsets = nullptr;
    char *scanTmp = nullptr;
    const uint32_t nSlots = (uint32_t)(2 * nPairs);
    size_t tb = 0;
    PCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, dCounts, dOffsets, (int)(nSlots + 1), st));
    if ((rc = talloc((void **)&dBounds, 16)) || (rc = talloc((void **)&dPairs, nPairs * 4)) || (rc = talloc((void **)&dCounts, ((size_t)nSlots + 1) * 4)) ||
        (rc = talloc((void **)&dOffsets, ((size_t)nSlots + 1) * 4)) || (rc = talloc((void **)&scanTmp, tb + 16))) return rc;
    PCHK(hipMemcpyAsync(dPairs, pairFirstReads, nPairs * 4, hipMemcpyHostToDevice, st));
    PCHK(hipMemsetAsync(dCounts, 0, ((size_t)nSlots + 1) * 4, st));
    hipLaunchKernelGGL(runBoundsKernel, dim3(1), dim3(64), 0, st, keys, n, dBounds);
    uint32_t bnd[3] = {0, 0, 0};
    PCHK(hipMemcpyAsync(bnd, dBounds, 12, hipMemcpyDeviceToHost, st));
    PCHK(hipStreamSynchronize(st));
    // runs: [end 0 '+'] [end 0 '-'] [end 1 '+'] [end 1 '-'];  orientation 0: end 0 '+' (left) against end 1 '-' (right);  1: end 1 '+' against end 0 '-'
    const uint32_t runLo[4] = {0, bnd[0], bnd[1], bnd[2]}, runHi[4] = {bnd[0], bnd[1], bnd[2], n};
    const uint32_t waveMin = (uint32_t)envSize("S3DP_PAIRMERGE_WAVE", S3_PAIRMERGE_WAVE);      // (read per call: the tests switch it)
    auto launch = [&](uint32_t o, const uint32_t *offs, PairCandDev *out) {
        const uint32_t l = o ? 2u : 0u, r = o ? 1u : 3u;
        const uint32_t cnt = runHi[l] - runLo[l];
        if (!cnt || runHi[r] == runLo[r]) return;
        hipLaunchKernelGGL(pairMergeKernel, dim3((cnt + 255) / 256), dim3(256), 0, st, keys, runLo[l], runHi[l], runLo[r], runHi[r], o, b->dLens, dPairs, (uint32_t)nPairs,
                           insertLow, insertHigh, waveMin, dCounts, offs, out);
    };
    launch(0, nullptr, nullptr);
    launch(1, nullptr, nullptr);
    PCHK(hipcub::DeviceScan::ExclusiveSum(scanTmp, tb, dCounts, dOffsets, (int)(nSlots + 1), st));
    std::vector<uint32_t> offs((size_t)nSlots + 1);
    PCHK(hipMemcpyAsync(offs.data(), dOffsets, ((size_t)nSlots + 1) * 4, hipMemcpyDeviceToHost, st));
    PCHK(hipStreamSynchronize(st));
    const uint32_t total = offs[nSlots];
    for (size_t k = 0; k < nPairs; k++) perPair[k] = offs[2 * k + 2] - offs[2 * k];
    if (total) {
        PairCandDev *dOut = nullptr;
        if ((rc = talloc((void **)&dOut, (size_t)total * sizeof(PairCandDev)))) return rc;
        launch(0, dOffsets, dOut);
        launch(1, dOffsets, dOut);
        cands.resize(total);
        PCHK(hipMemcpyAsync(cands.data(), dOut, (size_t)total * sizeof(PairCandDev), hipMemcpyDeviceToHost, st));
        PCHK(hipStreamSynchronize(st));
    }
#undef PCHK
    cleanup();
    return S3_OK;
}

extern "C" int s3_seed_candidates(s3_batch *reads, s3_batch *seedBatch, const uint32_t *readLengths, const uint32_t *readIDs, size_t nReadIDs, int stage,
                                  uint32_t maxHitNum, s3_seed_candidate **candOut, size_t *nCandOut, uint8_t **seedStatusOut, size_t *nSeedsOut)
{
    if (!reads || !seedBatch || !readLengths || !candOut || !nCandOut) return fail(S3_EINVAL, "s3_seed_candidates: null argument");
    if (reads->ix != seedBatch->ix || reads == seedBatch) return fail(S3_EINVAL, "s3_seed_candidates: the seed batch must be a second batch on the same index");
    std::vector<SeedDesc> seeds;
    const size_t nSel = readIDs ? nReadIDs : reads->n;
    for (size_t k = 0; k < nSel; k++) {
        const uint32_t r = readIDs ? readIDs[k] : (uint32_t)k;
        if (r >= reads->n) return fail(S3_EINVAL, "s3_seed_candidates: read index out of range");
        int seedLength = 0, num = 0, pos[128];
        if (s3_seed_positions(stage, (int)readLengths[r], &seedLength, pos, &num) != S3_OK) return fail(S3_EINVAL, "s3_seed_candidates: bad stage");
        for (int i = 0; i < num; i++) {
            if (pos[i] < 0 || seedLength < 16 || (uint32_t)(pos[i] + seedLength) > readLengths[r]) continue;   // reads too short for the stage's seeds
            seeds.push_back(SeedDesc{r, (uint32_t)pos[i], (uint32_t)seedLength, r});
        }
    }
    std::vector<uint64_t> keys;
    std::vector<uint8_t> status;
    uint64_t *dKeys = nullptr;
    uint32_t nKeys = 0;
    int rc = seedSearchKeys(reads, seedBatch, seeds, maxHitNum, 0xFFFFFFFFu, keys, status, &dKeys, &nKeys);
    if (rc) return rc;
    // singleMerge on the device: count per run head, scan, write; the candidates come over, not the positions
    std::vector<s3_seed_candidate> out;
    if (nKeys) {
        hipStream_t st = seedBatch->stream;
        uint32_t *dCounts = nullptr, *dOffsets = nullptr;
        s3_seed_candidate *dOut = nullptr;
        char *dTmp = nullptr;
        size_t tb = 0;
        auto freeAll = [&]() { scratchFree(dKeys, st); scratchFree(dCounts, st); scratchFree(dOffsets, st); scratchFree(dOut, st); scratchFree(dTmp, st); };
#define MCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { freeAll(); return fail(S3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
        MCHK(scratchAlloc((void **)&dCounts, ((size_t)nKeys + 1) * 4, st));
        MCHK(scratchAlloc((void **)&dOffsets, ((size_t)nKeys + 1) * 4, st));
        MCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, dCounts, dOffsets, (int)(nKeys + 1), st));
        MCHK(scratchAlloc((void **)&dTmp, tb + 16, st));
        MCHK(hipMemsetAsync(dCounts + nKeys, 0, 4, st));
        hipLaunchKernelGGL(singleMergeKernel, dim3((nKeys + 255) / 256), dim3(256), 0, st, dKeys, nKeys, nullptr, dCounts, (s3_seed_candidate *)nullptr);
        MCHK(hipcub::DeviceScan::ExclusiveSum(dTmp, tb, dCounts, dOffsets, (int)(nKeys + 1), st));
        uint32_t total = 0;
        MCHK(hipMemcpyAsync(&total, dOffsets + nKeys, 4, hipMemcpyDeviceToHost, st));
        MCHK(hipStreamSynchronize(st));
        if (total) {
            MCHK(scratchAlloc((void **)&dOut, (size_t)total * sizeof(s3_seed_candidate), st));
            hipLaunchKernelGGL(singleMergeKernel, dim3((nKeys + 255) / 256), dim3(256), 0, st, dKeys, nKeys, dOffsets, dCounts, dOut);
            out.resize(total);
            MCHK(hipMemcpyAsync(out.data(), dOut, (size_t)total * sizeof(s3_seed_candidate), hipMemcpyDeviceToHost, st));
            MCHK(hipStreamSynchronize(st));
        }
#undef MCHK
        freeAll();
    }
    const size_t nSeeds = status.size();
    s3_seed_candidate *c = (s3_seed_candidate *)malloc(out.size() * sizeof(s3_seed_candidate) + 1);
    if (!c) return fail(S3_ENOMEM, "s3_seed_candidates: out of host memory");
    memcpy(c, out.data(), out.size() * sizeof(s3_seed_candidate));
    *candOut = c; *nCandOut = out.size();
    if (seedStatusOut) {
        uint8_t *sp = (uint8_t *)malloc(nSeeds + 1);
        if (!sp) { free(c); return fail(S3_ENOMEM, "s3_seed_candidates: out of host memory"); }
        memcpy(sp, status.data(), nSeeds);
        *seedStatusOut = sp;
    }
    if (nSeedsOut) *nSeedsOut = nSeeds;
    return S3_OK;
}

// one kernel of this file's code object, asked for its attributes: the runtime loads a code object at the first use of one of its kernels (preloadKernels, s3dp_kernels.hip)
void s3::preloadSeed() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)seedPackKernel); (void)hipGetLastError(); }
