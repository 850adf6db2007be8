// s3dp_pair.hip -- device side of the paired-end path with the DP stages off (`soap3-dp pair ... -s k`):
// the per-pair decisions of hostKernel's PAIR_END_READ branch (CPUfunctions.cpp:1980-2560) and the pairing of the two ends'
// occurrence lists (PEMappingOccurrences / PEMappingCore / PEStatsPEPairList, PEAlgnmt.cpp:201-291,480-560,777-832).
// Plumbing around the search kernel: integer bookkeeping, one lane per pair.
#include <hip/hip_runtime.h>
#include "s3dp_device.h"

namespace s3 {

__global__ void pairExpandKernel(uint32_t nPairs, const uint32_t *activePairs, uint32_t *readList)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs) return;
    const uint32_t p = activePairs ? activePairs[i] : i;
    readList[2 * i] = 2 * p;
    readList[2 * i + 1] = 2 * p + 1;
}

// Per active pair, after the search of both ends with budget mm (slot 2i = first end, 2i+1 = second end):
//   first end without hit, not the terminal phase            -> whole pair to the next phase       (CPUfunctions.cpp:1985-1996)
//   MAPQ modes, not terminal, first end's best == mm          -> whole pair to the next phase       (:2003-2027)
//   MAPQ modes, an end's best == mm (and mm < 4, not capped)  -> that end is searched again with the (mm+1) model (:2059-2138)
// slot states: 0 nothing to decode, 1 decode this pass, 2 search again
__global__ void pairClassifyKernel(PairClassifyArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nPairs) return;
    const uint32_t p = a.activePairs ? a.activePairs[i] : i;
    const uint32_t t1 = a.totalOcc[2 * i], t2 = a.totalOcc[2 * i + 1];
    uint32_t min1 = 999, min2 = 999;
    for (uint32_t e = 0; e <= a.mm; e++) if (a.withError[(size_t)(2 * i) * 8 + e] > 0) { min1 = e; break; }
    for (uint32_t e = 0; e <= a.mm; e++) if (a.withError[(size_t)(2 * i + 1) * 8 + e] > 0) { min2 = e; break; }
    a.occCount[2 * p] = 0;
    a.occCount[2 * p + 1] = 0;
    bool next = false;
    if (!a.terminal) {
        if (t1 == 0) next = true;
        else if (a.mapq && min1 == a.mm && t1 < a.maxHits) next = true;
    }
    uint8_t s1 = 0, s2 = 0;
    if (!next) {
        const bool h1 = a.mapq && t1 > 0 && min1 == a.mm && t1 < a.maxHits && a.mm < 4;
        const bool h2 = a.mapq && t2 > 0 && min2 == a.mm && t2 < a.maxHits && a.mm < 4;
        s1 = h1 ? 2 : 1;
        s2 = h2 ? 2 : 1;
        a.pairMin[2 * p] = (a.mapq && t1 > 0) ? (uint8_t)min1 : (uint8_t)255;
        a.pairMin[2 * p + 1] = (a.mapq && t2 > 0) ? (uint8_t)min2 : (uint8_t)255;
    }
    a.slotState[2 * i] = s1;
    a.slotState[2 * i + 1] = s2;
    a.pairNext[i] = next ? 1 : 0;
}

// 64-bit sort keys (position, index in the region): unique, so any sort is the stable sort by position that PERadixSort is.
// A pair is sorted only when both ends have hits (PEMappingOccurrences is only called then, CPUfunctions.cpp:2281).
// (a slot's entries by its own lane up to eight, longer lists -- an end inside a repeat family holds thousands -- by the whole wave: one lane walking 8000
//  entries set the time of the launch, 6 ms here and 16 ms in pairSortBackKernel whatever the phase)
__global__ void pairSortKeysKernel(PairSortArgs a)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;      // slot = end of an active pair
    uint32_t beg = 0, cnt = 0;
    if (s < 2 * a.nPairs) {
        const uint32_t i = s >> 1;
        const uint32_t p = a.activePairs ? a.activePairs[i] : i;
        const uint32_t r = 2 * p + (s & 1u);
        const bool on = !a.pairNext[i] && a.occCount[2 * p] > 0 && a.occCount[2 * p + 1] > 0;
        beg = a.occStart[r] - a.regionBase; cnt = on ? a.occCount[r] : 0u;
        a.segBegin[s] = on ? beg : 0u;
        a.segEnd[s] = on ? beg + cnt : 0u;
    }
    // key = position, then the hit's place in its own list (unique keys: the order of the reference's stable sort) -- in 32 + idxBits bits, which is what the
    // radix sort is told to look at (a list holds at most maxHitsPerEnd hits: 13 bits in the command line, where 64-bit keys cost a quarter more passes)
    if (cnt <= 8u) for (uint32_t k = 0; k < cnt; k++) a.keys[beg + k] = (uint64_t)a.occPos[a.regionBase + beg + k] << a.idxBits | k;
    unsigned long long big = __ballot(cnt > 8u);
    while (big) {
        const int l = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t c = (uint32_t)__shfl((int)cnt, l), b0 = (uint32_t)__shfl((int)beg, l);
        for (uint32_t k = lane; k < c; k += 64u) a.keys[b0 + k] = (uint64_t)a.occPos[a.regionBase + b0 + k] << a.idxBits | k;
    }
}

__global__ void pairSortBackKernel(PairSortArgs a, const uint64_t *keysSorted, const uint32_t *metaSorted)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    uint32_t beg = 0, cnt = 0;
    if (s < 2 * a.nPairs) { beg = a.segBegin[s]; cnt = a.segEnd[s] - beg; }
    if (cnt <= 8u)
        for (uint32_t k = beg; k < beg + cnt; k++) {
            a.occPos[a.regionBase + k] = (uint32_t)(keysSorted[k] >> a.idxBits);
            a.occMeta[a.regionBase + k] = metaSorted[k];
        }
    unsigned long long big = __ballot(cnt > 8u);
    while (big) {
        const int l = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t c = (uint32_t)__shfl((int)cnt, l), b0 = (uint32_t)__shfl((int)beg, l);
        for (uint32_t k = b0 + lane; k < b0 + c; k += 64u) {
            a.occPos[a.regionBase + k] = (uint32_t)(keysSorted[k] >> a.idxBits);
            a.occMeta[a.regionBase + k] = metaSorted[k];
        }
    }
}

struct Occ { uint32_t pos, strand, mis; };

// PEMappingCore (PEAlgnmt.cpp:201-291): two-pointer sweep over the position-sorted lists; a '+' occurrence opens a window in the other
// list (PEIsPairEndMatch :566, PEIsPairOutOfRange :599).  `f(o1, o2)` is called for every pair, o1 from list 1, o2 from list 2.
template <class F>
__device__ __forceinline__ void pairSweep(const uint32_t *pos, const uint32_t *meta, uint32_t b1, uint32_t n1, uint32_t b2, uint32_t n2, uint32_t patLen,
                                          uint32_t lb, uint32_t ub, F f)
{
    auto get = [&](uint32_t idx) -> Occ { const uint32_t m = meta[idx]; return Occ{pos[idx], m & 0xFFu, (m >> 8) & 0xFFu}; };
    auto match = [&](const Occ &l, const Occ &r) -> bool {
        const uint32_t gap = r.pos + patLen - 1u - l.pos + 1u;
        return lb <= gap && gap <= ub && l.strand == 1u && r.strand == 2u;
    };
    // The reference leaves the inner loop at the first occurrence of the OTHER strand that lies beyond the window (PEIsPairOutOfRange); one of the same
    // strand never ends it, so two lists of thousands of same-strand occurrences (a read pair inside a tandem array) cost their product.  An occurrence
    // beyond the window cannot match whatever its strand, and the lists are sorted by position: leaving at the first one beyond the window reports the
    // same pairs in the same order.
    auto beyond = [&](const Occ &l, const Occ &r) -> bool { return (uint64_t)l.pos + ub < (uint64_t)r.pos + patLen - 1u; };
    uint32_t i1 = 0, i2 = 0;
    while (i1 < n1 && i2 < n2) {
        const Occ o1 = get(b1 + i1), o2 = get(b2 + i2);
        if (o1.pos <= o2.pos) {
            if (o1.strand == 1u)
                for (uint32_t k = i2; k < n2; k++) {
                    const Occ r = get(b2 + k);
                    if (match(o1, r)) f(o1, r);
                    if (beyond(o1, r)) break;
                }
            i1++;
        } else {
            if (o2.strand == 1u)
                for (uint32_t k = i1; k < n1; k++) {
                    const Occ r = get(b1 + k);
                    if (match(o2, r)) f(r, o2);
                    if (beyond(o2, r)) break;
                }
            i2++;
        }
    }
}

// pass 1: count the pairs, pick the optimal one (fewest mismatches, then smallest |m1 - m2|, first wins) and the "suboptimal" one
// (the last demoted optimal), PEStatsPEPairList (PEAlgnmt.cpp:777-832).  No pair: not terminal -> next phase; terminal -> the two
// occurrence lists stay published for the improper-pair records.
__global__ void pairStatsKernel(PairSweepArgs a)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nPairs) return;
    a.xaCount[i] = 0;
    if (a.pairNext[i]) return;
    const uint32_t p = a.activePairs ? a.activePairs[i] : i;
    const uint32_t n1 = a.occCount[2 * p], n2 = a.occCount[2 * p + 1];
    DevPair &P = a.pairs[p];
    P.min1 = a.pairMin[2 * p]; P.min2 = a.pairMin[2 * p + 1];
    P.x0_1 = P.x1_1 = P.x0_2 = P.x1_2 = 0;
    if (P.min1 != 255) { P.x0_1 = a.finalWithError[(size_t)(2 * p) * 8 + P.min1]; P.x1_1 = a.finalWithError[(size_t)(2 * p) * 8 + P.min1 + 1]; }
    if (P.min2 != 255) { P.x0_2 = a.finalWithError[(size_t)(2 * p + 1) * 8 + P.min2]; P.x1_2 = a.finalWithError[(size_t)(2 * p + 1) * 8 + P.min2 + 1]; }
    if (n1 > 0 && n2 > 0 && n1 + n2 > a.heavyMin) {                 // a whole wave takes this one (pairStatsHeavyKernel)
        a.heavyList[atomicAdd(a.heavyCount, 1u)] = i;
        return;
    }
    uint32_t count = 0, bestIdx = 0, oc = 255, od = 255, subTot = 127;
    uint32_t stats[16];
    for (int k = 0; k < 16; k++) stats[k] = 0;
    Occ b1{0, 0, 0}, b2{0, 0, 0};
    if (n1 > 0 && n2 > 0)
        pairSweep(a.occPos, a.occMeta, a.occStart[2 * p], n1, a.occStart[2 * p + 1], n2, a.readLengths[2 * p + 1], a.insertLow, a.insertHigh,
                  [&](const Occ &o1, const Occ &o2) {
                      const uint32_t tot = o1.mis + o2.mis, d = o1.mis > o2.mis ? o1.mis - o2.mis : o2.mis - o1.mis;
                      stats[tot & 15u]++;
                      if (tot < oc) { subTot = oc == 255 ? 127u : oc; oc = tot; od = d; bestIdx = count; b1 = o1; b2 = o2; }
                      else if (tot == oc && d < od) { od = d; bestIdx = count; b1 = o1; b2 = o2; }
                      count++;
                  });
    if (count == 0) {
        P.phase = 0; P.numPairs = 0; P.numMin = 0; P.minTotal = 127; P.secMinTotal = 127;
        if (!a.terminal) a.pairNext[i] = 1;
        return;
    }
    P.phase = (uint8_t)(a.mm + 1);
    P.numPairs = count; P.numMin = stats[oc & 15u]; P.minTotal = (uint8_t)oc; P.secMinTotal = (uint8_t)subTot;
    P.bestA1 = b1.pos; P.bestS1 = (uint8_t)b1.strand; P.bestM1 = (uint8_t)b1.mis;
    P.bestA2 = b2.pos; P.bestS2 = (uint8_t)b2.strand; P.bestM2 = (uint8_t)b2.mis;
    a.bestIdx[i] = bestIdx;
    // XA:Z holds the other pairs with the optimal mismatch sum, at most peMaxOutputPerPair records in total (BGS-IO.cpp:3577-3612)
    uint32_t xc = stats[oc & 15u] - 1u;
    if (a.peMaxOutputPerPair == 0) xc = 0;
    else if (xc > a.peMaxOutputPerPair - 1u) xc = a.peMaxOutputPerPair - 1u;
    a.xaCount[i] = xc;
}

// pass 2: the same sweep again, writing the XA pairs in pairing order
__global__ void pairEmitKernel(PairSweepArgs a, const uint32_t *xaOffsets, uint32_t xaBase, DevPairXa *xa)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nPairs) return;
    const uint32_t want = a.xaCount[i];
    const uint32_t p = a.activePairs ? a.activePairs[i] : i;
    if (a.pairs[p].phase != (uint8_t)(a.mm + 1) || a.pairNext[i]) return;
    a.xaStart[p] = xaBase + xaOffsets[i];
    a.xaCountOut[p] = want;
    a.occCount[2 * p] = 0;                                        // a properly paired read publishes no occurrence list
    a.occCount[2 * p + 1] = 0;
    if (!want) return;
    const uint32_t n1 = a.occCountIn[2 * i], n2 = a.occCountIn[2 * i + 1];
    if (n1 + n2 > a.heavyMin) return;                                // pairEmitHeavyKernel writes this pair's list
    const uint32_t oc = a.pairs[p].minTotal, best = a.bestIdx[i];
    uint32_t idx = 0, w = 0;
    DevPairXa *dst = xa + xaBase + xaOffsets[i];
    pairSweep(a.occPos, a.occMeta, a.occStart[2 * p], n1, a.occStart[2 * p + 1], n2, a.readLengths[2 * p + 1], a.insertLow, a.insertHigh,
              [&](const Occ &o1, const Occ &o2) {
                  if (idx != best && o1.mis + o2.mis <= oc && w < want) {
                      DevPairXa x;
                      x.a1 = o1.pos; x.a2 = o2.pos; x.s1 = (uint8_t)o1.strand; x.s2 = (uint8_t)o2.strand; x.m1 = (uint8_t)o1.mis; x.m2 = (uint8_t)o2.mis;
                      dst[w++] = x;
                  }
                  idx++;
              });
}

// ---- pairs with long lists: one wave per pair --------------------------------------------------------------------------------------------------------
// The sweep above is a merge: serial, two dependent loads per step.  A pair inside a repeat family brings up to MaxHitsEachEndForPairing hits per end, and the
// lane that walked the longest pair set the time of the launch (16 ms for pairStatsKernel, 10 ms for pairEmitKernel, whatever the phase).  What the sweep
// reports can be said without the merge: an occurrence x takes the part of the LEFT leg when the sweep reaches it -- in the merged order of both lists,
// list 1 first on equal positions -- and pairs with the entries of the other list that the sweep has not passed yet (positions >= x's; > for an x of
// list 2) and that lie in the insert window.  So: rank(x) = its index + the other list's entries before it (one binary search), its partners = a
// contiguous stretch of the other list (a second search when the window starts beyond x), and the order of the reported pairs = (rank(x), partner index).
// A wave deals the occurrences of both lists to its lanes; the optimal pair is the minimum of the key (mismatch sum, difference, rank, partner number) --
// "first wins" is part of the key --, the statistics are sums, the index of the optimal pair in reporting order a sum over the ranks below it, and the
// "suboptimal" sum (the running minimum before the LAST improvement, PEStatsPEPairList) the minimum over everything reported before the first pair with the
// final minimum.  The XA list (pass 2) is an exclusive scan of per-occurrence counts in rank order.  Same pairs, same order, same numbers as pairSweep.
struct HeavyPair { uint32_t b1, n1, b2, n2, patLen, lb, ub; const uint32_t *pos, *meta; };

__device__ __forceinline__ uint32_t lowerBoundPos(const uint32_t *pos, uint32_t b, uint32_t n, uint64_t v)   // first k in [0, n) with pos[b + k] >= v
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)pos[b + mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// occurrence g (g < n1: of list 1) when the sweep takes it: its rank in the merged order, the first entry of the other list not passed yet
__device__ __forceinline__ uint32_t heavyRank(const HeavyPair &h, uint32_t g, uint32_t &k0)
{
    const bool fromA = g < h.n1;
    const uint32_t own = fromA ? h.b1 + g : h.b2 + (g - h.n1);
    k0 = lowerBoundPos(h.pos, fromA ? h.b2 : h.b1, fromA ? h.n2 : h.n1, (uint64_t)h.pos[own] + (fromA ? 0u : 1u));
    return (fromA ? g : g - h.n1) + k0;
}

// its partners in the sweep's order: f(o1 of list 1, o2 of list 2, number of the partner); returns how many
template <class F>
__device__ __forceinline__ uint32_t heavyScan(const HeavyPair &h, uint32_t g, uint32_t k0, F f)
{
    const bool fromA = g < h.n1;
    const uint32_t own = fromA ? h.b1 + g : h.b2 + (g - h.n1);
    const uint32_t ob = fromA ? h.b2 : h.b1, on = fromA ? h.n2 : h.n1;
    const uint32_t xm = h.meta[own];
    if ((xm & 0xFFu) != 1u) return 0;
    const Occ x{h.pos[own], 1u, (xm >> 8) & 0xFFu};
    uint32_t k = k0;
    // entries in front of the window neither match nor end the sweep's inner loop: skipped with a search
    if (h.lb > h.patLen) { const uint32_t kw = lowerBoundPos(h.pos, ob, on, (uint64_t)x.pos + (h.lb - h.patLen)); if (kw > k) k = kw; }
    uint32_t c = 0;
    for (; k < on; k++) {
        const uint32_t m = h.meta[ob + k];
        const Occ r{h.pos[ob + k], m & 0xFFu, (m >> 8) & 0xFFu};
        const uint32_t gap = r.pos + h.patLen - 1u - x.pos + 1u;
        if (h.lb <= gap && gap <= h.ub && r.strand == 2u) { if (fromA) f(x, r, c); else f(r, x, c); c++; }
        if ((uint64_t)x.pos + h.ub < (uint64_t)r.pos + h.patLen - 1u) break;
    }
    return c;
}

__device__ __forceinline__ uint32_t waveSumU(uint32_t v) { for (int o = 32; o; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o); return v; }
__device__ __forceinline__ uint32_t waveMinU(uint32_t v) { for (int o = 32; o; o >>= 1) { const uint32_t w = (uint32_t)__shfl_xor((int)v, o); v = w < v ? w : v; } return v; }
__device__ __forceinline__ uint64_t waveMinU64(uint64_t v)
{
    for (int o = 32; o; o >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o);
        const uint64_t w = (uint64_t)hi << 32 | lo;
        v = w < v ? w : v;
    }
    return v;
}

__global__ void __launch_bounds__(64) pairStatsHeavyKernel(PairSweepArgs a)
{
    __shared__ uint32_t hist[16 * 64];
    const uint32_t lane = threadIdx.x;
    const uint32_t nH = *a.heavyCount;
    for (uint32_t hI = blockIdx.x; hI < nH; hI += gridDim.x) {
        const uint32_t i = a.heavyList[hI];
        const uint32_t p = a.activePairs ? a.activePairs[i] : i;
        const HeavyPair h{a.occStart[2 * p], a.occCount[2 * p], a.occStart[2 * p + 1], a.occCount[2 * p + 1], a.readLengths[2 * p + 1], a.insertLow, a.insertHigh,
                          a.occPos, a.occMeta};
        const uint32_t N = h.n1 + h.n2;
        for (int t = 0; t < 16; t++) hist[t * 64 + lane] = 0;
        uint32_t cnt = 0;
        uint64_t bestKey = ~0ull, firstKey = ~0ull;
        Occ b1{0, 0, 0}, b2{0, 0, 0};
        for (uint32_t g = lane; g < N; g += 64u) {
            uint32_t k0, cur = 255, prev = 255;
            const uint32_t rank = heavyRank(h, g, k0);
            const uint32_t c = heavyScan(h, g, k0, [&](const Occ &o1, const Occ &o2, uint32_t ci) {
                const uint32_t tot = o1.mis + o2.mis, d = o1.mis > o2.mis ? o1.mis - o2.mis : o2.mis - o1.mis;
                hist[(tot & 15u) * 64 + lane]++;
                if (tot < cur) { prev = cur; cur = tot; }
                const uint64_t key = (uint64_t)(tot & 15u) << 60 | (uint64_t)(d & 15u) << 56 | (uint64_t)rank << 28 | ci;
                if (key < bestKey) { bestKey = key; b1 = o1; b2 = o2; }
            });
            const uint32_t slot = (g < h.n1 ? h.b1 + g : h.b2 + (g - h.n1)) - a.regionBase;
            a.scratchW[slot] = c | cur << 24;
            a.scratchR[slot] = rank;
            cnt += c;
            if (c) { const uint64_t k2 = (uint64_t)cur << 40 | (uint64_t)rank << 8 | prev; firstKey = k2 < firstKey ? k2 : firstKey; }
        }
        const uint32_t count = waveSumU(cnt);
        DevPair &P = a.pairs[p];
        if (count == 0) {
            if (lane == 0) {
                P.phase = 0; P.numPairs = 0; P.numMin = 0; P.minTotal = 127; P.secMinTotal = 127;
                if (!a.terminal) a.pairNext[i] = 1;
            }
            continue;
        }
        const uint64_t best = waveMinU64(bestKey), first = waveMinU64(firstKey);
        const uint32_t oc = (uint32_t)(best >> 60), rankStar = (uint32_t)(best >> 28) & 0xFFFFFFFu, cStar = (uint32_t)best & 0xFFFFFFFu;
        const uint32_t rankF = (uint32_t)(first >> 8), prevF = (uint32_t)first & 0xFFu;
        const uint32_t numMin = waveSumU(hist[(oc & 15u) * 64 + lane]);
        uint32_t pre = 0, sub = 255;
        for (uint32_t g = lane; g < N; g += 64u) {                       // (every lane reads back what it wrote itself)
            const uint32_t slot = (g < h.n1 ? h.b1 + g : h.b2 + (g - h.n1)) - a.regionBase;
            const uint32_t w = a.scratchW[slot], rk = a.scratchR[slot];
            if (rk < rankStar) pre += w & 0xFFFFFFu;
            if (rk < rankF && (w & 0xFFFFFFu)) sub = min(sub, w >> 24);
        }
        pre = waveSumU(pre);
        sub = min(waveMinU(sub), prevF);
        if (bestKey == best) {                                          // (the key holds rank and partner number: one lane)
            P.bestA1 = b1.pos; P.bestS1 = (uint8_t)b1.strand; P.bestM1 = (uint8_t)b1.mis;
            P.bestA2 = b2.pos; P.bestS2 = (uint8_t)b2.strand; P.bestM2 = (uint8_t)b2.mis;
        }
        if (lane == 0) {
            P.phase = (uint8_t)(a.mm + 1);
            P.numPairs = count; P.numMin = numMin; P.minTotal = (uint8_t)oc; P.secMinTotal = (uint8_t)(sub == 255u ? 127u : sub);
            a.bestIdx[i] = pre + cStar;                                  // (what the one-lane kernels call the best index; the XA pass of this pair uses rank + number)
            a.bestRank[i] = rankStar; a.bestLocal[i] = cStar;
            uint32_t xc = numMin - 1u;
            if (a.peMaxOutputPerPair == 0) xc = 0;
            else if (xc > a.peMaxOutputPerPair - 1u) xc = a.peMaxOutputPerPair - 1u;
            a.xaCount[i] = xc;
        }
    }
}

__global__ void __launch_bounds__(64) pairEmitHeavyKernel(PairSweepArgs a, const uint32_t *xaOffsets, uint32_t xaBase, DevPairXa *xa)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t nH = *a.heavyCount;
    for (uint32_t hI = blockIdx.x; hI < nH; hI += gridDim.x) {
        const uint32_t i = a.heavyList[hI];
        const uint32_t want = a.xaCount[i];
        const uint32_t p = a.activePairs ? a.activePairs[i] : i;
        if (!want || a.pairs[p].phase != (uint8_t)(a.mm + 1) || a.pairNext[i]) continue;
        const HeavyPair h{a.occStart[2 * p], a.occCountIn[2 * i], a.occStart[2 * p + 1], a.occCountIn[2 * i + 1], a.readLengths[2 * p + 1], a.insertLow, a.insertHigh,
                          a.occPos, a.occMeta};
        const uint32_t N = h.n1 + h.n2, oc = a.pairs[p].minTotal, rankStar = a.bestRank[i], cStar = a.bestLocal[i];
        auto slotOfRank = [&](uint32_t rk) -> uint32_t { return (rk < h.n1 ? h.b1 + rk : h.b2 + (rk - h.n1)) - a.regionBase; };
        // how many list entries every occurrence contributes, stored at its RANK
        for (uint32_t g = lane; g < N; g += 64u) {
            uint32_t k0, e = 0;
            const uint32_t rank = heavyRank(h, g, k0);
            heavyScan(h, g, k0, [&](const Occ &o1, const Occ &o2, uint32_t ci) { if (o1.mis + o2.mis <= oc && !(rank == rankStar && ci == cStar)) e++; });
            a.scratchW[slotOfRank(rank)] = e;
        }
        __syncthreads();
        // exclusive scan in rank order
        uint32_t carry = 0;
        for (uint32_t base = 0; base < N; base += 64u) {
            const uint32_t rk = base + lane;
            const uint32_t v = rk < N ? a.scratchW[slotOfRank(rk)] : 0u;
            uint32_t inc = v;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o); if ((int)lane >= o) inc += t; }
            if (rk < N) a.scratchW[slotOfRank(rk)] = carry + inc - v;
            carry += (uint32_t)__shfl((int)inc, 63);
        }
        __syncthreads();
        DevPairXa *dst = xa + xaBase + xaOffsets[i];
        for (uint32_t g = lane; g < N; g += 64u) {
            uint32_t k0;
            const uint32_t rank = heavyRank(h, g, k0);
            uint32_t w = a.scratchW[slotOfRank(rank)];
            if (w >= want) continue;
            heavyScan(h, g, k0, [&](const Occ &o1, const Occ &o2, uint32_t ci) {
                if (o1.mis + o2.mis <= oc && !(rank == rankStar && ci == cStar)) {
                    if (w < want) {
                        DevPairXa x;
                        x.a1 = o1.pos; x.a2 = o2.pos; x.s1 = (uint8_t)o1.strand; x.s2 = (uint8_t)o2.strand; x.m1 = (uint8_t)o1.mis; x.m2 = (uint8_t)o2.mis;
                        dst[w] = x;
                    }
                    w++;
                }
            });
        }
        __syncthreads();                                                 // (the next pair of this wave reuses nothing of this one's scratch: its slots are its own)
    }
}

__global__ void pairSnapshotCountsKernel(uint32_t nPairs, const uint32_t *activePairs, const uint32_t *occCount, uint32_t *occCountIn)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs) return;
    const uint32_t p = activePairs ? activePairs[i] : i;
    occCountIn[2 * i] = occCount[2 * p];
    occCountIn[2 * i + 1] = occCount[2 * p + 1];
}

__global__ void pairGatherXaKernel(uint32_t nPairs, const uint32_t *xaStart, const uint32_t *outOffsets, const DevPairXa *xa, DevPairXa *out)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nPairs) return;
    const uint32_t cnt = outOffsets[p + 1] - outOffsets[p];
    for (uint32_t k = 0; k < cnt; k++) out[outOffsets[p] + k] = xa[xaStart[p] + k];
}

void launchPairExpand(uint32_t nPairs, const uint32_t *activePairs, uint32_t *readList, hipStream_t s)
{
    if (nPairs) hipLaunchKernelGGL(pairExpandKernel, dim3((nPairs + 255) / 256), dim3(256), 0, s, nPairs, activePairs, readList);
}
void launchPairClassify(const PairClassifyArgs &a, hipStream_t s)
{
    if (a.nPairs) hipLaunchKernelGGL(pairClassifyKernel, dim3((a.nPairs + 255) / 256), dim3(256), 0, s, a);
}
void launchPairSortKeys(const PairSortArgs &a, hipStream_t s)
{
    if (a.nPairs) hipLaunchKernelGGL(pairSortKeysKernel, dim3((2 * a.nPairs + 255) / 256), dim3(256), 0, s, a);
}
void launchPairSortBack(const PairSortArgs &a, const uint64_t *keysSorted, const uint32_t *metaSorted, hipStream_t s)
{
    if (a.nPairs) hipLaunchKernelGGL(pairSortBackKernel, dim3((2 * a.nPairs + 255) / 256), dim3(256), 0, s, a, keysSorted, metaSorted);
}
void launchPairSnapshotCounts(uint32_t nPairs, const uint32_t *activePairs, const uint32_t *occCount, uint32_t *occCountIn, hipStream_t s)
{
    if (nPairs) hipLaunchKernelGGL(pairSnapshotCountsKernel, dim3((nPairs + 255) / 256), dim3(256), 0, s, nPairs, activePairs, occCount, occCountIn);
}
// (the caller zeroes *a.heavyCount before launchPairStats; the heavy kernels run a fixed grid of waves over the list the light kernel made)
void launchPairStats(const PairSweepArgs &a, hipStream_t s)
{
    if (!a.nPairs) return;
    hipLaunchKernelGGL(pairStatsKernel, dim3((a.nPairs + 63) / 64), dim3(64), 0, s, a);
    hipLaunchKernelGGL(pairStatsHeavyKernel, dim3(4096), dim3(64), 0, s, a);
}
void launchPairEmit(const PairSweepArgs &a, const uint32_t *xaOffsets, uint32_t xaBase, DevPairXa *xa, hipStream_t s)
{
    if (!a.nPairs) return;
    hipLaunchKernelGGL(pairEmitKernel, dim3((a.nPairs + 63) / 64), dim3(64), 0, s, a, xaOffsets, xaBase, xa);
    hipLaunchKernelGGL(pairEmitHeavyKernel, dim3(4096), dim3(64), 0, s, a, xaOffsets, xaBase, xa);
}
void launchPairGatherXa(uint32_t nPairs, const uint32_t *xaStart, const uint32_t *outOffsets, const DevPairXa *xa, DevPairXa *out, hipStream_t s)
{
    if (nPairs) hipLaunchKernelGGL(pairGatherXaKernel, dim3((nPairs + 255) / 256), dim3(256), 0, s, nPairs, xaStart, outOffsets, xa, out);
}

}  // namespace s3

// one kernel of this file's code object, asked for its attributes: the runtime loads a code object at the first use of one of its kernels (preloadKernels, s3dp_kernels.hip)
void s3::preloadPair() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)s3::pairExpandKernel); (void)hipGetLastError(); }
