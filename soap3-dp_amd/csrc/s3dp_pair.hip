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
__global__ void pairSortKeysKernel(PairSortArgs a)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;      // slot = end of an active pair
    if (s >= 2 * a.nPairs) return;
    const uint32_t i = s >> 1;
    const uint32_t p = a.activePairs ? a.activePairs[i] : i;
    const uint32_t r = 2 * p + (s & 1u);
    const bool on = !a.pairNext[i] && a.occCount[2 * p] > 0 && a.occCount[2 * p + 1] > 0;
    const uint32_t beg = a.occStart[r] - a.regionBase, cnt = on ? a.occCount[r] : 0u;
    a.segBegin[s] = on ? beg : 0u;
    a.segEnd[s] = on ? beg + cnt : 0u;
    for (uint32_t k = 0; k < cnt; k++) a.keys[beg + k] = (uint64_t)a.occPos[a.regionBase + beg + k] << 32 | (beg + k);
}

__global__ void pairSortBackKernel(PairSortArgs a, const uint64_t *keysSorted, const uint32_t *metaSorted)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= 2 * a.nPairs) return;
    for (uint32_t k = a.segBegin[s]; k < a.segEnd[s]; k++) {
        a.occPos[a.regionBase + k] = (uint32_t)(keysSorted[k] >> 32);
        a.occMeta[a.regionBase + k] = metaSorted[k];
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
void launchPairStats(const PairSweepArgs &a, hipStream_t s)
{
    if (a.nPairs) hipLaunchKernelGGL(pairStatsKernel, dim3((a.nPairs + 63) / 64), dim3(64), 0, s, a);
}
void launchPairEmit(const PairSweepArgs &a, const uint32_t *xaOffsets, uint32_t xaBase, DevPairXa *xa, hipStream_t s)
{
    if (a.nPairs) hipLaunchKernelGGL(pairEmitKernel, dim3((a.nPairs + 63) / 64), dim3(64), 0, s, a, xaOffsets, xaBase, xa);
}
void launchPairGatherXa(uint32_t nPairs, const uint32_t *xaStart, const uint32_t *outOffsets, const DevPairXa *xa, DevPairXa *out, hipStream_t s)
{
    if (nPairs) hipLaunchKernelGGL(pairGatherXaKernel, dim3((nPairs + 255) / 256), dim3(256), 0, s, nPairs, xaStart, outOffsets, xa, out);
}

}  // namespace s3
