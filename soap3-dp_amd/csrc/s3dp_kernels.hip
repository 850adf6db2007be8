// s3dp_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X) around the 2BWT search: index re-layout, read orientation, long lookup
// tables, record ordering, per-read merge, SA decode.  The search itself is s3dp_first.hip / s3dp_unit.hip / s3dp_ce.hip.
//
// What is computed is fixed by the reference's CPU path (BGS-HostAlgnmtAlgo2.cpp driven by the
// SRA2BWTMdl.c case tables): same cases, same DFS order, same lookup-table start, same
// check-and-extend rule, so the emitted records are the CPU path's, in its order.  How it is
// computed is MI355X-first:
//   * index re-laid out at upload: ONE 64-byte line per rank query
//       [ Occ(.,128k)+C : 16 B | BWT chars 128k-64..128k-1 : 16 B | BWT chars 128k..128k+63 : 16 B | pad ]
//     (the file layout needs two uncorrelated 16-B gathers; DV-Kernel.cu:256-299)
//   * one work item = (read, case, strand); items are ordered so that a 64-lane wavefront shares
//     case and strand -> the step table walk is wave-uniform, divergence is data-only
//   * oriented reads live in LDS, word-interleaved by lane (bank = lane, conflict free)
//   * suffix array, packed text and both 13-mer lookup tables are HBM-resident, so check-and-extend
//     and SA decode happen on the device (no CPU fallback / second round as in alignment.cu:221-326)
//   * records are appended with wave-aggregated atomics and re-ordered by a counting scatter
//     (exclusive scan of per-item counts), no sort, no fixed answer slots.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "s3dp_model.h"
#include <algorithm>
#include "s3dp_device.h"

namespace s3 {

// ---------------------------------------------------------------------------------------------
// rank
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t hiMask(uint32_t nChars)   // first nChars of a 32-char MSB-first word
{
    return nChars >= 32 ? ~0ULL : (nChars == 0 ? 0ULL : (~0ULL << (64 - 2 * nChars)));
}
__device__ __forceinline__ uint64_t loMask(uint32_t nChars)   // last nChars
{
    return nChars >= 32 ? ~0ULL : ((1ULL << (2 * nChars)) - 1ULL);
}

// v[c] = Occ(c, i) + C[c]   (GPUBWTAllOccValue, DV-Kernel.cu:282-299; BWTAllOccValue, BWT.c:1211)
// Everything stays in registers: no dynamically indexed arrays (they would live in scratch memory).
__device__ __forceinline__ uint4 rankAll(const DevBwt &b, uint32_t i)
{
    i -= (i > b.inverseSa0);
    const uint32_t k = (i + 63u) >> 7, p = k << 7;
    const uint4 *line = b.lines + (size_t)k * 4;
    uint4 v = line[0];
    if (i != p) {
        const bool fwd = i > p;
        const uint32_t d = fwd ? i - p : p - i;
        // the d characters next to the sample are the FIRST d of either half-block: the half before the sample is stored reversed
        const uint4 w = line[fwd ? 2 : 1];
        const uint64_t wA = ((uint64_t)w.x << 32) | w.y, wB = ((uint64_t)w.z << 32) | w.w;
        const uint64_t mA = hiMask(d);
        const uint64_t mB = d > 32 ? hiMask(d - 32) : 0ULL;
        const uint64_t M5 = 0x5555555555555555ULL;
        const uint64_t loA = wA & M5, hiA = (wA >> 1) & M5, loB = wB & M5, hiB = (wB >> 1) & M5;
        const uint64_t a = mA & M5, bm = mB & M5;
        const uint32_t c3 = __popcll(hiA & loA & a) + __popcll(hiB & loB & bm);
        const uint32_t c2 = __popcll(hiA & ~loA & a) + __popcll(hiB & ~loB & bm);
        const uint32_t c1 = __popcll(~hiA & loA & a) + __popcll(~hiB & loB & bm);
        const uint32_t c0 = d - c1 - c2 - c3;
        if (fwd) { v.x += c0; v.y += c1; v.z += c2; v.w += c3; }
        else { v.x -= c0; v.y -= c1; v.z -= c2; v.w -= c3; }
    }
    return v;
}

__device__ __forceinline__ uint32_t sel4(const uint4 &v, uint32_t c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

// Sampled suffix array -> full array (BWTSaValue, 2bwt-lib/BWT.c:1694-1724): from index i walk backwards through the BWT (BWTPsiMinusValue :1641:
// the row of the previous text position, 0 from the row of the text's first character) until an index divisible by the sampling interval, and add
// the steps to the value stored there.  SA[0] is stored as -1 (BWT.c:286), so the sums wrap the way the reference's do.
__global__ void saExpandKernel(DevBwt b, const uint32_t *samples, uint32_t interval, uint32_t n, uint32_t *sa)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint32_t j = (uint32_t)i, steps = 0;
    while (j % interval != 0u) {
        if (j == b.inverseSa0) j = 0;
        else {
            // the character at BWT position j is the one whose count grows between j and j + 1; its new count + C[c] is the row
            const uint4 lo = rankAll(b, j), hi = rankAll(b, j + 1u);
            j = hi.x != lo.x ? hi.x : hi.y != lo.y ? hi.y : hi.z != lo.z ? hi.z : hi.w;
        }
        steps++;
    }
    sa[i] = samples[j / interval] + steps;
}

void launchSaExpand(const DevBwt &b, const uint32_t *samples, uint32_t interval, uint32_t n, uint32_t *sa, hipStream_t s)
{
    hipLaunchKernelGGL(saExpandKernel, dim3((unsigned)(((uint64_t)n + 1 + 255) / 256)), dim3(256), 0, s, b, samples, interval, n, sa);
}

// ---------------------------------------------------------------------------------------------
// ordering: records -> CSR by item (counting scatter; offsets = exclusive scan of itemCounts)
// ---------------------------------------------------------------------------------------------
// (the slot gets the record's NUMBER; a record of the second pass is given its item back in place)
__global__ void scatterKernel(uint4 *recs, const uint32_t *recCount, uint32_t recCap, const uint32_t *itemOffsets, uint32_t *ordIdx, uint8_t *valid,
                              const uint8_t *heavyFlags, const uint32_t *heavyList, uint32_t groups, uint32_t unitBase, uint32_t unitsPerItem, ScatterDyn dyn)
{
    uint32_t n = *recCount;
    if (n > recCap) n = recCap;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint4 r = recs[i];
        if (r.x == 0xFFFFFFFFu || (r.y & S3_REC_TASK)) continue;            // unused slot of a reserved block / check-and-extend task (v2 kernel)
        uint32_t at = r.x;
        if (r.x & S3_REC_PASS2) {
            // written by the second search pass (or by check-and-extend of its tasks): counted per unit; downstream the record carries its ITEM again
            at = r.x & ~S3_REC_PASS2;
            uint32_t inChain = 0;
            if (dyn.cap && at >= dyn.base) {
                // counted by an adopted frame (s3dp_unit.hip): its records sit inside the slice of the unit its chain started from
                inChain = dyn.local[at - dyn.base];
                at = dyn.owner[at - dyn.base];
            }
            const uint32_t unit = at - unitBase;
            recs[i].x = heavyList[unit / (groups * unitsPerItem)] * groups + (unit / unitsPerItem) % groups;
            const uint32_t dst = itemOffsets[at] + inChain + (r.y & 0xFFFFFFu);
            ordIdx[dst] = i; valid[dst] = 1;
            continue;
        } else if (heavyFlags[r.x / groups]) continue;                         // stale: the second pass searched this read again
        const uint32_t dst = itemOffsets[at] + (r.y & 0xFFFFFFu);
        ordIdx[dst] = i; valid[dst] = 1;
    }
}

// the read's slice [beg, end) of the ordered array: its items' offsets, or -- for a read the second search pass took apart -- its units' offsets
__device__ __forceinline__ void readSlice(const MergeArgs &a, uint32_t slot, uint32_t groups, uint32_t &beg, uint32_t &end)
{
    if (a.heavyFlags && a.heavyFlags[slot]) {
        const uint32_t u0 = a.unitBase + a.heavyIndex[slot] * a.unitsPerRead;
        beg = a.itemOffsets[u0]; end = a.itemOffsets[u0 + a.unitsPerRead];
    } else { beg = a.itemOffsets[slot * groups]; end = a.itemOffsets[slot * groups + groups]; }
}

// The validity bytes of the eight slots from w0 on (w0 a multiple of 8; the array is 8-byte aligned with 8 bytes to spare: s3dp_api.hip) as a bit mask,
// bit q = slot w0 + q holds a record; slots outside [lo, hi) masked out.  One 8-byte load where a walk slot by slot makes eight 1-byte loads -- and a load
// instruction costs the vector cache a clock per lane whatever its width.
__device__ __forceinline__ uint32_t validBits8(const uint8_t *valid, uint32_t w0, uint32_t lo, uint32_t hi)
{
    const uint2 v = *(const uint2 *)(valid + w0);
    auto nz = [](uint32_t x) -> uint32_t { x |= x >> 4; x |= x >> 2; x |= x >> 1; x &= 0x01010101u; return ((x * 0x01020408u) >> 24) & 0xFu; };   // bit q: byte q is not zero
    uint32_t bits = nz(v.x) | nz(v.y) << 4;
    if (w0 < lo) bits = lo - w0 >= 8u ? 0u : bits & (0xFFu << (lo - w0));
    if (w0 >= hi) bits = 0u;
    else if (hi - w0 < 8u) bits &= (1u << (hi - w0)) - 1u;
    return bits;
}

// Per read: walk the items in the CPU path's order (case 0 '+', case 0 '-', case 1 '+', ...), apply the
// MaxOutputPerRead truncation across items (TotalOccurrences is per read: CPUfunctions.cpp:1246, the cut is
// BGS-HostAlgnmtAlgo2.cpp:1612-1615 / :160-163), tally WithError[], and count what survives.
__global__ void mergeCountKernel(MergeArgs a)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= a.nActive) return;
    const uint32_t groups = a.nCases * 2u;
    uint32_t beg, end;
    readSlice(a, slot, groups, beg, end);
    if (a.longList && end - beg > S3_MERGE_LONG) { a.longList[atomicAdd(a.longCount, 1u)] = slot; return; }   // a wave's job (mergeCountLongKernel)
    uint32_t total = 0, nRec = 0, nOcc = 0, nCe = 0;
    uint32_t we[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t w0 = beg & ~7u; w0 < end && total < a.maxOut; w0 += 8u) {
        uint32_t bits = validBits8(a.valid, w0, beg, end);                   // (a slot without a record: the sequence number of a rejected check-and-extend candidate)
        while (bits && total < a.maxOut) {
            const uint32_t i = w0 + (uint32_t)__ffs((int)bits) - 1u;
            bits &= bits - 1u;
            const uint4 r = a.recs[a.ordIdx[i]];
            const uint32_t kind = (r.y >> 24) & 1u, nmis = (r.y >> 25) & 7u;
            uint32_t cnt = kind ? 1u : r.w - r.z + 1u;
            if (total + cnt > a.maxOut) cnt = a.maxOut - total;
            total += cnt; we[nmis] += cnt; nRec++; nOcc += cnt; nCe += kind;
        }
    }
    a.keptRecs[slot] = nRec;
    a.occCounts[slot] = nOcc;
    a.ceCounts[slot] = nCe;
    a.totalOcc[slot] = total;
    for (int k = 0; k < 8; k++) a.withError[(size_t)slot * 8 + k] = we[k];
}

__device__ __forceinline__ uint32_t waveExclusiveSum(uint32_t v, uint32_t lane, uint32_t &total)
{
    uint32_t inc = v;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, (int)d);
        if (lane >= d) inc += t;
    }
    total = (uint32_t)__shfl((int)inc, 63);
    return inc - v;
}

// The long kernels walk a read's slice 512 slots per trip, and only the slots that hold a record get a lane: nine in ten slots of a read inside a repeat
// family are the sequence numbers of rejected check-and-extend candidates.  A lane reads the validity bytes of EIGHT consecutive slots as one 8-byte word
// (the next trip's are already on their way), the wave numbers the records of the trip through (one prefix sum over the lanes' counts) and writes their slot
// numbers to LDS in order; then 64 RECORDS at a time -- not 64 slots -- are loaded and go through the prefix sums of the cut.  (Round 3 loaded 64 sixteen-byte
// entries per step, holes included; rounds 4-5 walked the trip in eight steps of 64 slots, a prefix-sum round each, for the four or five records a step held.)
#define S3_LONG_TRIP 512u
__device__ __forceinline__ uint2 loadTripValid(const MergeArgs &a, uint32_t tb, uint32_t end, uint32_t lane)
{
    const uint32_t pos = tb + 8u * lane;                                       // (tb is a multiple of 8 and the array 8-byte aligned with 8 bytes to spare: s3dp_api.hip)
    return pos < end ? *(const uint2 *)(a.valid + pos) : make_uint2(0u, 0u);
}
// the trip's records, in slot order: their slot numbers in idx[0 .. V), V returned (wave-uniform)
__device__ __forceinline__ uint32_t tripRecords(uint2 v, uint32_t tb, uint32_t beg, uint32_t end, uint32_t lane, uint32_t *idx)
{
    const uint32_t pos = tb + 8u * lane;
    auto nz = [](uint32_t x) -> uint32_t { x |= x >> 4; x |= x >> 2; x |= x >> 1; x &= 0x01010101u; return ((x * 0x01020408u) >> 24) & 0xFu; };   // bit q: byte q is not zero
    uint32_t bits = nz(v.x) | nz(v.y) << 4;
    if (pos < beg) bits = beg - pos >= 8u ? 0u : bits & (0xFFu << (beg - pos));
    if (pos >= end) bits = 0u;
    else if (end - pos < 8u) bits &= (1u << (end - pos)) - 1u;
    uint32_t V = 0;
    uint32_t k = waveExclusiveSum((uint32_t)__popc(bits), lane, V);
    while (bits) {
        const uint32_t q = (uint32_t)__ffs((int)bits) - 1u;
        bits &= bits - 1u;
        idx[k++] = pos + q;
    }
    __builtin_amdgcn_wave_barrier();
    return V;
}

// (The listed reads are dealt to the waves of the long kernels in a fixed order, read waveId + k * nWaves.  Dealing them through a counter as the waves come
//  free was built twice: round 4 saw the ladder hang and took it out unexplained; round 5's form -- a counter word per kernel, zeroed right before the
//  launch, drawn unconditionally by the whole wave -- ran green and 11 ms per ladder SLOWER (8,192 waves on one word), and was taken out in round 6.)
// mergeCountKernel for the listed (long) reads: one wave per read -- the holes of rejected check-and-extend
// candidates (nine in ten entries of a read inside a repeat family) cost a byte of a coalesced load instead of a turn of a serial loop
__global__ void __launch_bounds__(256) mergeCountLongKernel(MergeArgs a)
{
    __shared__ uint32_t sIdx[4][S3_LONG_TRIP];
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *idx = sIdx[threadIdx.x >> 6];
    const uint32_t waveId = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t nLong = *a.longCount, groups = a.nCases * 2u;
    for (uint32_t li = waveId; li < nLong; li += nWaves) {
        const uint32_t slot = a.longList[li];
        uint32_t beg, end;
        readSlice(a, slot, groups, beg, end);
        uint32_t total = 0, nRec = 0, nOcc = 0, nCe = 0;                      // nRec / nOcc / nCe / we: per-lane partial sums
        uint32_t we[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint32_t tb0 = beg & ~7u;
        uint2 next = loadTripValid(a, tb0, end, lane);
        for (uint32_t tb = tb0; tb < end && total < a.maxOut; tb += S3_LONG_TRIP) {
            const uint2 cur = next;
            if (tb + S3_LONG_TRIP < end) next = loadTripValid(a, tb + S3_LONG_TRIP, end, lane);
            const uint32_t V = tripRecords(cur, tb, beg, end, lane, idx);
            for (uint32_t r0 = 0; r0 < V && total < a.maxOut; r0 += 64u) {
                const bool valid = r0 + lane < V;
                const uint4 r = a.recs[valid ? a.ordIdx[idx[r0 + lane]] : 0u];            // (a lane without a record reads record 0: a hole's slot holds no number)
                const uint32_t kind = (r.y >> 24) & 1u, nmis = (r.y >> 25) & 7u;
                // a range is never longer than maxOut can use: clamp before the prefix sum (32-bit)
                uint32_t cnt = valid ? (kind ? 1u : r.w - r.z + 1u) : 0u;
                if (cnt > a.maxOut) cnt = a.maxOut;
                uint32_t sum = 0;
                const uint32_t pre = waveExclusiveSum(cnt, lane, sum);
                const bool incl = valid && total + pre < a.maxOut;
                uint32_t c = 0;
                if (incl) { c = cnt; if (total + pre + c > a.maxOut) c = a.maxOut - total - pre; }
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) we[k] += nmis == k ? c : 0u;
                nRec += incl ? 1u : 0u; nOcc += c; nCe += incl ? kind : 0u;
                total = total + sum > a.maxOut ? a.maxOut : total + sum;
            }
            __builtin_amdgcn_wave_barrier();
        }
        (void)waveExclusiveSum(nRec, lane, nRec); (void)waveExclusiveSum(nOcc, lane, nOcc); (void)waveExclusiveSum(nCe, lane, nCe);
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) (void)waveExclusiveSum(we[k], lane, we[k]);
        if (lane == 0) {
            a.keptRecs[slot] = nRec; a.occCounts[slot] = nOcc; a.ceCounts[slot] = nCe; a.totalOcc[slot] = total;
            for (int k = 0; k < 8; k++) a.withError[(size_t)slot * 8 + k] = we[k];
        }
    }
}

// final per-read record list in the hit format of the C-ABI (kept records only, last range truncated)
__global__ void mergeFillKernel(MergeArgs a, const uint32_t *hitOffsets, DevHit *hits)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= a.nActive) return;
    const uint32_t groups = a.nCases * 2u;
    uint32_t beg, endUnused;
    readSlice(a, slot, groups, beg, endUnused);
    const uint32_t n = a.keptRecs[slot];
    uint32_t total = 0;
    DevHit *dst = hits + hitOffsets[slot];
    for (uint32_t k = 0, i = beg; k < n; i++) {
        if (!a.valid[i]) continue;
        const uint4 r = a.recs[a.ordIdx[i]];
        const uint32_t kind = (r.y >> 24) & 1u, nmis = (r.y >> 25) & 7u;
        const uint32_t grp = r.x % groups;
        uint32_t lo = r.z, hi = r.w;
        uint32_t cnt = kind ? 1u : hi - lo + 1u;
        if (total + cnt > a.maxOut) { cnt = a.maxOut - total; hi = lo + cnt - 1u; }
        total += cnt;
        DevHit h;
        h.a = lo; h.b = hi; h.meta = kind | ((grp & 1u) + 1u) << 8 | nmis << 16 | (grp >> 1) << 24;
        dst[k] = h;
        k++;
    }
}

// occurrences in the CPU path's final order: check-and-extend hits first, then every SA range expanded
// through the suffix array (transferAllSAToOcc, SAList.cpp:392-418; BWTFullSaValue, BWT.c:1726)
__global__ void decodeKernel(MergeArgs a, DevIndex ix, const uint32_t *occOffsets, uint32_t occBase, uint32_t occCap, uint32_t *occPos,
                             uint32_t *occMeta, uint32_t *flags)
{
    // One lane per read walks the read's slice.  An SA range of more than 16 suffixes -- a read inside a repeat with identical copies holds ONE record for a
    // thousand occurrences (8,000 in the paired ladder) -- is expanded by the whole wave, suffix q by lane q % 64: consecutive suffix-array entries, whole
    // lines, instead of a thousand dependent 4-byte loads in one lane while 63 wait (round 5: 121 ms of a 10 M-pair run were this loop).  So the walk is
    // wave-synchronous: every lane takes one slot of its slice per turn, nobody leaves before all are through.
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    const uint32_t groups = a.nCases * 2u;
    bool mine = slot < a.nActive && !(a.decodeMask && a.decodeMask[slot] != a.decodeWant);
    uint32_t beg = 0, end = 0, n = 0, o = 0, oSa = 0;
    if (mine) {
        readSlice(a, slot, groups, beg, end);
        if (a.longList && end - beg > S3_MERGE_LONG) mine = false;                     // decodeLongKernel's
    }
    if (mine && (uint64_t)occBase + occOffsets[slot + 1] > occCap) { atomicOr(flags, S3_FLAG_OCC_OVERFLOW); mine = false; }
    if (mine) { n = a.keptRecs[slot]; o = occBase + occOffsets[slot]; oSa = o + a.ceCounts[slot]; }
    uint32_t total = 0, k = 0, i = beg;
    uint32_t w0 = beg & ~7u, bits = 0;                                        // the validity bits of the eight slots from w0 on that are still to be looked at
    bool loaded = false;
    while (__any(mine && k < n)) {
        const bool turn = mine && k < n;
        // a turn either fetches the next eight validity bytes (one load) or takes the next record they show
        if (turn && !bits) {
            if (loaded) w0 += 8u;
            // (a slice that holds fewer records than keptRecs says -- a stale pass decoded again -- must end the lane with an error, not spin the wave forever)
            if (w0 >= end) { atomicOr(flags, S3_FLAG_OCC_OVERFLOW); mine = false; }
            else { bits = validBits8(a.valid, w0, beg, end); loaded = true; }
        }
        const bool have = turn && bits != 0u;
        uint4 r = make_uint4(0, 0, 0, 0);
        if (have) {
            i = w0 + (uint32_t)__ffs((int)bits) - 1u;
            bits &= bits - 1u;
            r = a.recs[a.ordIdx[i]];
        }
        const uint32_t kind = (r.y >> 24) & 1u, nmis = (r.y >> 25) & 7u;
        const uint32_t meta = (((r.x % groups) & 1u) + 1u) | nmis << 8;
        uint32_t cnt = 0, dst = 0;
        if (have) {
            k++;
            if (kind) { occPos[o] = r.z; occMeta[o] = meta; o++; total++; }
            else {
                cnt = r.w - r.z + 1u;
                if (total + cnt > a.maxOut) cnt = a.maxOut - total;
                dst = oSa;
                if (cnt <= 16u) for (uint32_t q = 0; q < cnt; q++) { occPos[dst + q] = ix.sa[r.z + q]; occMeta[dst + q] = meta | 0x10000u; }   // bit 16: came out of an SA range
                oSa += cnt; total += cnt;
            }
        }
        unsigned long long big = __ballot(have && !kind && cnt > 16u);
        while (big) {
            const int l = __ffsll((long long)big) - 1;
            big &= big - 1;
            const uint32_t z = (uint32_t)__shfl((int)r.z, l), cc = (uint32_t)__shfl((int)cnt, l), d = (uint32_t)__shfl((int)dst, l), m = (uint32_t)__shfl((int)meta, l) | 0x10000u;
            for (uint32_t q = lane; q < cc; q += 64u) { occPos[d + q] = ix.sa[z + q]; occMeta[d + q] = m; }
        }
    }
}

// decodeKernel for the listed (long) reads: one wave per read (see mergeCountLongKernel); the lanes place their entries with prefix sums, the
// occurrences of an SA range are expanded by the lane that holds it
__global__ void __launch_bounds__(256) decodeLongKernel(MergeArgs a, DevIndex ix, const uint32_t *occOffsets, uint32_t occBase, uint32_t occCap, uint32_t *occPos,
                                                        uint32_t *occMeta, uint32_t *flags)
{
    __shared__ uint32_t sIdx[4][S3_LONG_TRIP];
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *idx = sIdx[threadIdx.x >> 6];
    const uint32_t waveId = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t nLong = *a.longCount, groups = a.nCases * 2u;
    for (uint32_t li = waveId; li < nLong; li += nWaves) {
        const uint32_t slot = a.longList[li];
        if (a.decodeMask && a.decodeMask[slot] != a.decodeWant) continue;
        uint32_t beg, end;
        readSlice(a, slot, groups, beg, end);
        if ((uint64_t)occBase + occOffsets[slot + 1] > occCap) { if (lane == 0) atomicOr(flags, S3_FLAG_OCC_OVERFLOW); continue; }
        uint32_t o = occBase + occOffsets[slot], oSa = o + a.ceCounts[slot];  // wave-uniform cursors: check-and-extend hits / expanded SA ranges
        uint32_t total = 0;
        const uint32_t tb0 = beg & ~7u;
        uint2 next = loadTripValid(a, tb0, end, lane);
        for (uint32_t tb = tb0; tb < end && total < a.maxOut; tb += S3_LONG_TRIP) {
          const uint2 cur = next;
          if (tb + S3_LONG_TRIP < end) next = loadTripValid(a, tb + S3_LONG_TRIP, end, lane);
          const uint32_t V = tripRecords(cur, tb, beg, end, lane, idx);
          for (uint32_t r0 = 0; r0 < V && total < a.maxOut; r0 += 64u) {
            const bool valid = r0 + lane < V;
            const uint4 r = a.recs[valid ? a.ordIdx[idx[r0 + lane]] : 0u];
            const uint32_t kind = (r.y >> 24) & 1u, nmis = (r.y >> 25) & 7u;
            uint32_t cnt = valid ? (kind ? 1u : r.w - r.z + 1u) : 0u;
            if (cnt > a.maxOut) cnt = a.maxOut;
            uint32_t sum = 0;
            const uint32_t pre = waveExclusiveSum(cnt, lane, sum);
            const bool incl = valid && total + pre < a.maxOut;
            uint32_t c = 0;
            if (incl) { c = cnt; if (total + pre + c > a.maxOut) c = a.maxOut - total - pre; }
            uint32_t nKind = 0, nSa = 0;
            const uint32_t preKind = waveExclusiveSum(incl && kind ? 1u : 0u, lane, nKind);
            const uint32_t preSa = waveExclusiveSum(incl && !kind ? c : 0u, lane, nSa);
            if (incl) {
                const uint32_t strand = ((r.x % groups) & 1u) + 1u;
                const uint32_t meta = strand | nmis << 8;
                if (kind) { occPos[o + preKind] = r.z; occMeta[o + preKind] = meta; }
                else if (c <= 8u) for (uint32_t q = 0; q < c; q++) { occPos[oSa + preSa + q] = ix.sa[r.z + q]; occMeta[oSa + preSa + q] = meta | 0x10000u; }
            }
            // an SA range of more than a few suffixes (up to maxOut identical copies) is expanded by the whole wave, suffix q by lane q % 64
            {
                unsigned long long big = __ballot(incl && !kind && c > 8u);
                const uint32_t strandB = ((r.x % groups) & 1u) + 1u, metaB = strandB | nmis << 8 | 0x10000u, dstB = oSa + preSa;
                while (big) {
                    const int l = __ffsll((long long)big) - 1;
                    big &= big - 1;
                    const uint32_t z = (uint32_t)__shfl((int)r.z, l), cc = (uint32_t)__shfl((int)c, l), d = (uint32_t)__shfl((int)dstB, l), m = (uint32_t)__shfl((int)metaB, l);
                    for (uint32_t q = lane; q < cc; q += 64u) { occPos[d + q] = ix.sa[z + q]; occMeta[d + q] = m; }
                }
            }
            o += nKind; oSa += nSa;
            total = total + sum > a.maxOut ? a.maxOut : total + sum;
          }
          __builtin_amdgcn_wave_barrier();
        }
    }
}

// all-best ladder bookkeeping (hostKernel single-end branch, CPUfunctions.cpp:1755-1801,1858-1867):
//   state 0: no hit          -> carried to the next phase (unaligned after the last)
//   state 1: resolved        -> reported in this phase
//   state 2: best == mm      -> deferred to phase mm+1 (non terminal), or re-searched with the mm+1 model (terminal)
__global__ void classifyKernel(uint32_t nActive, const uint32_t *totalOcc, const uint32_t *withError,
                               uint32_t mm, uint32_t allowSpecial, uint8_t *state)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nActive) return;
    uint8_t st = 1;
    if (totalOcc[slot] == 0) st = 0;
    else if (allowSpecial) {
        uint32_t best = 999;
        for (uint32_t k = 0; k <= mm; k++) if (withError[(size_t)slot * 8 + k] > 0) { best = k; break; }
        if (best == mm) st = 2;
    }
    state[slot] = st;
}

// One 64-byte line per 128 BWT characters (sample k at position p = 128 k):
//   [0] Occ(c, p) + C[c], c = 0..3
//   [1] the 64 characters BEFORE the sample in REVERSED order (character t = BWT[p - 1 - t]), 16 per word, first in the top bits
//   [2] the 64 characters from the sample on (character t = BWT[p + t])
//   [3] .x / .y: counts of the symbols 1, 2, 3 (one byte each, bits 0-7 / 8-15 / 16-23) among the first 32 characters of [1] / [2]
// so a rank query is: the sampled counts +/- the symbol counts of the first d <= 64 characters of ONE half, and with [3] never more than
// 32 characters (two words) have to be counted.
__global__ void relayoutKernel(const uint32_t *bwt, size_t bwtWords, const uint32_t *occ, size_t occRecs, uint4 *lines)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= occRecs) return;
    auto word = [&](long long idx) -> uint32_t { return (idx >= 0 && (size_t)idx < bwtWords) ? bwt[idx] : 0u; };
    auto rev16 = [](uint32_t x) -> uint32_t { x = __brev(x); return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); };
    auto counts = [](uint32_t a, uint32_t b) -> uint32_t {
        uint32_t c1 = 0, c2 = 0, c3 = 0;
        for (int t = 0; t < 32; t++) {
            const uint32_t ch = ((t < 16 ? a : b) >> (30 - 2 * (t & 15))) & 3u;
            c1 += ch == 1; c2 += ch == 2; c3 += ch == 3;
        }
        return c1 | c2 << 8 | c3 << 16;
    };
    const long long w0 = (long long)k * 8;
    uint4 o, prev, next, pad;
    o.x = occ[4 * k]; o.y = occ[4 * k + 1]; o.z = occ[4 * k + 2]; o.w = occ[4 * k + 3];
    prev.x = rev16(word(w0 - 1)); prev.y = rev16(word(w0 - 2)); prev.z = rev16(word(w0 - 3)); prev.w = rev16(word(w0 - 4));
    next.x = word(w0); next.y = word(w0 + 1); next.z = word(w0 + 2); next.w = word(w0 + 3);
    pad.x = counts(prev.x, prev.y); pad.y = counts(next.x, next.y); pad.z = pad.w = 0;
    lines[4 * k] = o; lines[4 * k + 1] = prev; lines[4 * k + 2] = next; lines[4 * k + 3] = pad;
}

// ---------------------------------------------------------------------------------------------
// launchers (called from s3dp_api.cpp through s3dp_device.h)
// ---------------------------------------------------------------------------------------------
// Per read, both orientations of the searched prefix as MSB-first words (16 characters per word, first one in the top bits), the layout
// the wave-cooperative search kernel keeps in LDS:  readsT[r * 2 S + strand * S + w], strand 0 = the read, 1 = its reverse complement
// (3 - c, DV-Kernel.h:35), S = words per orientation.  `lengths` are the SEARCHED lengths (the seed length in long-read mode).
__global__ void orientReadsKernel(const uint32_t *queries, const uint32_t *lengths, uint32_t *readsT, uint32_t n, uint32_t wpq, uint32_t S)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n * 2u * S) return;
    const uint32_t r = (uint32_t)(t / (2u * S)), k = (uint32_t)(t % (2u * S)), strand = k / S, w = k % S;
    const uint32_t *src = queries + (size_t)(r >> 5) * 32u * wpq + (r & 31u);
    const uint32_t len = lengths[r];
    uint32_t out = 0;
    for (uint32_t c = 0; c < 16u; c++) {
        const uint32_t pos = w * 16u + c;
        uint32_t ch = 0;
        if (pos < len) {
            const uint32_t q = strand ? len - 1u - pos : pos;
            ch = (src[(size_t)(q >> 4) * 32u] >> ((q & 15u) * 2u)) & 3u;
            if (strand) ch = 3u - ch;
        }
        out |= ch << (30u - 2u * c);
    }
    readsT[t] = out;
}

void launchOrientReads(const uint32_t *queries, const uint32_t *lengths, uint32_t *readsT, uint32_t n, uint32_t wpq, uint32_t S, hipStream_t s)
{
    const uint64_t total = (uint64_t)n * 2u * S;
    if (!total) return;
    hipLaunchKernelGGL(orientReadsKernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, queries, lengths, readsT, n, wpq, S);
}

void launchRelayout(const uint32_t *bwt, size_t bwtWords, const uint32_t *occ, size_t occRecs, uint4 *lines, hipStream_t s)
{
    const unsigned blocks = (unsigned)((occRecs + 255) / 256);
    hipLaunchKernelGGL(relayoutKernel, dim3(blocks), dim3(256), 0, s, bwt, bwtWords, occ, occRecs, lines);
}

// ---------------------------------------------------------------------------------------------
// long lookup tables: level k+1 from level k by ONE exact step for each of the four characters, computed exactly as the search kernels step
// (same rank, same child ranges), so that an item that starts from the longer table is in the state it would have reached by walking.
// Entry c * 4^k + x of level k+1 extends entry x of level k with character c: in the index order of the table (read order for the backward
// table, reversed for the bidirectional one) the new character is the most significant -- inputs, outputs and the rank lines touched are all
// visited in increasing order: the build streams.
// ---------------------------------------------------------------------------------------------
__global__ void ltBaseBackKernel(DevIndex ix, uint2 *out, uint32_t n)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    out[x] = make_uint2(x ? ix.lkt[x - 1] + 1u : 1u, ix.lkt[x]);          // BWTExactModelBackward_Lookup2, BGS-HostAlgnmtAlgo2.cpp:311-330
}

__global__ void ltExtendBackKernel(DevIndex ix, const uint2 *in, uint2 *out, uint32_t nIn)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;               // one rank pair serves the four extensions of an entry
    if (x >= nIn) return;
    const uint2 g = in[x];
    uint2 o[4] = {make_uint2(1u, 0u), make_uint2(1u, 0u), make_uint2(1u, 0u), make_uint2(1u, 0u)};   // a dead range is never stepped
    if (g.x <= g.y) {
        const uint4 oL = rankAll(ix.fwd, g.x), oR = rankAll(ix.fwd, g.y + 1u);
        o[0] = make_uint2(oL.x + 1u, oR.x); o[1] = make_uint2(oL.y + 1u, oR.y); o[2] = make_uint2(oL.z + 1u, oR.z); o[3] = make_uint2(oL.w + 1u, oR.w);
    }
#pragma unroll
    for (int c = 0; c < 4; c++) out[(size_t)c * nIn + x] = o[c];
}

__global__ void ltBaseBiKernel(DevIndex ix, uint4 *out, uint32_t n, int K)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;               // the K characters reversed: character 0 in the lowest bits
    if (t >= n) return;
    uint32_t key = 0;
    for (int j = 0; j < K; j++) key |= ((t >> (2 * j)) & 3u) << (2 * (K - 1 - j));
    uint32_t l = key ? ix.lkt[key - 1] + 1u : 1u, rr_ = ix.lkt[key], rl = t ? ix.rlkt[t - 1] + 1u : 1u, rrr = ix.rlkt[t];
    if (rr_ - l < rrr - rl) rl = rrr - (rr_ - l);                          // BWTExactModelForward_Lookup2 :349-356, 32-bit arithmetic as there
    else if (rr_ - l > rrr - rl) l = rr_ - (rrr - rl);
    out[t] = make_uint4(l, rr_, rl, rrr);
}

__global__ void ltExtendBiKernel(DevIndex ix, const uint4 *in, uint4 *out, uint32_t nIn)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nIn) return;
    const uint4 g = in[t];
    uint4 o[4] = {make_uint4(1u, 0u, 1u, 0u), make_uint4(1u, 0u, 1u, 0u), make_uint4(1u, 0u, 1u, 0u), make_uint4(1u, 0u, 1u, 0u)};
    if (g.x <= g.y) {
        const uint4 oL = rankAll(ix.rev, g.z), oR = rankAll(ix.rev, g.w + 1u);
        const uint32_t d1 = oR.y - oL.y, d2 = oR.z - oL.z, d3 = oR.w - oL.w;
        // the passive (forward-index) range shrinks by the occurrences of the larger characters
        const uint32_t al[4] = {oL.x + 1u, oL.y + 1u, oL.z + 1u, oL.w + 1u}, ar[4] = {oR.x, oR.y, oR.z, oR.w}, oc[4] = {d1 + d2 + d3, d2 + d3, d3, 0u};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t pr = g.y - oc[c], pl = pr - (ar[c] - al[c]);
            o[c] = make_uint4(pl, pr, al[c], ar[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) out[(size_t)c * nIn + t] = o[c];
}

int launchLongLookup(DevIndex &d, int extra, uint2 **backOut, uint4 **biOut, hipStream_t s)
{
    *backOut = nullptr; *biOut = nullptr;
    if (extra < 1 || extra > 3 || d.ltSize + extra > 16) return (int)hipErrorInvalidValue;
    const int K0 = d.ltSize;
    uint2 *back[4] = {nullptr, nullptr, nullptr, nullptr};
    uint4 *bi[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int lv = 0; lv <= extra && e == hipSuccess; lv++) {
        const size_t n = (size_t)1 << (2 * (K0 + lv));
        if ((e = hipMalloc((void **)&back[lv], n * sizeof(uint2))) != hipSuccess) break;
        if ((e = hipMalloc((void **)&bi[lv], n * sizeof(uint4))) != hipSuccess) break;
    }
    if (e == hipSuccess) {
        const uint32_t n0 = 1u << (2 * K0);
        hipLaunchKernelGGL(ltBaseBackKernel, dim3((n0 + 255) / 256), dim3(256), 0, s, d, back[0], n0);
        hipLaunchKernelGGL(ltBaseBiKernel, dim3((n0 + 255) / 256), dim3(256), 0, s, d, bi[0], n0, K0);
        for (int lv = 0; lv < extra; lv++) {
            const uint32_t nIn = 1u << (2 * (K0 + lv));
            hipLaunchKernelGGL(ltExtendBackKernel, dim3((nIn + 255) / 256), dim3(256), 0, s, d, back[lv], back[lv + 1], nIn);
            hipLaunchKernelGGL(ltExtendBiKernel, dim3((nIn + 255) / 256), dim3(256), 0, s, d, bi[lv], bi[lv + 1], nIn);
        }
        e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipGetLastError();
    }
    for (int lv = 0; lv < 4; lv++) {
        if (e == hipSuccess && lv == extra) continue;
        if (back[lv]) (void)hipFree(back[lv]);
        if (bi[lv]) (void)hipFree(bi[lv]);
    }
    if (e != hipSuccess) return (int)e;
    *backOut = back[extra]; *biOut = bi[extra];
    return 0;
}

// the listed reads count per unit: what the first pass left in their items' own counters (or never wrote) is void
__global__ void zeroListedCountsKernel(const uint32_t *heavyList, const uint32_t *heavyCount, uint32_t groups, uint32_t *itemCounts)
{
    const uint64_t total = (uint64_t)*heavyCount * groups;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (uint64_t)gridDim.x * blockDim.x)
        itemCounts[(size_t)heavyList[k / groups] * groups + (uint32_t)(k % groups)] = 0;
}

void launchZeroListedCounts(const uint32_t *heavyList, const uint32_t *heavyCount, uint32_t nHeavy, uint32_t groups, uint32_t *itemCounts, hipStream_t s)
{
    const uint64_t total = (uint64_t)nHeavy * groups;
    if (!total) return;
    const unsigned blocks = (unsigned)std::min<uint64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(zeroListedCountsKernel, dim3(blocks), dim3(256), 0, s, heavyList, heavyCount, groups, itemCounts);
}

void launchScatter(uint4 *recs, const uint32_t *recCount, uint32_t recCap, const uint32_t *itemOffsets, uint32_t *ordIdx, uint8_t *valid, const uint8_t *heavyFlags,
                   const uint32_t *heavyList, uint32_t groups, uint32_t unitBase, uint32_t unitsPerItem, ScatterDyn dyn, hipStream_t s)
{
    hipLaunchKernelGGL(scatterKernel, dim3(2048), dim3(256), 0, s, recs, recCount, recCap, itemOffsets, ordIdx, valid, heavyFlags, heavyList, groups, unitBase, unitsPerItem, dyn);
}

void launchMergeCount(const MergeArgs &a, hipStream_t s)
{
    if (!a.nActive) return;
    if (a.longList) (void)hipMemsetAsync(a.longCount, 0, 4, s);
    hipLaunchKernelGGL(mergeCountKernel, dim3((a.nActive + 127) / 128), dim3(128), 0, s, a);
    if (a.longList) hipLaunchKernelGGL(mergeCountLongKernel, dim3(2048), dim3(256), 0, s, a);
}

void launchMergeFill(const MergeArgs &a, const uint32_t *hitOffsets, DevHit *hits, hipStream_t s)
{
    if (!a.nActive) return;
    hipLaunchKernelGGL(mergeFillKernel, dim3((a.nActive + 127) / 128), dim3(128), 0, s, a, hitOffsets, hits);
}

void launchDecode(const MergeArgs &a, const DevIndex &ix, const uint32_t *occOffsets, uint32_t occBase, uint32_t occCap, uint32_t *occPos,
                  uint32_t *occMeta, uint32_t *flags, hipStream_t s)
{
    if (!a.nActive) return;
    hipLaunchKernelGGL(decodeKernel, dim3((a.nActive + 127) / 128), dim3(128), 0, s, a, ix, occOffsets, occBase, occCap, occPos, occMeta, flags);
    if (a.longList) hipLaunchKernelGGL(decodeLongKernel, dim3(2048), dim3(256), 0, s, a, ix, occOffsets, occBase, occCap, occPos, occMeta, flags);
}

void launchClassify(uint32_t nActive, const uint32_t *totalOcc, const uint32_t *withError, uint32_t mm, uint32_t allowSpecial, uint8_t *state,
                    hipStream_t s)
{
    if (!nActive) return;
    hipLaunchKernelGGL(classifyKernel, dim3((nActive + 255) / 256), dim3(256), 0, s, nActive, totalOcc, withError, mm, allowSpecial, state);
}

}  // namespace s3

// ---------------------------------------------------------------------------------------------
// measurement aid: random 64-byte-line gather rate of this GPU (the practical ceiling of a rank-query
// stream).  Every lane draws pseudo-random lines of a table, reads `loadsPerLine` x 16 B from each and
// keeps `ILP` independent lines in flight.  Not part of the alignment path.
// ---------------------------------------------------------------------------------------------
namespace s3 {

// streaming copy, 16 bytes per lane and load, UNROLL independent loads in flight per lane.  Three shapes are
// tried by s3_probe_copy, which reports the fastest: a grid-stride loop over a fixed grid with non-temporal
// accesses, the same with ordinary accesses, and one workgroup per contiguous piece (no loop: the grid covers
// the buffer, the shape of the guide's float4 copy).
typedef uint32_t s3_v4u __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT, bool COVER>
__global__ void __launch_bounds__(256) copyProbeKernel(const s3_v4u *__restrict__ src, s3_v4u *__restrict__ dst, size_t n16)
{
    if (COVER) {
        const size_t base = (size_t)blockIdx.x * (256 * UNROLL) + threadIdx.x;
        s3_v4u v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++)
            if (base + (size_t)k * 256 < n16) v[k] = NT ? __builtin_nontemporal_load(src + base + (size_t)k * 256) : src[base + (size_t)k * 256];
#pragma unroll
        for (int k = 0; k < UNROLL; k++)
            if (base + (size_t)k * 256 < n16) {
                if (NT) __builtin_nontemporal_store(v[k], dst + base + (size_t)k * 256);
                else dst[base + (size_t)k * 256] = v[k];
            }
        return;
    }
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        s3_v4u v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) v[k] = NT ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            if (NT) __builtin_nontemporal_store(v[k], dst + i + k * stride);
            else dst[i + k * stride] = v[k];
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

int copyProbeVariants() { return 6; }

void launchCopyProbe(const uint4 *src, uint4 *dst, size_t n16, int variant, hipStream_t s)
{
    const s3_v4u *a = (const s3_v4u *)src;
    s3_v4u *b = (s3_v4u *)dst;
    const dim3 blk(256);
    switch (variant) {
    case 0: hipLaunchKernelGGL((copyProbeKernel<4, true, false>), dim3(256 * 16), blk, 0, s, a, b, n16); break;
    case 1: hipLaunchKernelGGL((copyProbeKernel<4, false, false>), dim3(256 * 16), blk, 0, s, a, b, n16); break;
    case 2: hipLaunchKernelGGL((copyProbeKernel<8, false, false>), dim3(256 * 8), blk, 0, s, a, b, n16); break;
    case 3: hipLaunchKernelGGL((copyProbeKernel<1, false, true>), dim3((unsigned)((n16 + 255) / 256)), blk, 0, s, a, b, n16); break;
    case 4: hipLaunchKernelGGL((copyProbeKernel<4, false, true>), dim3((unsigned)((n16 + 1023) / 1024)), blk, 0, s, a, b, n16); break;
    default: hipLaunchKernelGGL((copyProbeKernel<4, true, true>), dim3((unsigned)((n16 + 1023) / 1024)), blk, 0, s, a, b, n16); break;
    }
}

template <int ILP>
__global__ void __launch_bounds__(256) gatherProbeKernel(const uint4 *table, uint64_t nLines, uint32_t itersPerLane, int loadsPerLine, int dependent,
                                                         uint32_t *sink)
{
    uint64_t x = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < itersPerLane; it++) {
        uint4 v[ILP];
        uint64_t line[ILP];
#pragma unroll
        for (int k = 0; k < ILP; k++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;                 // xorshift64
            line[k] = (x + (dependent ? acc : 0u)) % nLines;
        }
#pragma unroll
        for (int k = 0; k < ILP; k++) {
            if (loadsPerLine >= 10 && loadsPerLine < 20) {                                   // 32-byte granules: does the memory system fetch less than 64 B for them?
                const uint4 *p32 = table + (line[k] * 4 + ((x >> 40) & 2));     // one of the two 32-B halves of a random 64-B line
                v[k] = p32[0];
                if (loadsPerLine > 11) { uint4 w = p32[1]; v[k].x ^= w.x; v[k].y ^= w.w; }
                continue;
            }
            if (loadsPerLine >= 20) {                                   // 128-byte granules: both 64-B halves of a random 128-B aligned block (21: 256-B)
                const uint4 *p128 = table + (line[k] & ~(uint64_t)(loadsPerLine > 20 ? 3 : 1)) * 4;
                v[k] = p128[0];
                uint4 w = p128[4]; v[k].x ^= w.x; v[k].y ^= w.w;
                if (loadsPerLine > 20) { uint4 w2 = p128[8], w3 = p128[12]; v[k].z ^= w2.x ^ w3.y; }
                continue;
            }
            const uint4 *p = table + line[k] * 4;
            v[k] = p[0];
            if (loadsPerLine > 1) { uint4 w = p[1 + (line[k] & 1)]; v[k].x ^= w.x; v[k].y ^= w.w; }
            if (loadsPerLine > 2) { uint4 w = p[3]; v[k].z ^= w.y; }
        }
#pragma unroll
        for (int k = 0; k < ILP; k++) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;                           // keeps the loads alive
}

void launchGatherProbe(const uint4 *table, uint64_t nLines, uint32_t blocks, uint32_t itersPerLane, int ilp, int loadsPerLine, int dependent,
                       uint32_t *sink, hipStream_t s)
{
    if (ilp >= 8) hipLaunchKernelGGL(gatherProbeKernel<8>, dim3(blocks), dim3(256), 0, s, table, nLines, itersPerLane, loadsPerLine, dependent, sink);
    else if (ilp >= 4) hipLaunchKernelGGL(gatherProbeKernel<4>, dim3(blocks), dim3(256), 0, s, table, nLines, itersPerLane, loadsPerLine, dependent, sink);
    else if (ilp >= 2) hipLaunchKernelGGL(gatherProbeKernel<2>, dim3(blocks), dim3(256), 0, s, table, nLines, itersPerLane, loadsPerLine, dependent, sink);
    else hipLaunchKernelGGL(gatherProbeKernel<1>, dim3(blocks), dim3(256), 0, s, table, nLines, itersPerLane, loadsPerLine, dependent, sink);
}

}  // namespace s3

// The HIP runtime loads a file's code object when one of its kernels is first launched: 10 to 70 ms each (the files that instantiate rocPRIM sorts are the
// large ones), with every other thread's launches waiting -- 0.17 s of the first ladder of a process, device idle (profiles/README_r06.md section 8).
// s3_index_upload runs this on a thread of its own while the index goes up: one kernel per file is asked for its attributes, which loads the file's code object.
void s3::preloadKernels()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, (const void *)saExpandKernel);
    (void)hipGetLastError();
    preloadFirst(); preloadUnit(); preloadCe(); preloadApi(); preloadLadder(); preloadFlank(); preloadPair(); preloadRoute(); preloadSeed(); preloadDp(); preloadDpApi();
}
