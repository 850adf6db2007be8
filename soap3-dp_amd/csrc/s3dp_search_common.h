// s3dp_search_common.h -- what the search kernels share: s3dp_first.hip (first pass, trunk pass), s3dp_unit.hip (unit pass), s3dp_ce.hip
// (check-and-extend kernel): cooperative line ids, DFS frames, the model step of a lane in registers, and the rank arithmetic on the 64-byte rank
// lines (layout: relayoutKernel, s3dp_kernels.hip).
//
// THE SEARCH, in one place.  Same results, record for record, as the reference CPU path (BGS-HostAlgnmtAlgo2.cpp); different machine mapping.
//   * Measurement on MI355X (tools/hbm_probe.py) shows that a rank-query stream is bound by the number of vector-memory REQUESTS, not by DRAM bytes:
//     two 16-B lane loads into the same 64-B line cost as much as two lines.  So a rank query is ONE coalesced 64-B request: the index is re-laid
//     out at upload into one line per 128 BWT characters (counts, the half-block before the sample reversed, the half-block after it, precounts), 4
//     adjacent lanes load the 4 x 16 B of one owner's line (the owner's address travels by ds_bpermute), the line is staged in LDS and read back
//     by its owner -- "Occ/rank tables staged through LDS for coalesced HBM reads".
//   * The grids are persistent; a lane runs a flat state machine and takes new work from a queue when it is done.  Every trip of the loop first
//     ISSUES all memory operations of all lanes (rank lines, lookup entries, read words, the next stack frame) without a branch in between, waits
//     ONCE, then every lane advances its own search (pure ALU + LDS) until it needs memory again.
//   * Work item of the first pass = a read: its (case, strand) items in the CPU path's order, each from the longest lookup table the index holds.
//     Check-and-extend (a small SA range inside a model's check window is verified against the text instead of walked) runs in its own kernel:
//     the stepping lane writes a TASK record, reserves one sequence number per candidate and goes on; accepted candidates are written under the
//     reserved numbers, so the record stream keeps the CPU path's order; rejected ones leave holes the ordering pass skips.
//   * EARLY check-and-extend: the first step of every case is an exact walk of 20-60 characters, and behind the lookup table almost every range
//     is ONE suffix wide; a lane that finds a single suffix inside the lookup step hands it over at once (S3_REC_EARLY) when the next step would
//     start check-and-extend at its first position anyway; the task also verifies what is left of the lookup step, inside the text.  Exact steps
//     map a range onto a range order-preservingly, so survivors come out in the order the CPU path emits them.  Ranges that may hold a padded
//     lookup-table element (DevIndex::padBack / padBi) are walked, as the CPU path walks them.
//   * HEAVY reads.  A read inside a repeat family with 10^5 copies has a 4-mismatch neighbourhood of 10^5 rank queries: one lane per read leaves
//     the chip idle behind a few lanes.  A read that reserves more candidates than in.heavyLimit or costs more rank queries than in.heavyRank
//     leaves the first pass (listed), and the SECOND pass takes the listed reads apart: its work unit is the part of ONE (read, case, strand)
//     item's search tree whose FIRST mismatch sits at one given position of the first step that allows mismatches (or that has none there).  The
//     depth-first order of the CPU path visits exactly these subtrees one after the other (at every position: the mismatch children first,
//     ascending, then the continuation), so the units of an item, numbered by position, ARE its sequence order: each unit counts its records from
//     zero and the counting scatter puts them behind each other -- a thousand lanes per heavy read, no sort.  The TRUNK pass (one lane per listed
//     read and item) walks the item without taking a mismatch in that step and leaves the range it holds at every position; the UNIT pass (one
//     lane per read, item and position) starts from those.  Since round 5 a unit is a GROUP of consecutive positions (two by default): the lane walks the
//     trunk's path through its group -- writing nothing there: those records are the trunk's --, takes the mismatch children at every position and drops the
//     continuation at the group's last one (s3dp_unit.hip, "GROUPS"): half the unit starts, the same order.
//   * THE CUT.  The CPU path stops a read at MaxOutputPerRead occurrences.  The lanes of the unit pass book the SA-range occurrences
//     they hold for sure (their check-and-extend candidates go to ceKernel, as the first pass's do) in counters the units of their read share (itemAcc per item, the spare word of the trunk states per position), and
//     everything in sequence order up to a unit that adds up to MaxOutputPerRead puts the read's cut behind that unit (SearchOut::readCut); a lane
//     that finds itself behind its read's cut stops.  The merge applies the cut exactly either way.
//   * THE TAIL.  Once the queue is empty, an idle lane of the unit pass adopts the oldest frame of a busy lane of its wave; the frame's records
//     count in a counter of their own chained behind the giver's (SearchOut::chainNext) -- s3dp_unit.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "s3dp_device.h"
#include "s3dp_model.h"

namespace s3 {

namespace v2 {

enum : uint32_t { ST_DONE = 0, ST_STEP = 1, ST_TRUNK = 2, ST_LOAD = 4, ST_LT = 5, ST_KEYS = 6, ST_RESUME = 7, ST_CE_SA = 8, ST_CE_TEXT = 9 };
constexpr uint32_t ID_NONE = 0x7FFFFFFFu;
constexpr uint32_t ID_TEXT = 0x40000000u;                 // cooperative line ids: bit 31 = reverse BWT, ID_TEXT | k = 64-byte line k of the packed text
constexpr int STACK = 20;

// 16 bytes from an address put together out of two 32-bit halves (the owner's line address travels by ds_bpermute): said to be GLOBAL memory.  A pointer made
// from integers is a generic one, and the compiler then issues FLAT loads -- which count on the LDS counter too, so the wait for the next ds_bpermute behind the
// first of the trip's four line loads LOOKED like a wait for that load in the ISA.  Measured (round 6): no difference -- the hardware takes a flat load that turns
// out to be global off the LDS counter at once -- but the global form needs no 64-bit additions and keeps the trip's waits on the memory counter alone.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// (byteOff < 64 and the address is that of a 64-byte line: the offset is OR-ed into the low half, no 64-bit addition)
__device__ __forceinline__ uint4 loadGlobal16(uint32_t lo, uint32_t hi, uint32_t byteOff)
{
    const unsigned long long a = ((unsigned long long)hi << 32) | (lo | byteOff);
    const u32x4_t v = *(const __attribute__((address_space(1))) u32x4_t *)a;
    return make_uint4(v.x, v.y, v.z, v.w);
}

struct Frame { uint32_t fl, fr, rl, meta; };        // rr = rl + (fr - fl);  meta = s | i << 3 | ins << 14 | occ << 17

__device__ __forceinline__ uint64_t hiMask(uint32_t nChars) { return nChars >= 32 ? ~0ULL : (nChars == 0 ? 0ULL : (~0ULL << (64 - 2 * nChars))); }
__device__ __forceinline__ uint64_t loMask(uint32_t nChars) { return nChars >= 32 ? ~0ULL : ((1ULL << (2 * nChars)) - 1ULL); }
// (two levels of selects: the chain c == 0 ? .. : c == 1 ? .. came out of the compiler as three nested branches on the execution mask -- round 6, read in the ISA of the
//  stepping kernels, which are bound by instruction issue)
__device__ __forceinline__ uint32_t sel4(const uint4 &v, uint32_t c)
{
    const uint32_t lo = (c & 1u) ? v.y : v.x, hi = (c & 1u) ? v.w : v.z;
    return (c & 2u) ? hi : lo;
}
// occurrences of the symbols ABOVE c in a range whose counts of symbols 1, 2, 3 are e1, e2, e3 (what the passive range shrinks by when the step takes symbol c)
__device__ __forceinline__ uint32_t occAbove(uint32_t c, uint32_t e1, uint32_t e2, uint32_t e3)
{
    return (c < 1u ? e1 : 0u) + (c < 2u ? e2 : 0u) + (c < 3u ? e3 : 0u);
}
__device__ __forceinline__ uint32_t pairRev(uint32_t x)   // reverse the order of the 16 2-bit characters of a word
{
    x = __brev(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}

// mask of the first n (0..16) characters of an MSB-first word, on the even bit of every character
__device__ __forceinline__ uint32_t topChars(uint32_t n)
{
    const uint32_t t = 2u * n;
    return (t >= 32u ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> t)) & 0x55555555u;
}

// Rank of all four symbols at index i from the line of sample k = (i + 63) >> 7 (layout: relayoutKernel, s3dp_kernels.hip):
//   occ  = line[0]; half = the two words of the half-block that hold the counted characters; sub = line[3].x (before) / .y (after).
// d = |i - 128 k| <= 64 characters next to the sample are counted; the first 32 of them come precounted in `sub`, so at most two
// words go through the popcount.  32-bit VALU only.
struct RankSel { uint32_t d, n; bool fwd, far; };     // far: d > 32 -> words z,w of the half + precounts, else words x,y
__device__ __forceinline__ RankSel rankSel(uint32_t i)
{
    const uint32_t p = ((i + 63u) >> 7) << 7;
    RankSel r;
    r.fwd = i > p;
    r.d = r.fwd ? i - p : p - i;
    r.far = r.d > 32u;
    r.n = r.far ? r.d - 32u : r.d;
    return r;
}
__device__ __forceinline__ uint4 rankFromWords(const RankSel &q, uint4 occ, uint32_t wa, uint32_t wb, uint32_t sub)
{
    const uint32_t n0 = q.n < 16u ? q.n : 16u, n1 = q.n - n0;
    const uint32_t m0 = topChars(n0), m1 = topChars(n1);
    const uint32_t la = wa & m0, ha = (wa >> 1) & m0, ba = la & ha;
    const uint32_t lb = wb & m1, hb = (wb >> 1) & m1, bb = lb & hb;
    uint32_t c3 = __popc(ba) + __popc(bb);
    uint32_t c2 = __popc(ha ^ ba) + __popc(hb ^ bb);
    uint32_t c1 = __popc(la ^ ba) + __popc(lb ^ bb);
    // (no branch on far / fwd: the precounts are masked, the sign is a factor of a 24-bit multiply-add)
    const uint32_t pre = q.far ? sub : 0u;
    c1 += pre & 0xFFu; c2 += (pre >> 8) & 0xFFu; c3 += (pre >> 16) & 0xFFu;
    const uint32_t c0 = q.d - c1 - c2 - c3;
    const int sg = q.fwd ? 1 : -1;
    uint4 v = occ;
    v.x += (uint32_t)__mul24((int)c0, sg); v.y += (uint32_t)__mul24((int)c1, sg); v.z += (uint32_t)__mul24((int)c2, sg); v.w += (uint32_t)__mul24((int)c3, sg);
    return v;
}

struct Range { uint32_t fl, fr, rl, rr; };

// Does the lookup key of an item belong to the PADDED k-mers of the tables it reads (DevIndex::padBack / padBi: 13-mers the file format's tables count for text
// positions next to the text's ends)?  Its lookup range may then hold an element that is no occurrence, and its forward and reverse ranges need not hold the
// same occurrences (BWTExactModelForward_Lookup2, BGS-HostAlgnmtAlgo2.cpp:349-356, makes their sizes equal by moving one end).  st = the case's step 0;
// window(p) = 16 characters of the item's strand from position p >= 0, MSB first.
// (Round 6: the 36 compares run behind a FILTER -- 2,048 bits in LDS, one per value of a key's low 11 bits, set for the keys of both lists by padFilterBuild at the
//  start of a workgroup.  A key is next to never a padded one, the stepping kernels are bound by instruction issue, and this test runs at the start of every item of
//  the first pass and of every unit: the lists themselves are walked only when some lane of the wave passes the filter.)
constexpr uint32_t S3_PAD_FILTER_BYTES = 256u;
__device__ __forceinline__ void padFilterBuild(const DevIndex &ix, uint32_t *filter)      // by the whole workgroup, before its first item
{
    for (uint32_t k = threadIdx.x; k < S3_PAD_FILTER_BYTES / 4u; k += blockDim.x) filter[k] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 12; j++) { const uint32_t k = ix.padBack[j]; if (k != 0xFFFFFFFFu) filter[(k >> 5) & 63u] |= 1u << (k & 31u); }
#pragma unroll
        for (int j = 0; j < 24; j++) { const uint32_t k = ix.padBi[j]; if (k != 0xFFFFFFFFu) filter[(k >> 5) & 63u] |= 1u << (k & 31u); }
    }
    __syncthreads();
}
template <class W>
__device__ __forceinline__ bool lookupPadded(const DevIndex &ix, const uint32_t *filter, int stType, int stStart, int stLen, W window)
{
    const int k0 = ix.ltSize;
    if (stLen < k0) return true;                                // shorter lookup steps read a whole block of entries
    const uint32_t m13 = (1u << (2 * k0)) - 1u;
    const bool back = stType == S3_T_BACK_LOOKUP;
    const uint32_t key = back ? (window(stStart - k0 + 1) >> (32 - 2 * k0)) & m13 : pairRev(window(stStart)) & m13;
    bool padded = false;
    if ((filter[(key >> 5) & 63u] >> (key & 31u)) & 1u) {
        if (back) {
#pragma unroll
            for (int j = 0; j < 12; j++) padded |= key == ix.padBack[j];
        } else {
#pragma unroll
            for (int j = 0; j < 24; j++) padded |= key == ix.padBi[j];
        }
    }
    return padded;
}

// the 48 characters of a read next to the matched part of a task (position `pos` = the next one to compare, direction dir), the adjacent one first; word(w) =
// word w of the item's strand, 16 characters MSB first, 0 for w outside the read (w comes as an unsigned number: the words "before" the read are huge indices) --
// characters before the read's start read as 0: they never count, ceKernel compares inside the step.
// (Round 6: the 48 characters are ONE stretch of the read -- [pos, pos + 48) forward, [pos - 47, pos] backward: four words, three funnel shifts, and for a backward step
//  the three windows reversed and in reverse order.  It used to be three windows cut out one by one in each direction: 250 vector instructions of every full trip that
//  writes a task record, a quarter of the first pass's full trip.)
template <class W>
__device__ __forceinline__ void adjacentWindows(int pos, int dir, W word, uint32_t &q0, uint32_t &q1, uint32_t &q2)
{
    const int P = dir > 0 ? pos : pos - 47;                          // first character of the stretch (may be negative)
    const int wb = P >> 4;                                           // floor(P / 16)
    const uint32_t sh = ((uint32_t)P & 15u) * 2u;
    const uint32_t w0 = word((uint32_t)wb), w1 = word((uint32_t)(wb + 1)), w2 = word((uint32_t)(wb + 2)), w3 = word((uint32_t)(wb + 3));
    // 16 characters from character (P & 15) of {hi, lo}: the pair shifted right by 32 - sh, low half (sh == 0: hi itself)
    const uint32_t a = sh ? __builtin_amdgcn_alignbit(w0, w1, 32u - sh) : w0;
    const uint32_t b = sh ? __builtin_amdgcn_alignbit(w1, w2, 32u - sh) : w1;
    const uint32_t c = sh ? __builtin_amdgcn_alignbit(w2, w3, 32u - sh) : w2;
    const bool fw = dir > 0;
    const uint32_t ra = pairRev(a), rb = pairRev(b), rc = pairRev(c);
    q0 = fw ? a : rc; q1 = fw ? b : rb; q2 = fw ? c : ra;
}

// the model step a lane is in, as the five words of S3Step with the check-and-extend threshold moved into the (unused) type byte
struct StepC {
    uint32_t w0, w1, w2, w3;
    __device__ __forceinline__ int start() const { return (int)(int16_t)(w0 & 0xFFFFu); }
    __device__ __forceinline__ int end() const { return (int)w0 >> 16; }
    __device__ __forceinline__ int ceStart() const { return (int)(int16_t)(w1 & 0xFFFFu); }
    __device__ __forceinline__ int ceEnd() const { return (int)w1 >> 16; }
    __device__ __forceinline__ int lm0() const { return (int)(int16_t)(w2 & 0xFFFFu); }
    __device__ __forceinline__ int len() const { return (int)w2 >> 16; }
    __device__ __forceinline__ int dir() const { return (int)(int8_t)(w3 & 0xFFu); }
    __device__ __forceinline__ int ceThr() const { return (int)((w3 >> 8) & 0xFFu); }
    __device__ __forceinline__ int minErr() const { return (int)((w3 >> 16) & 0xFFu); }
    __device__ __forceinline__ int maxErr() const { return (int)(w3 >> 24); }
};
__device__ __forceinline__ StepC packStep(const uint32_t *p)
{
    StepC c;
    c.w0 = p[0]; c.w1 = p[1]; c.w2 = p[2];
    c.w3 = (p[3] & 0xFFFF00FFu) | ((p[4] & 0xFFu) << 8);
    return c;
}


}  // namespace v2

}  // namespace s3
