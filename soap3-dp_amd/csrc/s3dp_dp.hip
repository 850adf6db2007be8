// s3dp_dp.hip -- semi-global affine-gap DP for gfx950: anti-diagonal wavefront, one problem per 64-lane wavefront.
//
// What is computed is the reference's DP (DV-DPfunctions.cu:61-512: DPScoreNHitPos, GenerateDPTable, the scheme-1/2
// driver SemiGlobalAligntment, GPUBacktrack), cell for cell: int running values inside a column, every stored cell
// clamped to int16 >= -32000, free left-clip re-seeding, anchors, first maximum in (column, row) order with a tie
// count, and the value-comparing traceback with its fixed priority.  How it is computed is different:
//   * the reference gives one CUDA thread a whole problem and keeps two short tables in global memory that the same
//     thread reads back (8 B of table traffic per cell, 97 k serial cells per thread for a 150 x 650 problem);
//   * here a wavefront sweeps the table by anti-diagonals: lane k owns read row 64c+1+k of the current 64-row chunk,
//     at step t it computes column t-k; H[j][i-1], the running insertion score, the diagonal and the window character
//     travel to the next lane by DPP (wave_shr:1), lane 0 taking the boundary row prefetched from LDS as the move's
//     "old" operand; the previous column's H/E stay in the lane's registers; the read is read once;
//   * what the traceback needs is written as ONE coalesced 128-B streaming store per wave step (H as int16, anti-diagonal-major) plus one
//     64-bit word per step: the traceback looks at E in a single test -- "did H(j,i) come out of an extended deletion: H(j,i) == gapExtend +
//     E(j-1,i)" -- and the lane that computes the cell has both values at hand, so the answer is a ballot bit (2.1 bytes per cell; the
//     reference writes 4 and reads them back);
//   * the maximum is tracked order-independently (max score, then smallest (column,row), plus tie count) and reduced
//     across the wave, which equals the reference's strictly-greater update in (column outer, row inner) order;
//   * the steady-state step carries no per-lane activity test and no clip / maximum code unless the chunk needs it
//     (dpStep's template flags): about 21 vector instructions per 64 cells.
// Bound: integer VALU; HBM traffic is 2.1 B written per cell.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "s3dp_device.h"

namespace s3 {

namespace dp {

constexpr int NEG_INF = -32000;
__device__ __forceinline__ int clampS(int x) { return x > NEG_INF ? x : NEG_INF; }

// the reference's packed sequences: 1-based, 16 chars per word MSB first, words of problem t interleaved by 32 (:57-59)
__device__ __forceinline__ uint32_t seqChar(const uint32_t *packed, uint32_t t, uint32_t maxLen, uint32_t i)
{
    const uint32_t base = (t >> 5) * (((maxLen + 15u) >> 4) << 5) + (t & 31u);
    return (packed[base + ((i >> 4) << 5)] >> ((15u - (i & 15u)) << 1)) & 3u;
}

struct Best { int score; uint32_t key; uint32_t count; };       // key = column << 16 | row  (smaller = earlier in the reference's order)

__device__ __forceinline__ void bestMerge(Best &a, int score, uint32_t key, uint32_t count)
{
    if (score > a.score) { a.score = score; a.key = key; a.count = count; }
    else if (score == a.score) { a.count += count; a.key = key < a.key ? key : a.key; }
}

// one DPP move across the whole wavefront (wave_shr:1) instead of a trip through the LDS crossbar
// value of the lane below; lane 0 keeps `first` (the boundary row, the same LDS word in every lane): bound_ctrl off leaves the old value there
__device__ __forceinline__ int waveShr1From(int first, int v) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xF, 0xF, false); }

// what a lane carries through one 64-row chunk
struct LaneState {
    int hPrev, ePrev;                 // H and E of this row in the previous column
    int upIn, fIn, dgIn, rIn;         // what this lane produced in the previous step (the lane above consumes it by DPP)
    int nb0, nb1, nb2, nr;            // boundary row + window character of the column lane 0 meets next (prefetched from LDS)
    int bS; uint32_t bT, bC;          // this row's maximum: score, step of its first occurrence, occurrences
    uint32_t hHold, xBits;            // table words in the making: H of the even step; the last steps' extended-deletion bits
};

struct ChunkConst {
    int go, ge, gi, match, mismatch;
    uint32_t lane, W, rc, refOffset, aL, dBase;
    int span;
    bool rowClip, rowBest;
    const unsigned char *refLds;
    int *bnd;
    uint32_t *tabH;                   // this chunk's part of the table: H of two steps per dword, [step / 2][lane] ...
    uint32_t *tabX;                   // ... and "H came out of an extended deletion", 32 steps per dword, [step / 32][lane]
};

// one anti-diagonal step.  Lanes outside the table (column < 1 while RAMP, column > W, rows beyond the read) compute on garbage that no
// cell inside the table ever consumes: the lane above a live cell was live one step earlier.  Their table words are written and never read.
//   CLIP: some row of the chunk lies inside the free left clip;  BEST: some row of the chunk may hold the maximum;
//   RAMP: steps 1..63, where lanes have not reached column 1 yet (their column-0 state must survive) and lane 63 writes no boundary
template <bool TABLE, bool CLIP, bool BEST, bool RAMP>
__device__ __forceinline__ void dpStep(const ChunkConst &k, LaneState &s, uint32_t t)
{
    const int upS = waveShr1From(s.nb0, s.upIn), fS = waveShr1From(s.nb1, s.fIn), dgS = waveShr1From(s.nb2, s.dgIn), rS = waveShr1From(s.nr, s.rIn);
    const uint32_t jn = t + 1u < k.W ? t + 1u : k.W;                  // uniform: the column lane 0 computes in the next step
    s.nb0 = k.bnd[3u * jn]; s.nb1 = k.bnd[3u * jn + 1u]; s.nb2 = k.bnd[3u * jn + 2u]; s.nr = k.refLds[jn];
    const int prevR = s.hPrev, ePrevIn = s.ePrev;
    int gapped = max(k.go + prevR, k.ge + s.ePrev);
    const int eSt = clampS(gapped);
    int f = max(k.ge + fS, k.go + upS);
    gapped = max(f, gapped);
    const int up = max(gapped, dgS + ((uint32_t)rS == k.rc ? k.match : k.mismatch));
    const int hSt = clampS(up);
    int dg = prevR;
    if (CLIP) {
        const uint32_t j = t - k.lane;
        const int init = (j + k.refOffset >= k.aL) ? NEG_INF : 0;
        const int prevInit = ((int)j <= 1) ? 0 : ((j - 1u + k.refOffset >= k.aL) ? NEG_INF : 0);
        if (k.rowClip) { f = max(init + k.gi, f); dg = max(prevInit, dg); }
    }
    if (RAMP) { const bool started = t > k.lane; s.hPrev = started ? hSt : s.hPrev; s.ePrev = started ? eSt : s.ePrev; }
    else { s.hPrev = hSt; s.ePrev = eSt; }
    s.upIn = up; s.fIn = f; s.dgIn = dg; s.rIn = rS;
    // written once and read only by the traceback kernel: streaming stores keep the table out of the caches' way
    if (TABLE) {
        // (t is uniform: the branches cost a scalar compare; every store is 64 dwords in a row)
        s.xBits = (s.xBits << 1) | (hSt == k.ge + ePrevIn ? 1u : 0u);
        if (t & 1u) __builtin_nontemporal_store(s.hHold | (uint32_t)hSt << 16, &k.tabH[(size_t)(t >> 1) * 64u + k.lane]);
        else s.hHold = (uint32_t)hSt & 0xFFFFu;
        if ((t & 31u) == 31u) __builtin_nontemporal_store(s.xBits, &k.tabX[(size_t)(t >> 5) * 64u + k.lane]);
    }
    if (BEST) {
        // columns jLo..W of the rows that may end the alignment; inside a row the columns come in increasing order, so the first maximum is the
        // reference's (strictly-greater update) and later equals only count
        const bool in = k.rowBest && (t - k.dBase) <= (uint32_t)k.span;
        const bool gt = in && up > s.bS, eq = in && up == s.bS;
        s.bC = gt ? 1u : s.bC + (eq ? 1u : 0u);
        s.bT = gt ? t : s.bT;
        s.bS = gt ? up : s.bS;
    }
    if (!RAMP) {
        // the chunk's last row becomes the boundary of the next chunk; lane 0 fetched this column 64 steps ago
        if (k.lane == 63u) { const uint32_t j = t - 63u; k.bnd[3u * j] = up; k.bnd[3u * j + 1u] = f; k.bnd[3u * j + 2u] = dg; }
    }
}

template <bool TABLE, bool CLIP, bool BEST>
__device__ __forceinline__ void chunkSweep(const DpArgs &a, uint32_t p, uint32_t c, uint32_t W, uint32_t L, uint32_t refOffset, int clipLt, int clipRtLoc,
                                           uint32_t aL, uint32_t aR, const unsigned char *refLds, int *bnd, uint32_t *table, uint32_t T, Best &best)
{
    ChunkConst k;
    k.go = a.gapOpen; k.ge = a.gapExtend; k.gi = k.go - k.ge; k.match = a.match; k.mismatch = a.mismatch;
    asm volatile("" : "+v"(k.match), "+v"(k.mismatch));              // the select between them wants vector registers: move once, not per step
    k.lane = threadIdx.x & 63u; k.W = W; k.refOffset = refOffset; k.aL = aL; k.refLds = refLds; k.bnd = bnd;
    k.tabH = TABLE ? table + (size_t)c * dpTableChunkWords(T) : nullptr;
    k.tabX = TABLE ? table + (size_t)c * dpTableChunkWords(T) + dpTableXBase(T) : nullptr;
    const uint32_t i = (c << 6) + 1u + k.lane;                       // this lane's read row
    const bool rowOk = i <= L;
    k.rc = rowOk ? seqChar(a.reads, p, a.maxReadLength, i) : 0u;
    const int jLo = max(1, (int)aR - (int)refOffset);                // first column that may hold the maximum (right anchor)
    k.span = (int)W - jLo;
    k.dBase = k.lane + (uint32_t)jLo;
    k.rowBest = BEST && rowOk && (int)i >= clipRtLoc && k.span >= 0;
    k.rowClip = CLIP && (int)i <= clipLt;
    LaneState s;
    // column 0 of this row (:165-183)
    if ((int)i <= clipLt) { s.hPrev = clampS(k.go); s.ePrev = clampS(k.go + k.gi); }
    else { const int up0 = k.gi + k.ge * ((int)i - (clipLt > 0 ? clipLt : 0)); s.hPrev = clampS(up0); s.ePrev = clampS(up0 + k.gi); }
    s.upIn = 0; s.fIn = 0; s.dgIn = 0; s.rIn = 0;
    s.bS = NEG_INF; s.bT = 0; s.bC = 0;
    s.hHold = 0; s.xBits = 0;
    const uint32_t j1 = W ? 1u : 0u;
    s.nb0 = bnd[3u * j1]; s.nb1 = bnd[3u * j1 + 1u]; s.nb2 = bnd[3u * j1 + 2u]; s.nr = refLds[j1];
    const uint32_t steps = W + 63u;
    uint32_t t = 1;
    for (; t <= 63u; t++) dpStep<TABLE, CLIP, BEST, true>(k, s, t);
    for (; t + 1u <= steps; t += 2u) { dpStep<TABLE, CLIP, BEST, false>(k, s, t); dpStep<TABLE, CLIP, BEST, false>(k, s, t + 1u); }
    if (t <= steps) dpStep<TABLE, CLIP, BEST, false>(k, s, t);
    if (TABLE) {
        // what the last steps left in the registers (steps is the last step taken)
        if (!(steps & 1u)) __builtin_nontemporal_store(s.hHold, &k.tabH[(size_t)(steps >> 1) * 64u + k.lane]);
        if ((steps & 31u) != 31u) __builtin_nontemporal_store(s.xBits << (31u - (steps & 31u)), &k.tabX[(size_t)(steps >> 5) * 64u + k.lane]);
    }
    if (BEST && s.bC) bestMerge(best, s.bS, ((s.bT - k.lane) << 16) | i, s.bC);
}

// one sweep over W columns (starting after refOffset) of problem p by the calling wavefront
template <bool TABLE>
__device__ void sweep(const DpArgs &a, uint32_t p, uint32_t W, uint32_t L, uint32_t refOffset, int clipLt, int clipRtLoc, uint32_t aL, uint32_t aR,
                      unsigned char *refLds, int *bnd, uint32_t *table, uint32_t T, Best &best)
{
    const uint32_t lane = threadIdx.x & 63u;
    const int gi = a.gapOpen - a.gapExtend;
    for (uint32_t j = lane; j <= W; j += 64u) refLds[j] = j ? (unsigned char)seqChar(a.dna, p, a.maxDNALength, j + refOffset) : 0;
    // row 0 boundary: H[j][0] = init(j), running open = init + gi, diagonal for row 1 = init(j-1)
    for (uint32_t j = lane; j <= W; j += 64u) {
        const int init = (j + refOffset >= aL) ? NEG_INF : 0;
        const int prevInit = (j <= 1) ? 0 : ((j - 1 + refOffset >= aL) ? NEG_INF : 0);
        bnd[3 * j] = init; bnd[3 * j + 1] = init + gi; bnd[3 * j + 2] = prevInit;
    }
    __builtin_amdgcn_wave_barrier();
    best.score = NEG_INF; best.key = 0xFFFFFFFFu; best.count = 0;
    const uint32_t chunks = (L + 63u) >> 6;
    for (uint32_t c = 0; c < chunks; c++) {
        // wave-uniform: does the chunk touch the free left clip / hold rows that may end the alignment?
        const bool clip = clipLt > (int)(c << 6);
        const bool mayEnd = (int)((c << 6) + 64u) >= clipRtLoc;
        if (clip) {
            if (mayEnd) chunkSweep<TABLE, true, true>(a, p, c, W, L, refOffset, clipLt, clipRtLoc, aL, aR, refLds, bnd, table, T, best);
            else chunkSweep<TABLE, true, false>(a, p, c, W, L, refOffset, clipLt, clipRtLoc, aL, aR, refLds, bnd, table, T, best);
        } else {
            if (mayEnd) chunkSweep<TABLE, false, true>(a, p, c, W, L, refOffset, clipLt, clipRtLoc, aL, aR, refLds, bnd, table, T, best);
            else chunkSweep<TABLE, false, false>(a, p, c, W, L, refOffset, clipLt, clipRtLoc, aL, aR, refLds, bnd, table, T, best);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // wave reduction: max score, then smallest key, summed count of the maxima
    for (int off = 32; off > 0; off >>= 1) {
        const int s2 = __shfl_xor(best.score, off);
        const uint32_t k2 = (uint32_t)__shfl_xor((int)best.key, off), c2 = (uint32_t)__shfl_xor((int)best.count, off);
        bestMerge(best, s2, k2, c2);
    }
}

}  // namespace dp

using namespace dp;

// SemiGlobalAligntment (:243-314): scheme 1 = one table sweep; scheme 2 = score-only sweep, then a windowed table sweep
__global__ void __launch_bounds__(256) dpFillKernel(DpArgs a)
{
    extern __shared__ __align__(16) unsigned char sh[];
    // the wavefront number is uniform: saying so keeps the problem's lengths, clips, anchors and loop counters in scalar registers
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const uint32_t p = a.first + blockIdx.x * (blockDim.x >> 6) + wv;      // 1, 2 or 4 problems per block: what the windows leave room for in LDS
    if (p >= a.n) return;
    const uint32_t perWave = a.ldsPerWave;
    unsigned char *refLds = sh + wv * perWave;
    int *bnd = (int *)(sh + wv * perWave + ((a.maxDNALength + 1u + 15u) & ~15u));
    const uint32_t clipLt = a.clipLt ? a.clipLt[p] : 0u, clipRt = a.clipRt ? a.clipRt[p] : 0u;
    const uint32_t aL = a.anchorLeft ? a.anchorLeft[p] : a.maxDNALength, aR = a.anchorRight ? a.anchorRight[p] : 0u;
    const uint32_t L = a.readLengths[p];
    uint32_t W = a.dnaLengths[p], refOffset = 0;
    Best best;
    if (a.scheme == 2) {
        sweep<false>(a, p, W, L, 0u, (int)clipLt, (int)(L - clipRt), aL, aR, refLds, bnd, nullptr, 0u, best);
        // a maximum that never rose above the initial -32000 leaves hitPos at 0 in the reference (strict > update, :131-138)
        const uint32_t hitPos = (best.count && best.score > NEG_INF) ? best.key >> 16 : 0u;
        if (W > a.maxDPTableLength - 1u) W = a.maxDPTableLength - 1u;
        if (hitPos > W) {
            refOffset = hitPos - W;
            if (refOffset >= aL) refOffset = aL > 0 ? aL - 1u : 0u;
        }
    }
    const uint32_t T = a.maxDPTableLength + 64u;
    uint32_t *table = a.table + (size_t)p * a.tableWordsPerProblem;
    sweep<true>(a, p, W, L, refOffset, (int)clipLt, (int)(L - clipRt), aL, aR, refLds, bnd, table, T, best);
    if (lane == 0) {
        const bool found = best.count && best.score > NEG_INF;     // ties with the initial -32000 only count (:228-236)
        const uint32_t hitPos = found ? best.key >> 16 : 0u, row = found ? best.key & 0xFFFFu : 0u;
        a.startOffsets[p] = refOffset;
        a.hitLocs[p] = hitPos;
        a.scores[p] = best.score;
        a.maxScoreCounts[p] = best.count;
        a.tableCols[p] = W;
        if (a.clipRt) a.clipRt[p] = found ? L - row : 0u;
    }
}

// Traceback (what GPUBacktrack computes, DV-DPfunctions.cu:316-512): from the maximum back to the read's first aligned base.
// The reference's walk decides every move by comparing table VALUES, in a fixed priority -- match/mismatch, then a deletion that opens here, then
// one that extends, then (inside the left-clip allowance) the two ways out, then an insertion that opens, else one that extends -- and inside a gap
// it follows the gap's own score downwards until the gap is seen to open.  The same decisions are taken here from H alone plus the one bit per
// cell the fill kernel recorded; the walk is organised as "where am I, what does the cell say" (View) and three small movers, one lane per
// problem (a walk is a chain of dependent reads: 64 problems per wave is what hides its latency).
namespace tb {
struct View {
    const DpArgs &a; uint32_t p, T, refOffset, aL, cl;
    const uint32_t *table;
    int go, ge, gi;
    // H(j, i) as the table holds it; column 0 and row 0 are never stored: they are the initial conditions (:165-198)
    __device__ int H(uint32_t j, uint32_t i) const
    {
        if (j == 0u) return clampS(i == 0u ? 0 : (i <= cl ? go : gi + ge * (int)(i - cl)));
        if (i == 0u) return (j + refOffset >= aL) ? NEG_INF : 0;
        const uint32_t c = (i - 1u) >> 6, k = (i - 1u) & 63u;
        const uint32_t t = j + k;                                    // the step that computed the cell
        const uint32_t w = table[(size_t)c * dpTableChunkWords(T) + (size_t)(t >> 1) * 64u + k];
        return (int)(int16_t)(uint16_t)((t & 1u) ? w >> 16 : w);
    }
    // did H(j, i) come out of an extended deletion, H(j,i) == gapExtend + E(j-1,i)?  (j >= 1; for j == 1 the previous column is the initial one)
    __device__ bool fromExtendedDeletion(uint32_t j, uint32_t i) const
    {
        if (i == 0u) return false;
        const uint32_t c = (i - 1u) >> 6, k = (i - 1u) & 63u;
        const uint32_t t = j + k;
        return (table[(size_t)c * dpTableChunkWords(T) + dpTableXBase(T) + (size_t)(t >> 5) * 64u + k] >> (31u - (t & 31u))) & 1u;
    }
};
}  // namespace tb

// the walk of problem p by the calling lane; score / hitLoc / startOffset / clipRt: what the fill left for p (handed over in registers where the fill kernel
// itself goes on to walk, read back from the arrays in dpBacktrackKernel)
__device__ __forceinline__ void dpBacktrackOne(const DpArgs &a, uint32_t p, int score, uint32_t hitLoc, uint32_t startOffset, uint32_t clipRt)
{
    unsigned char *pat = a.pattern + (size_t)p * a.patternLen;
    pat[0] = 0;
    if (score < a.cutoffs[p]) return;
    const uint32_t L = a.readLengths[p];
    const tb::View v{a, p, a.maxDPTableLength + 64u, startOffset, a.anchorLeft ? a.anchorLeft[p] : a.maxDNALength, a.clipLt ? a.clipLt[p] : 0u,
                     a.table + (size_t)p * a.tableWordsPerProblem, a.gapOpen, a.gapExtend, a.gapOpen - a.gapExtend};
    uint32_t out = 0;
    auto put = [&](uint32_t x) { if (out < a.patternLen) pat[out] = (unsigned char)x; ++out; };
    auto run = [&](char op, uint32_t n) { put((uint32_t)op); put('V'); put(n); };       // 'V'<n>: n - 1 more of the op before (decoded by the engines)
    uint32_t i = L - clipRt, j = hitLoc;                                                 // read position, window column (1-based, 0 = before the first)
    if (clipRt) run('S', clipRt);
    auto readAt = [&](uint32_t q) { return seqChar(a.reads, p, a.maxReadLength, q); };
    auto refAt = [&](uint32_t q) { return seqChar(a.dna, p, a.maxDNALength, v.refOffset + q); };
    auto rowInit = [&](uint32_t q) -> int { return (v.refOffset + q >= v.aL) ? NEG_INF : 0; };     // H(q, 0)
    enum Exit { NONE, CLIP_AFTER_MATCH, CLIP_AFTER_INSERT } exitBy = NONE;
    int cur = v.H(j, i);
    while (i > 0u && j > 0u && exitBy == NONE) {
        const bool same = refAt(j) == readAt(i);
        const int sub = same ? a.match : a.mismatch;
        int nxt;
        if (cur == sub + (nxt = v.H(j - 1u, i - 1u))) { put(same ? 'M' : 'm'); --j; --i; cur = nxt; continue; }
        if (cur == v.go + (nxt = v.H(j - 1u, i))) { put('D'); --j; cur = nxt; continue; }
        if (v.fromExtendedDeletion(j, i)) {
            // inside a deletion: its score falls by gapExtend per base until the column where it opened out of H
            put('D'); --j;
            int gap = (int16_t)(cur - v.ge);
            for (;;) {
                if (!(i > 0u && j > 0u)) { cur = gap; break; }
                put('D'); --j;
                if (gap == v.go + (nxt = v.H(j, i))) { cur = nxt; break; }
                gap = (int16_t)(gap - v.ge);
            }
            continue;
        }
        if (i <= v.cl + 1u) {                                                              // the read's start may be clipped away from here
            if (cur == rowInit(j - 1u) + sub && j >= 1u) { exitBy = CLIP_AFTER_MATCH; break; }
            if (cur == rowInit(j) + v.go) { exitBy = CLIP_AFTER_INSERT; break; }
        }
        if (cur == v.go + (nxt = v.H(j, i - 1u))) { put('I'); --i; cur = nxt; continue; }
        // inside an insertion: likewise, along the read
        put('I'); --i;
        int gap = (int16_t)(cur - v.ge);
        for (;;) {
            if (!(i > 0u && j > 0u)) { cur = gap; break; }
            if (i <= v.cl + 1u && gap == rowInit(j) + v.go) { cur = gap; exitBy = CLIP_AFTER_INSERT; break; }
            put('I'); --i;
            if (gap == v.go + (nxt = v.H(j, i))) { cur = nxt; break; }
            gap = (int16_t)(gap - v.ge);
        }
    }
    if (j == 0u) {
        // ran off the window's left edge: what is left of the read is clipped as far as allowed, the rest counts as inserted
        const uint32_t clipped = v.cl < i ? v.cl : i;
        if (clipped < i) run('I', i - clipped);
        run('S', clipped);
    } else if (exitBy == CLIP_AFTER_INSERT) { put('I'); run('S', i - 1u); }
    else if (exitBy == CLIP_AFTER_MATCH) { put(refAt(j) == readAt(i) ? 'M' : 'm'); run('S', i - 1u); --j; }
    put(0);
    if (a.patMax) atomicMax(a.patMax, out < a.patternLen ? out : a.patternLen);        // bytes of the longest pattern, terminator included: what the host fetches per row
    a.hitLocs[p] = v.refOffset + j;
}

// ---- two problems per wavefront ---------------------------------------------------------------------------------------------------------------------------
// Every value of the recurrence fits 16 bits: stored cells are clamped to [-32000, 32767] by the reference itself, and the running values of a column stay
// within -32000 + 2 gapOpen + mismatch + gapExtend ... L x match (dpPairable checks the scores).  So a wave sweeps TWO problems at once, problem 2q in the low
// half and problem 2q + 1 in the high half of every register, with the packed 16-bit instructions of the vector ALU (v_pk_add_i16, v_pk_max_i16,
// v_pk_mad_i16 / _u16, v_pk_min_u16): one instruction per step and lane serves two cells.  Both problems take the steps of the longer window and the
// chunks of the longer read; cells outside a problem's own table compute on values no cell inside consumes, as before.  What has no packed form is written
// as arithmetic on the halves: "equal" is min_u16(a ^ b, 1), a select is a bit-field insert under a mask made of that.  The table each problem gets is the one
// the one-problem kernel writes, bit for bit (two steps per dword, 32 steps of bits per dword): the traceback and everything behind it do not know.
// Scheme 1 only (scheme 2 and score ranges beyond 16 bits: dpFillKernel).
namespace dp2 {

typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t bitsOf(s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s2 asS2(uint32_t v) { return __builtin_bit_cast(s2, v); }
__device__ __forceinline__ s2 both(int lo, int hi) { s2 r; r.x = (short)lo; r.y = (short)hi; return r; }
__device__ __forceinline__ s2 pmax(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }
// (the compiler turns min(x, 1) on halves into two compares, two selects and a byte permute, and a multiply-add on halves likewise: the three packed
//  instructions that do them in one are asked for by name)
__device__ __forceinline__ uint32_t pkMinU16(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t pkMadI16(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ uint32_t pkMadU16(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// per half: 0 where the half is zero, 1 elsewhere
__device__ __forceinline__ uint32_t nonZero01(uint32_t x) { return pkMinU16(x, 0x00010001u); }
// per half: 0xFFFF where the half of x is non-zero / negative
__device__ __forceinline__ uint32_t nonZeroMask(uint32_t x) { const s2 z = {0, 0}; return bitsOf(z - asS2(nonZero01(x))); }
__device__ __forceinline__ uint32_t negativeMask(uint32_t x) { const s2 sh = {15, 15}; return bitsOf(asS2(x) >> sh); }
__device__ __forceinline__ uint32_t pick(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }          // v_bfi_b32
__device__ __forceinline__ uint32_t shr1From(uint32_t first, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xF, 0xF, false); }

constexpr uint32_t NEVER2 = 0x80008000u;             // -32768 in both halves: below every value of the recurrence
constexpr uint32_t ONE2 = 0x00010001u;

struct Lane2 {
    s2 hPrev, ePrev, upIn, fIn, dgIn;                // as LaneState, both problems; dgIn carries the diagonal PLUS the match score
    uint32_t rIn;
    uint4 nb;                                        // the column lane 0 meets next: boundary row (x, y, z) and window characters (w)
    uint32_t initPrev;                               // CLIP: the initial row at the column this lane had one step ago
    s2 bS; uint32_t bT, bC; s2 j2;                   // BEST: maximum, its step, its count -- 16 bits each per problem; j2 = this lane's column in both halves
    uint32_t hHold, xBits, xHold;                    // table words in the making
};

struct Const2 {
    s2 go, ge, gi, mt, dM, negInf;                   // mt = match, dM = mismatch - match
    uint32_t rc, lane, W;                            // W = the longer window
    uint32_t clipMask, notBest;                      // per half: 0xFFFF where the row lies in the free left clip; 0x8000 where it cannot hold the maximum
    s2 jLo, Wown;                                    // per problem: first column that may hold the maximum, last column of its own window
    uint4 *col;                                      // LDS: boundary row + window characters per column
    const uint32_t *initRow;                         // LDS: H(j, 0) of both problems per column (entry 0 = 0: what the diagonal of column 1 sees)
    uint32_t *tabH[2], *tabX[2];
};

template <bool CLIP, bool BEST, bool RAMP>
__device__ __forceinline__ void dpStep2(const Const2 &k, Lane2 &s, uint32_t t)
{
    const s2 upS = asS2(shr1From(s.nb.x, bitsOf(s.upIn))), fS = asS2(shr1From(s.nb.y, bitsOf(s.fIn))), dgS = asS2(shr1From(s.nb.z, bitsOf(s.dgIn)));
    const uint32_t rS = shr1From(s.nb.w, s.rIn);
    const uint32_t jn = t + 1u < k.W ? t + 1u : k.W;                  // uniform
    s.nb = k.col[jn];
    const s2 prevR = s.hPrev;
    const s2 geE = k.ge + s.ePrev;
    s2 gapped = pmax(k.go + prevR, geE);
    const s2 eSt = pmax(gapped, k.negInf);
    s2 f = pmax(k.ge + fS, k.go + upS);
    gapped = pmax(f, gapped);
    const s2 diag = asS2(pkMadI16(nonZero01(rS ^ k.rc), bitsOf(k.dM), bitsOf(dgS)));      // the diagonal came down with the match score added
    const s2 up = pmax(gapped, diag);
    const s2 hSt = pmax(up, k.negInf);
    s2 dg = prevR;
    if (CLIP) {
        const int jc = (int)(t - k.lane);
        const uint32_t initJ = k.initRow[jc < 0 ? 0 : (jc > (int)k.W ? (int)k.W : jc)];
        f = pmax(f, asS2(pick(k.clipMask, bitsOf(asS2(initJ) + k.gi), NEVER2)));
        dg = pmax(dg, asS2(pick(k.clipMask, s.initPrev, NEVER2)));
        s.initPrev = initJ;
    }
    if (RAMP) { const bool started = t > k.lane; s.hPrev = started ? hSt : s.hPrev; s.ePrev = started ? eSt : s.ePrev; }
    else { s.hPrev = hSt; s.ePrev = eSt; }
    s.upIn = up; s.fIn = f; s.dgIn = dg + k.mt; s.rIn = rS;
    {
        // the bit kept is the complement of the one-problem kernel's ("H did NOT come out of an extended deletion"); the words are complemented when stored
        s.xBits = pkMadU16(s.xBits, 0x00020002u, nonZero01(bitsOf(hSt) ^ bitsOf(geE)));
        const uint32_t h2 = bitsOf(hSt);
        if (t & 1u) {
            __builtin_nontemporal_store((s.hHold & 0xFFFFu) | (h2 << 16), &k.tabH[0][(size_t)(t >> 1) * 64u + k.lane]);
            __builtin_nontemporal_store((s.hHold >> 16) | (h2 & 0xFFFF0000u), &k.tabH[1][(size_t)(t >> 1) * 64u + k.lane]);
        } else s.hHold = h2;
        if ((t & 15u) == 15u) {
            if (t & 16u) {
                __builtin_nontemporal_store(~((s.xHold << 16) | (s.xBits & 0xFFFFu)), &k.tabX[0][(size_t)(t >> 5) * 64u + k.lane]);
                __builtin_nontemporal_store(~((s.xHold & 0xFFFF0000u) | (s.xBits >> 16)), &k.tabX[1][(size_t)(t >> 5) * 64u + k.lane]);
            } else s.xHold = s.xBits;
        }
    }
    if (BEST) {
        // a cell counts where the row may end the alignment and jLo <= column <= the problem's own window; elsewhere its candidate is below everything
        const uint32_t outMask = negativeMask(bitsOf(s.j2 - k.jLo) | bitsOf(k.Wown - s.j2) | k.notBest);
        const s2 cand = asS2(pick(outMask, NEVER2, bitsOf(up)));
        const s2 m = pmax(cand, s.bS);
        const uint32_t gt = nonZeroMask(bitsOf(m) ^ bitsOf(s.bS));
        const uint32_t eq = ONE2 - nonZero01(bitsOf(cand) ^ bitsOf(s.bS));
        s.bC = pick(gt, ONE2, s.bC + eq);
        s.bT = pick(gt, t * ONE2, s.bT);
        s.bS = m;
        s.j2 = s.j2 + asS2(ONE2);
    }
    if (!RAMP) {
        // the chunk's last row becomes the boundary of the next chunk; lane 0 fetched this column 64 steps ago
        if (k.lane == 63u) { uint4 *c = &k.col[t - 63u]; c->x = bitsOf(up); c->y = bitsOf(f); c->z = bitsOf(s.dgIn); }
    }
}

struct Problem2 { uint32_t p[2], L[2], W[2], aL[2], aR[2]; int clipLt[2], clipRtLoc[2]; };

template <bool CLIP, bool BEST>
__device__ __forceinline__ void chunkSweep2(const DpArgs &a, const Problem2 &q, uint32_t c, uint32_t Wmax, uint4 *col, const uint32_t *initRow, Best best[2])
{
    Const2 k;
    const int go = a.gapOpen, ge = a.gapExtend, gi = go - ge;
    k.go = both(go, go); k.ge = both(ge, ge); k.gi = both(gi, gi); k.mt = both(a.match, a.match); k.dM = both(a.mismatch - a.match, a.mismatch - a.match);
    k.negInf = both(NEG_INF, NEG_INF);
    k.lane = threadIdx.x & 63u; k.W = Wmax; k.col = col; k.initRow = initRow;
    const uint32_t T = a.maxDPTableLength + 64u;
    const uint32_t i = (c << 6) + 1u + k.lane;                       // this lane's read row
    uint32_t rc[2], clipM[2], notB[2];
    int jLo[2], h0[2], e0[2];
    for (int h = 0; h < 2; h++) {
        uint32_t *table = a.table + (size_t)q.p[h] * a.tableWordsPerProblem + (size_t)c * dpTableChunkWords(T);
        k.tabH[h] = table; k.tabX[h] = table + dpTableXBase(T);
        const bool rowOk = i <= q.L[h];
        rc[h] = rowOk ? seqChar(a.reads, q.p[h], a.maxReadLength, i) : 0u;
        jLo[h] = max(1, (int)q.aR[h]);                               // first column that may hold the maximum (right anchor; no start offset in scheme 1)
        const bool rowBest = BEST && rowOk && (int)i >= q.clipRtLoc[h] && (int)q.W[h] - jLo[h] >= 0;
        notB[h] = rowBest ? 0u : 0x8000u;
        clipM[h] = CLIP && (int)i <= q.clipLt[h] ? 0xFFFFu : 0u;
        // column 0 of this row (:165-183)
        if ((int)i <= q.clipLt[h]) { h0[h] = clampS(go); e0[h] = clampS(go + gi); }
        else { const int up0 = gi + ge * ((int)i - (q.clipLt[h] > 0 ? q.clipLt[h] : 0)); h0[h] = clampS(up0); e0[h] = clampS(up0 + gi); }
    }
    k.rc = rc[0] | rc[1] << 16; k.clipMask = clipM[0] | clipM[1] << 16; k.notBest = notB[0] | notB[1] << 16;
    k.jLo = both(jLo[0], jLo[1]); k.Wown = both((int)q.W[0], (int)q.W[1]);
    Lane2 s;
    s.hPrev = both(h0[0], h0[1]); s.ePrev = both(e0[0], e0[1]);
    s.upIn = both(0, 0); s.fIn = both(0, 0); s.dgIn = both(0, 0); s.rIn = 0; s.initPrev = 0;
    s.bS = both(NEG_INF, NEG_INF); s.bT = 0; s.bC = 0;
    s.j2 = both(1 - (int)k.lane, 1 - (int)k.lane);                   // the column of step 1
    s.hHold = 0; s.xBits = 0; s.xHold = 0;
    s.nb = col[Wmax ? 1u : 0u];
    const uint32_t steps = Wmax + 63u;
    uint32_t t = 1;
    for (; t <= 63u; t++) dpStep2<CLIP, BEST, true>(k, s, t);
    for (; t + 1u <= steps; t += 2u) { dpStep2<CLIP, BEST, false>(k, s, t); dpStep2<CLIP, BEST, false>(k, s, t + 1u); }
    if (t <= steps) dpStep2<CLIP, BEST, false>(k, s, t);
    // what the last steps left in the registers (steps is the last step taken)
    for (int h = 0; h < 2; h++) {
        const uint32_t sh = h ? 16u : 0u;
        if (!(steps & 1u)) __builtin_nontemporal_store((s.hHold >> sh) & 0xFFFFu, &k.tabH[h][(size_t)(steps >> 1) * 64u + k.lane]);
        const uint32_t r = steps & 31u;
        if (r != 31u) {
            // the bits of steps (steps & ~31) .. steps: the first sixteen wait in xHold once there are more, the rest sit at the low end of xBits
            const uint32_t cur = (((s.xBits >> sh) & 0xFFFFu) << (15u - (r & 15u))) & 0xFFFFu, held = (s.xHold >> sh) & 0xFFFFu;
            const uint32_t w = r >= 16u ? (held << 16) | cur : cur << 16;
            __builtin_nontemporal_store(~w, &k.tabX[h][(size_t)(steps >> 5) * 64u + k.lane]);
        }
        if (BEST) {
            const uint32_t cnt = (s.bC >> sh) & 0xFFFFu, at = (s.bT >> sh) & 0xFFFFu;
            const int score = (int)(short)((bitsOf(s.bS) >> sh) & 0xFFFFu);
            if (cnt) bestMerge(best[h], score, ((at - k.lane) << 16) | i, cnt);
        }
    }
}

}  // namespace dp2

using namespace dp2;

__global__ void __launch_bounds__(256) dpFillPairKernel(DpArgs a)
{
    extern __shared__ __align__(16) unsigned char sh[];
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const uint32_t p0 = 2u * (blockIdx.x * (blockDim.x >> 6) + wv);
    if (p0 >= a.n) return;
    // an odd last problem sits in both halves: the same values go to the same places twice (a launch of its own for it cost a whole sweep of one wave,
    // 0.15-0.4 ms, on every call of the engines)
    const uint32_t twin = p0 + 1u >= a.n ? 0u : 1u;
    Problem2 q;
    for (int h = 0; h < 2; h++) {
        const uint32_t p = p0 + (uint32_t)h * twin;
        q.p[h] = p; q.L[h] = a.readLengths[p]; q.W[h] = a.dnaLengths[p];
        q.aL[h] = a.anchorLeft ? a.anchorLeft[p] : a.maxDNALength; q.aR[h] = a.anchorRight ? a.anchorRight[p] : 0u;
        q.clipLt[h] = (int)(a.clipLt ? a.clipLt[p] : 0u);
        q.clipRtLoc[h] = (int)(q.L[h] - (a.clipRt ? a.clipRt[p] : 0u));
    }
    uint4 *col = (uint4 *)(sh + (size_t)wv * a.ldsPerWavePair);
    uint32_t *initRow = (uint32_t *)(col + (a.maxDNALength + 1u));
    const uint32_t Wmax = q.W[0] > q.W[1] ? q.W[0] : q.W[1], Lmax = q.L[0] > q.L[1] ? q.L[0] : q.L[1];
    const int gi = a.gapOpen - a.gapExtend;
    // row 0 of both problems: H[j][0] = init(j), running open = init + gi, diagonal for row 1 = init(j-1) (+ match, see Lane2::dgIn), the window characters
    for (uint32_t j = lane; j <= Wmax; j += 64u) {
        uint32_t v[4] = {0, 0, 0, 0}, ini = 0;
        for (int h = 0; h < 2; h++) {
            const int init = (j >= q.aL[h]) ? NEG_INF : 0;
            const int prevInit = (j <= 1u) ? 0 : ((j - 1u >= q.aL[h]) ? NEG_INF : 0);
            const uint32_t ch = (j >= 1u && j <= q.W[h]) ? seqChar(a.dna, q.p[h], a.maxDNALength, j) : 0u;
            const uint32_t sh16 = h ? 16u : 0u;
            v[0] |= ((uint32_t)init & 0xFFFFu) << sh16; v[1] |= ((uint32_t)(init + gi) & 0xFFFFu) << sh16; v[2] |= ((uint32_t)(prevInit + a.match) & 0xFFFFu) << sh16;
            v[3] |= ch << sh16;
            ini |= ((uint32_t)init & 0xFFFFu) << sh16;
        }
        col[j] = make_uint4(v[0], v[1], v[2], v[3]);
        initRow[j] = j ? ini : 0u;
    }
    __builtin_amdgcn_wave_barrier();
    Best best[2];
    for (int h = 0; h < 2; h++) { best[h].score = NEG_INF; best[h].key = 0xFFFFFFFFu; best[h].count = 0; }
    const uint32_t chunks = (Lmax + 63u) >> 6;
    for (uint32_t c = 0; c < chunks; c++) {
        // wave-uniform: does the chunk touch the free left clip / hold rows that may end the alignment -- of either problem?
        const bool clip = q.clipLt[0] > (int)(c << 6) || q.clipLt[1] > (int)(c << 6);
        const bool mayEnd = (int)((c << 6) + 64u) >= q.clipRtLoc[0] || (int)((c << 6) + 64u) >= q.clipRtLoc[1];
        if (clip) {
            if (mayEnd) chunkSweep2<true, true>(a, q, c, Wmax, col, initRow, best);
            else chunkSweep2<true, false>(a, q, c, Wmax, col, initRow, best);
        } else {
            if (mayEnd) chunkSweep2<false, true>(a, q, c, Wmax, col, initRow, best);
            else chunkSweep2<false, false>(a, q, c, Wmax, col, initRow, best);
        }
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t hitPosOf[2], clipRtOf[2];
    for (int h = 0; h < 2; h++) {
        Best &b = best[h];
        for (int off = 32; off > 0; off >>= 1) {
            const int s2v = __shfl_xor(b.score, off);
            const uint32_t k2 = (uint32_t)__shfl_xor((int)b.key, off), c2 = (uint32_t)__shfl_xor((int)b.count, off);
            bestMerge(b, s2v, k2, c2);
        }
        const bool found = b.count && b.score > NEG_INF;                  // ties with the initial -32000 only count (:228-236)
        const uint32_t hitPos = found ? b.key >> 16 : 0u, row = found ? b.key & 0xFFFFu : 0u;
        hitPosOf[h] = hitPos;
        clipRtOf[h] = a.clipRt ? (found ? q.L[h] - row : 0u) : 0u;
        if (lane == 0) {
            const uint32_t p = q.p[h];
            a.startOffsets[p] = 0u;
            a.hitLocs[p] = hitPos;
            a.scores[p] = b.score;
            a.maxScoreCounts[p] = b.count;
            a.tableCols[p] = q.W[h];
            if (a.clipRt) a.clipRt[p] = clipRtOf[h];
        }
    }
    if (a.fuseTraceback) {
        // The traceback is a chain of dependent loads, one lane per problem: as a kernel of its own it took a quarter of the fill's time on the engines' small
        // launches, with the device nearly empty.  Here lanes 0 and 1 of the wave walk the two tables the wave has just written while the other waves of
        // the launch are still filling theirs: the walk's latency hides behind their arithmetic.  (The fence makes the wave's own table stores visible to
        // its loads; nobody else reads or writes these tables.)
        __threadfence();
        if (lane < 1u + twin)
            dpBacktrackOne(a, lane ? q.p[1] : q.p[0], lane ? best[1].score : best[0].score, lane ? hitPosOf[1] : hitPosOf[0], 0u, lane ? clipRtOf[1] : clipRtOf[0]);
    }
}


__global__ void dpBacktrackKernel(DpArgs a)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n) return;
    dpBacktrackOne(a, p, a.scores[p], a.hitLocs[p], a.startOffsets[p], a.clipRt ? a.clipRt[p] : 0u);
}

// may the launch's problems go two to a wavefront (dpFillPairKernel)?  Scheme 1, every value of the recurrence within 16 bits, the paired rows in LDS
static bool dpPairable(const DpArgs &a)
{
    if (!a.pairMode || a.scheme != 1) return false;
    if (a.match < 0 || a.match > 250 || a.mismatch > 0 || a.mismatch < -250 || a.gapOpen > 0 || a.gapOpen < -250 || a.gapExtend > 0 || a.gapExtend < -250) return false;
    if ((long long)a.maxReadLength * a.match > 30000) return false;
    if (-2 * a.gapOpen - a.gapExtend - a.mismatch > 700) return false;
    // (the window bound also bounds the 16-bit counters of the maximum tracking: a lane's tie count of one chunk grows by at most one per step, at most
    //  window + 63 steps, and the step number itself sits in a 16-bit half; the rows' counts are added up in 32 bits (bestMerge), so W x (clipRt + 1) ties --
    //  a homopolymer window -- pass 65,535 without harm: tests/test_gpu_dp.py::test_tie_counts_beyond_16_bits_in_the_paired_kernel)
    if (a.maxDNALength > 12000u || (size_t)a.ldsPerWavePair > 160u * 1024u) return false;
    return true;
}

// returns a hipError_t (hipErrorInvalidValue: a window that does not fit into LDS even with one problem per block)
int launchDp(const DpArgs &a0, hipStream_t s)
{
    if (!a0.n) return (int)hipSuccess;
    DpArgs a = a0;
    const size_t kLdsMax = 160u * 1024u;
    a.first = 0;
    if (dpPairable(a)) {
        uint32_t waves = 4;
        while (waves > 1 && (size_t)a.ldsPerWavePair * waves > kLdsMax) waves >>= 1;
        const size_t lds = (size_t)a.ldsPerWavePair * waves;
        if (lds > 48u * 1024u) {
            const hipError_t e = hipFuncSetAttribute((const void *)dpFillPairKernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        const uint32_t pairs = (a.n + 1u) >> 1;
        a.fuseTraceback = a.fuseMode;
        hipLaunchKernelGGL(dpFillPairKernel, dim3((pairs + waves - 1u) / waves), dim3(64u * waves), lds, s, a);
        if (a.fuseTraceback) return (int)hipSuccess;
        a.first = a.n;
    }
    if (a.first < a.n) {
        // a wave keeps its window and the boundary row of its pass in LDS (ldsPerWave bytes); blocks hold 4 waves while that fits into the CU's 160 KB,
        // then 2, then 1 (windows up to about 12,000 columns)
        uint32_t waves = 4;
        while (waves > 1 && (size_t)a.ldsPerWave * waves > kLdsMax) waves >>= 1;
        const size_t lds = (size_t)a.ldsPerWave * waves;
        if (lds > kLdsMax) return (int)hipErrorInvalidValue;
        if (lds > 48u * 1024u) {
            // above the default limit for dynamic LDS: without the attribute the launch fails (and only shows at the next synchronisation)
            const hipError_t e = hipFuncSetAttribute((const void *)dpFillKernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(dpFillKernel, dim3((a.n - a.first + waves - 1u) / waves), dim3(64u * waves), lds, s, a);
    }
    a.first = 0;
    hipLaunchKernelGGL(dpBacktrackKernel, dim3((a.n + 63u) / 64u), dim3(64), 0, s, a);
    return (int)hipSuccess;
}

}  // namespace s3

// one kernel of this file's code object, asked for its attributes: the runtime loads a code object at the first use of one of its kernels (preloadKernels, s3dp_kernels.hip)
void s3::preloadDp() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)s3::dpBacktrackKernel); (void)hipGetLastError(); }
