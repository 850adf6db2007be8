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
    const uint32_t p = blockIdx.x * (blockDim.x >> 6) + wv;                // 1, 2 or 4 problems per block: what the windows leave room for in LDS
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

__global__ void dpBacktrackKernel(DpArgs a)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n) return;
    unsigned char *pat = a.pattern + (size_t)p * a.patternLen;
    pat[0] = 0;
    if (a.scores[p] < a.cutoffs[p]) return;
    const uint32_t L = a.readLengths[p];
    const tb::View v{a, p, a.maxDPTableLength + 64u, a.startOffsets[p], a.anchorLeft ? a.anchorLeft[p] : a.maxDNALength, a.clipLt ? a.clipLt[p] : 0u,
                     a.table + (size_t)p * a.tableWordsPerProblem, a.gapOpen, a.gapExtend, a.gapOpen - a.gapExtend};
    const uint32_t clipRt = a.clipRt ? a.clipRt[p] : 0u;
    uint32_t out = 0;
    auto put = [&](uint32_t x) { if (out < a.patternLen) pat[out] = (unsigned char)x; ++out; };
    auto run = [&](char op, uint32_t n) { put((uint32_t)op); put('V'); put(n); };       // 'V'<n>: n - 1 more of the op before (decoded by the engines)
    uint32_t i = L - clipRt, j = a.hitLocs[p];                                           // read position, window column (1-based, 0 = before the first)
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

// returns a hipError_t (hipErrorInvalidValue: a window that does not fit into LDS even with one problem per block)
int launchDp(const DpArgs &a, hipStream_t s)
{
    if (!a.n) return (int)hipSuccess;
    // a wave keeps its window and the boundary row of its pass in LDS (ldsPerWave bytes); blocks hold 4 waves while that fits into the CU's 160 KB,
    // then 2, then 1 (windows up to about 12,000 columns)
    const size_t kLdsMax = 160u * 1024u;
    uint32_t waves = 4;
    while (waves > 1 && (size_t)a.ldsPerWave * waves > kLdsMax) waves >>= 1;
    const size_t lds = (size_t)a.ldsPerWave * waves;
    if (lds > kLdsMax) return (int)hipErrorInvalidValue;
    if (lds > 48u * 1024u) {
        // above the default limit for dynamic LDS: without the attribute the launch fails (and only shows at the next synchronisation)
        const hipError_t e = hipFuncSetAttribute((const void *)dpFillKernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(dpFillKernel, dim3((a.n + waves - 1u) / waves), dim3(64u * waves), lds, s, a);
    hipLaunchKernelGGL(dpBacktrackKernel, dim3((a.n + 63u) / 64u), dim3(64), 0, s, a);
    return (int)hipSuccess;
}

}  // namespace s3
