// s3dp_flank.hip -- the FLANK TABLES of check-and-extend (round 6), built on the device at index upload.
//
// Why.  ceKernel (s3dp_ce.hip; the CPU path: BGS-HostAlgnmtAlgo2.cpp:36-176, 2bwt-flex/SRA2BWTCheckAndExtend.c:142) verifies 2.1 G candidates per ladder of the
// benchmark, 97 % of them rejected, and each cost a suffix-array line and then, dependently, a text line: 64.6 ms per ladder at the rate of random 64-byte
// lines, for work SURVEY 8d does not count at all.  The counted kernel says where the rejected ones are decided (s3_kernel_stats.ceRejected, bench.py
// roofline.per_kernel.ce.rejected): 91.7 % inside the 32 characters NEXT to the matched part, in the very step the candidate was handed over in.  So every
// suffix-array entry gets a 16-byte record {text position, the 48 characters next to it, the adjacent one first}: a rejected candidate costs that record -- a
// quarter of a line that its task's other candidates share -- and only survivors go on to the text.
//
// Two tables, because "next to the matched part" has two sides:
//   flankBack[k]  (order of the FORWARD index)   {S = SA[k], T[S-1], T[S-2], ... T[S-48]}: a task written in a BACKWARD step matched the read from S on and
//                 compares leftwards; S is the same whatever the matched length.
//   flankFwd[k']  (order of the REVERSE index)   {E, T[E], T[E+1], ... T[E+47]}: a task written in a FORWARD step compares rightwards from the END of its matched
//                 part, whose distance from S differs from task to task -- but the reverse index (the BWT of the reversed text) orders the same occurrences by
//                 their ENDS: entry k' of the reverse range stands for the occurrence that ends right before E, whatever its length.  The stepping kernels
//                 hold both ranges in a forward step (SearchOut::recAux carries the reverse one to ceKernel).
// The 2BWT files hold no suffix array of the reverse index.  It is not needed as an array: from the reverse rank x of the text prefix T[0..p] the rank of
// T[0..p+1] is one LF step with the character T[p+1] -- a walk along the text, sequential but for its rank line.  One walk of n steps would take an hour; so
// the text is cut at ANCHORS every S3_FLANK_ANCHOR positions whose reverse rank is found directly: a bidirectional backward search of T[..p] from p leftwards
// on the FORWARD index until its range is one suffix wide -- the reverse range it maintains is then the reverse rank of p (an anchor inside a repeat longer than
// S3_FLANK_ANCHOR_MAX characters is skipped: its left neighbour walks on through it).  A lane per anchor walks to the next valid anchor, writes flankFwd[x] at
// every position and CHECKS that it arrives at the rank the next anchor found for itself; the positions written are counted and must be n.  3.1 G positions:
// half a second.  A table that fails either check is dropped (the library then verifies those tasks the old way) and the upload says so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <chrono>
#include "s3dp_device.h"
#include "s3dp_search_common.h"

namespace s3 {

using namespace v2;

#define S3_FLANK_ANCHOR 1024u              // text positions between two anchors of the reverse-rank walk
#define S3_FLANK_ANCHOR_MAX 2048u          // characters an anchor's search may take to become unique

// all four ranks at index i of one half of the index, by ONE lane (the search kernels' own arithmetic: rankSel / rankFromWords)
__device__ __forceinline__ uint4 rank4Lane(const DevBwt &b, uint32_t i)
{
    i -= (i > b.inverseSa0);
    const RankSel q = rankSel(i);
    const uint4 *line = b.lines + (size_t)((i + 63u) >> 7) * 4u;
    const uint4 occ = line[0], half = line[q.fwd ? 2 : 1];
    const uint32_t sub = ((const uint32_t *)(line + 3))[q.fwd ? 1 : 0];
    return rankFromWords(q, occ, q.far ? half.z : half.x, q.far ? half.w : half.y, sub);
}

// 16 characters of the text from position p (MSB first); positions before the text's start read as 'A' (the text's end: its zero tail, s3dp_api.hip)
__device__ __forceinline__ uint32_t textWindow(const uint32_t *pac, long long p)
{
    if (p <= -16) return 0u;
    if (p < 0) return pac[0] >> (2u * (uint32_t)(-p));
    const uint32_t w = (uint32_t)((uint64_t)p >> 4), sh = ((uint32_t)p & 15u) * 2u;
    const uint32_t a = pac[w], b = pac[w + 1u];
    return sh ? (a << sh) | (b >> (32u - sh)) : a;
}
__device__ __forceinline__ uint32_t textChar(const uint32_t *pac, uint32_t p) { return (pac[p >> 4] >> (30u - 2u * (p & 15u))) & 3u; }

// flankBack[k] for every suffix-array entry
__global__ void __launch_bounds__(256) flankBackKernel(DevIndex ix, uint4 *out)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= ix.n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t S = ix.sa[k];
        if (S > ix.n) { out[k] = make_uint4(S, 0u, 0u, 0u); continue; }       // (entry 0, the empty suffix: the file format stores 0xFFFFFFFF there, BWT.c:286)
        const long long s = (long long)S;
        // characters S-1, S-2, ...: the windows that END at S-1, S-17, S-33, each reversed
        out[k] = make_uint4(S, pairRev(textWindow(ix.pac, s - 16)), pairRev(textWindow(ix.pac, s - 32)), pairRev(textWindow(ix.pac, s - 48)));
    }
}

// reverse rank of anchor a (text position a * S3_FLANK_ANCHOR), or 0xFFFFFFFF: the bidirectional backward search of T[..p] on the forward index, leftwards from
// p, until one suffix is left; the reverse range's end is then the rank, in the reverse index, of the occurrence that ENDS at p
__global__ void __launch_bounds__(256) flankAnchorKernel(DevIndex ix, uint32_t nAnchors, uint32_t *anchorRank)
{
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nAnchors) return;
    const uint32_t p = a * S3_FLANK_ANCHOR;
    if (a == 0u) {
        // the prefix of one character: the smallest suffix of the reversed text that starts with T[0]
        const uint4 o = rank4Lane(ix.rev, 0u);
        anchorRank[0] = sel4(o, textChar(ix.pac, 0u)) + 1u;
        return;
    }
    uint32_t l = 0u, r = ix.n, pr = ix.n, found = 0xFFFFFFFFu;
    for (uint32_t t = 0; t <= p && t < S3_FLANK_ANCHOR_MAX; t++) {
        const uint32_t c = textChar(ix.pac, p - t);
        const uint4 oL = rank4Lane(ix.fwd, l), oR = rank4Lane(ix.fwd, r + 1u);
        const uint32_t e1 = oR.y - oL.y, e2 = oR.z - oL.z, e3 = oR.w - oL.w;
        pr -= c == 0u ? e1 + e2 + e3 : c == 1u ? e2 + e3 : c == 2u ? e3 : 0u;      // the passive range shrinks by the occurrences of the larger characters
        l = sel4(oL, c) + 1u; r = sel4(oR, c);
        if (l > r) break;                                              // (cannot happen: the text holds its own substring)
        if (l == r) { found = pr; break; }
    }
    anchorRank[a] = found;
}

// the walk: lane a starts at its anchor (if it found a rank) and goes on to the next anchor that did
__global__ void __launch_bounds__(256) flankWalkKernel(DevIndex ix, uint32_t nAnchors, const uint32_t *anchorRank, uint4 *out, unsigned long long *written, uint32_t *bad)
{
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nAnchors) return;
    uint32_t x = anchorRank[a];
    if (x == 0xFFFFFFFFu) return;
    uint32_t p = a * S3_FLANK_ANCHOR, nextA = a + 1u;
    unsigned long long count = 0;
    for (;;) {
        // x = reverse rank of the occurrence that ends at p: what follows it in the text
        const long long e = (long long)p + 1;
        out[x] = make_uint4(p + 1u, textWindow(ix.pac, e), textWindow(ix.pac, e + 16), textWindow(ix.pac, e + 32));
        count++;
        if (p + 1u >= ix.n) { if (x != ix.rev.inverseSa0) atomicOr(bad, 2u); break; }      // the whole text: the suffix of the reversed text that has no predecessor
        p++;
        x = sel4(rank4Lane(ix.rev, x), textChar(ix.pac, p)) + 1u;
        if (nextA < nAnchors && p == nextA * S3_FLANK_ANCHOR) {
            const uint32_t want = anchorRank[nextA];
            if (want != 0xFFFFFFFFu) { if (want != x) atomicOr(bad, 1u); break; }
            nextA++;
        }
    }
    atomicAdd(written, count);
}

// Builds the tables the device has room for (`which`: 1 = flankBack only, 2 = both) and hangs them into d; reserveBytes of device memory stay free.  Returns a
// hipError_t as int (out of memory is NOT an error: the table is left out); *allocs receives what the index has to free.
int launchFlankTables(DevIndex &d, int which, size_t reserveBytes, void **allocs, int *nAllocs, size_t *bytes, bool verbose, hipStream_t s)
{
    *nAllocs = 0; *bytes = 0;
    d.flankBack = nullptr; d.flankFwd = nullptr;
    if (which < 1 || d.n < 64u) return (int)hipSuccess;
    const size_t tableBytes = ((size_t)d.n + 1u) * sizeof(uint4);
    auto room = [&]() -> bool {
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return false;
        return freeB >= tableBytes + reserveBytes;
    };
    hipError_t e;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!verbose) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[s3dp] flank tables: %-44s %8.3f s\n", what, std::chrono::duration<double>(now - t0).count());
        t0 = now;
    };
    if (room()) {
        uint4 *tb = nullptr;
        e = hipMalloc(&tb, tableBytes);
        lap("allocation (forward index order)");
        if (e == hipSuccess) {
            hipLaunchKernelGGL(flankBackKernel, dim3(8192), dim3(256), 0, s, d, tb);
            if ((e = hipStreamSynchronize(s)) != hipSuccess || (e = hipGetLastError()) != hipSuccess) { (void)hipFree(tb); return (int)e; }
            lap("forward index order: text windows");
            d.flankBack = tb; allocs[(*nAllocs)++] = tb; *bytes += tableBytes;
        } else (void)hipGetLastError();
    }
    if (which < 2 || !d.flankBack || !room()) return (int)hipSuccess;
    uint4 *tf = nullptr;
    uint32_t *anchorRank = nullptr;
    unsigned long long *written = nullptr;
    const uint32_t nAnchors = (d.n + S3_FLANK_ANCHOR - 1u) / S3_FLANK_ANCHOR;
    if (hipMalloc(&tf, tableBytes) != hipSuccess) { (void)hipGetLastError(); return (int)hipSuccess; }
    const size_t tail = ((size_t)nAnchors + 1u) & ~(size_t)1;           // (the two 64-bit words behind the ranks, aligned)
    e = hipMalloc(&anchorRank, tail * 4u + 16u);
    if (e == hipSuccess) e = hipMemsetAsync(tf, 0, tableBytes, s);
    if (e == hipSuccess) e = hipMemsetAsync(anchorRank + tail, 0, 16u, s);
    if (e != hipSuccess) { (void)hipFree(tf); if (anchorRank) (void)hipFree(anchorRank); (void)hipGetLastError(); return e == hipErrorOutOfMemory ? (int)hipSuccess : (int)e; }
    written = (unsigned long long *)(anchorRank + tail);               // [0]: positions written; [1]: the checks' flags
    uint32_t *bad = (uint32_t *)(written + 1);
    lap("allocation + zeroes (reverse index order)");
    hipLaunchKernelGGL(flankAnchorKernel, dim3((nAnchors + 255u) / 256u), dim3(256), 0, s, d, nAnchors, anchorRank);
    lap("reverse index order: anchors");
    hipLaunchKernelGGL(flankWalkKernel, dim3((nAnchors + 255u) / 256u), dim3(256), 0, s, d, nAnchors, anchorRank, tf, written, bad);
    unsigned long long res[2] = {0, 0};
    e = hipMemcpyAsync(res, written, 16, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    lap("reverse index order: the walk");
    (void)hipFree(anchorRank);
    if (e != hipSuccess) { (void)hipFree(tf); return (int)e; }
    const uint32_t flags = (uint32_t)res[1];
    if (res[0] != (unsigned long long)d.n || flags) {
        // never seen; if it ever happens the forward steps' tasks are verified the old way and the upload says why
        fprintf(stderr, "[s3dp] flank table of the reverse index dropped: %llu of %u positions written, check flags %u\n", res[0], d.n, flags);
        (void)hipFree(tf);
        return (int)hipSuccess;
    }
    if (verbose) fprintf(stderr, "[s3dp] flank tables: %u anchors, %llu positions walked\n", nAnchors, res[0]);
    d.flankFwd = tf; allocs[(*nAllocs)++] = tf; *bytes += tableBytes;
    return (int)hipSuccess;
}

}  // namespace s3

// one kernel of this file's code object, asked for its attributes: the runtime loads a code object at the first use of one of its kernels (preloadKernels, s3dp_kernels.hip)
void s3::preloadFlank() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)s3::flankBackKernel); (void)hipGetLastError(); }
