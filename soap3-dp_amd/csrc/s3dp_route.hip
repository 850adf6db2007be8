// s3dp_route.hip -- routing of the pairs the BWT ladder left without a proper pairing, on the device (round 4).
//
// Replaces the needProceedDP branch of the reference's hostKernel (CPUfunctions.cpp:2153-2262, 2444-2543): per unpaired pair the hit lists of its two ends
// are cut by the stage's hit limit (retainAllBest / retainAllBestAndSecBest, SAList.cpp:140,290: a list longer than the limit keeps its best, in the MAPQ
// modes also its second-best, stratum) and handed on --
//   no hit on either end                         -> deep DP (the pair is listed)
//   hits on one end (half-aligned)               -> that end's list, SA ranges before check-and-extend hits (HalfEndOccStream::fetchNextOcc,
//                                                   DV-DPfunctions.cu:1901-1960), to the default DP, or, still above the limit, to the new default DP
//   hits on both ends, both within the limit     -> both lists (position-sorted, as the pairing attempt left them) to the default DP
//   hits on both ends, otherwise                 -> the SHORTER list (the second end's when equal) to the default / new default DP, the other one is only
//                                                   kept for the unpaired output (otherSoap3Result)
// Round 3 did this on the host: every occurrence of every unpaired pair came over PCIe (12-15 M per chunk of 512 Ki pairs on a genome that is half
// repeats: a pair with an end inside a repeat family brings thousands), was copied four times by the routing loop and -- the new default DP's input, 11 M
// entries per chunk -- went back to the device: 0.19 s of routing, 0.15 s of fetching and 0.11 s of re-uploading per chunk.  Here the lists are built where
// the ladder left the occurrences; what comes to the host is one byte per pair, the default DP's list (a few hundred thousand entries) and, for the few
// pairs every stage fails on, their kept lists.
#include <hipcub/hipcub.hpp>
#include "s3dp_internal.h"

using namespace s3;
using namespace s3i;

namespace {

struct RouteArgs {
    uint32_t        n;                       // reads of the batch (2 per pair)
    const DevPair  *pairs;
    const uint32_t *offs, *pos, *meta;       // read-ordered occurrence lists of the ladder (dOutOffsets / dOutPos / dOutMeta: strand | mismatches << 8 | fromSA << 16)
    uint32_t        maxHit[2], sec;          // hit limit per end; sec = 1: the second-best stratum survives the cut (MAPQ modes)
    uint32_t       *kept;                    // per read: entries after the cut
    uint8_t        *thr;                     // per read: mismatch threshold of the cut (255: no cut)
    uint8_t        *route;                   // per read: list id (0 none, 1 default DP, 2 new default DP, 3 other) | 4 = SA ranges first
    uint8_t        *pairRoute;               // per pair: S3_ROUTE_* bits
    uint32_t       *counts;                  // [3][n] + 1: entries of read r in list l at counts[(l - 1) * n + r]
    const uint32_t *listOffs;                // exclusive scan of counts
    uint32_t       *out;                     // the three lists behind each other, s3_occ = 3 dwords
};

__device__ __forceinline__ uint32_t waveMin(uint32_t v) { for (int o = 32; o; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o)); return v; }
__device__ __forceinline__ uint32_t waveSum(uint32_t v) { for (int o = 32; o; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o); return v; }

// per read: how many entries survive the cut.  A list within its limit is kept whole (a lane's own business); a longer one is read twice by the whole wave
// (minimum, then count) -- the ends inside repeat families hold up to MaxHitsEachEndForPairing entries each.
__global__ void routeCountKernel(RouteArgs a)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    uint32_t cnt = 0, beg = 0;
    bool on = false;
    if (r < a.n) {
        on = a.pairs[r >> 1].phase == 0;
        beg = a.offs[r];
        cnt = on ? a.offs[r + 1] - beg : 0u;
        a.kept[r] = cnt;
        a.thr[r] = 255;
    }
    unsigned long long big = __ballot(cnt > a.maxHit[r & 1u]);
    while (big) {
        const int l = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t c = (uint32_t)__shfl((int)cnt, l), b0 = (uint32_t)__shfl((int)beg, l);
        uint32_t mn = 255;
        for (uint32_t k = lane; k < c; k += 64u) mn = min(mn, (a.meta[b0 + k] >> 8) & 0xFFu);
        mn = waveMin(mn);
        const uint32_t t = mn + a.sec;
        uint32_t m = 0;
        for (uint32_t k = lane; k < c; k += 64u) m += ((a.meta[b0 + k] >> 8) & 0xFFu) <= t;
        m = waveSum(m);
        if ((int)lane == l) { a.kept[r] = m; a.thr[r] = (uint8_t)t; }
    }
}

// per pair: where the two lists go
__global__ void routeDecideKernel(RouteArgs a)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * p + 1 >= a.n) return;
    const uint32_t r1 = 2 * p, r2 = r1 + 1;
    uint8_t ro[2] = {0, 0}, pr = 0;
    if (a.pairs[p].phase == 0) {
        const uint32_t n1 = a.offs[r1 + 1] - a.offs[r1], n2 = a.offs[r2 + 1] - a.offs[r2];
        const uint32_t k1 = a.kept[r1], k2 = a.kept[r2];
        pr = S3_ROUTE_UNPAIRED;
        if (n1 == 0 && n2 == 0) pr |= S3_ROUTE_DEEP;
        else if (n1 == 0 || n2 == 0) {
            const int e = n1 ? 0 : 1;
            const uint32_t k = e ? k2 : k1;
            ro[e] = (uint8_t)((k <= a.maxHit[e] ? 1 : 2) | 4);
        } else if (k1 <= a.maxHit[0] && k2 <= a.maxHit[1]) { ro[0] = 1; ro[1] = 1; }
        else if (k1 < k2) { ro[0] = k1 <= a.maxHit[0] ? 1 : 2; ro[1] = 3; }
        else { ro[1] = k2 <= a.maxHit[1] ? 1 : 2; ro[0] = 3; }
        for (int e = 0; e < 2; e++) {
            if ((ro[e] & 3) == 1) pr |= S3_ROUTE_DEFAULT;
            if ((ro[e] & 3) == 2) pr |= (uint8_t)(S3_ROUTE_NEW_DEFAULT | (e ? S3_ROUTE_GIVEN_END_2 : 0));
        }
    }
    a.pairRoute[p] = pr;
    for (int e = 0; e < 2; e++) {
        const uint32_t r = r1 + e, l = ro[e] & 3u, k = a.kept[r];
        a.route[r] = ro[e];
        a.counts[r] = l == 1 ? k : 0u;
        a.counts[(size_t)a.n + r] = l == 2 ? k : 0u;
        a.counts[2 * (size_t)a.n + r] = l == 3 ? k : 0u;
    }
    if (p == 0) a.counts[3 * (size_t)a.n] = 0;
}

// per read with a list: write it -- in the order the reference hands it on, cut at the threshold.  Lists of up to 16 entries by their own lane, longer ones
// by the whole wave, 64 entries per step with an ordered compaction (ballot + prefix count).
__global__ void routeWriteKernel(RouteArgs a)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    uint32_t cnt = 0, beg = 0, ro = 0, thr = 255, dst = 0;
    if (r < a.n) {
        ro = a.route[r];
        if (ro & 3u) {
            beg = a.offs[r]; cnt = a.offs[r + 1] - beg; thr = a.thr[r];
            dst = a.listOffs[(size_t)((ro & 3u) - 1u) * a.n + r];
        }
    }
    auto put = [&](uint32_t at, uint32_t k) {
        const uint32_t m = a.meta[k];
        uint32_t *o = a.out + (size_t)at * 3;
        o[0] = r; o[1] = a.pos[k]; o[2] = (m & 0xFFu) | (m & 0xFF00u);
    };
    if (cnt && cnt <= 16u) {
        uint32_t w = dst;
        for (int pass = (ro & 4u) ? 0 : 1; pass < 2; pass++)
            for (uint32_t k = beg; k < beg + cnt; k++) {
                const uint32_t m = a.meta[k];
                if ((ro & 4u) && (((m >> 16) & 1u) != 0u) != (pass == 0)) continue;
                if (((m >> 8) & 0xFFu) > thr) continue;
                put(w++, k);
            }
    }
    unsigned long long big = __ballot(cnt > 16u);
    while (big) {
        const int l = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t c = (uint32_t)__shfl((int)cnt, l), b0 = (uint32_t)__shfl((int)beg, l), t = (uint32_t)__shfl((int)thr, l);
        const uint32_t rr = (uint32_t)__shfl((int)r, l), roL = (uint32_t)__shfl((int)ro, l);
        uint32_t w = (uint32_t)__shfl((int)dst, l);
        for (int pass = (roL & 4u) ? 0 : 1; pass < 2; pass++)
            for (uint32_t k0 = 0; k0 < c; k0 += 64u) {
                const uint32_t k = k0 + lane;
                uint32_t m = 0;
                bool keep = k < c;
                if (keep) {
                    m = a.meta[b0 + k];
                    if ((roL & 4u) && (((m >> 16) & 1u) != 0u) != (pass == 0)) keep = false;
                    if (((m >> 8) & 0xFFu) > t) keep = false;
                }
                const unsigned long long mask = __ballot(keep);
                if (keep) {
                    const uint32_t at = w + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                    uint32_t *o = a.out + (size_t)at * 3;
                    o[0] = rr; o[1] = a.pos[b0 + k]; o[2] = (m & 0xFFu) | (m & 0xFF00u);
                }
                w += (uint32_t)__popcll(mask);
            }
    }
}

// kept lists of selected reads, packed behind each other (the leftovers' unpaired output)
__global__ void routeGatherKernel(uint32_t nSel, const uint32_t *src, const uint32_t *dstOffs, const uint32_t *routeOcc, uint32_t *out)
{
    const uint32_t i = blockIdx.x, lane = threadIdx.x;                         // one wave per selected read
    if (i >= nSel) return;
    const uint32_t c = dstOffs[i + 1] - dstOffs[i], s = src[i], d = dstOffs[i];
    for (uint32_t k = lane; k < c * 3u; k += 64u) out[(size_t)d * 3 + k] = routeOcc[(size_t)s * 3 + k];
}

}  // namespace

static_assert(sizeof(s3_occ) == 12, "s3_occ is three dwords");

extern "C" int s3_batch_route_pairs(s3_batch *b, const uint32_t maxHitNum[2], int keepSecondBest, s3_pair_routing *out)
{
    if (!b || !maxHitNum || !out) return fail(S3_EINVAL, "s3_batch_route_pairs: null argument");
    if (b->lastKind != 3) return fail(S3_EINVAL, "s3_batch_route_pairs: no paired-end result pending");
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    HIPCHK(hipSetDevice(b->ix->dev));
    const uint32_t n = b->n, np = n / 2;
    memset(out, 0, sizeof *out);
    b->routed = false;
    b->hPairRoute.assign(np, 0);
    b->hRouteOffs.assign(3 * (size_t)n + 1, 0);
    if (!np) { b->routed = true; return S3_OK; }
    int rc;
    if (!b->dRouteTemp) {                                                    // (the LAST buffer of the chain: a chain that failed half-way is tried again, not trusted)
        const size_t m = b->maxReads + 2;
        if ((!b->dRouteKept && (rc = balloc(b, &b->dRouteKept, m))) || (!b->dRouteThr && (rc = balloc(b, &b->dRouteThr, m))) || (!b->dRoute && (rc = balloc(b, &b->dRoute, m))) ||
            (!b->dPairRoute && (rc = balloc(b, &b->dPairRoute, m / 2 + 1))) || (!b->dRouteCounts && (rc = balloc(b, &b->dRouteCounts, 3 * m + 1))) ||
            (!b->dRouteOffs && (rc = balloc(b, &b->dRouteOffs, 3 * m + 1)))) return rc;
        size_t tb = 0;
        HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, b->dRouteCounts, b->dRouteOffs, (int)(3 * m + 1), b->stream));
        if ((rc = balloc(b, &b->dRouteTemp, tb + 16))) return rc;
        b->routeTempBytes = tb + 16;
    }
    RouteArgs a{};
    a.n = n; a.pairs = b->dPairs; a.offs = b->dOutOffsets; a.pos = b->dOutPos; a.meta = b->dOutMeta;
    a.maxHit[0] = maxHitNum[0]; a.maxHit[1] = maxHitNum[1]; a.sec = keepSecondBest ? 1u : 0u;
    a.kept = b->dRouteKept; a.thr = b->dRouteThr; a.route = b->dRoute; a.pairRoute = b->dPairRoute; a.counts = b->dRouteCounts; a.listOffs = b->dRouteOffs;
    hipLaunchKernelGGL(routeCountKernel, dim3((n + 255) / 256), dim3(256), 0, b->stream, a);
    hipLaunchKernelGGL(routeDecideKernel, dim3((np + 255) / 256), dim3(256), 0, b->stream, a);
    size_t tb = b->routeTempBytes;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->dRouteTemp, tb, b->dRouteCounts, b->dRouteOffs, (int)(3 * (size_t)n + 1), b->stream));
    HIPCHK(hipMemcpyAsync(b->hRouteOffs.data(), b->dRouteOffs, (3 * (size_t)n + 1) * 4, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(b->hPairRoute.data(), b->dPairRoute, np, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    // (the lists hold at most the ladder's occurrences, whose total is below 2^32: the scan cannot wrap)
    const uint32_t total = b->hRouteOffs[3 * (size_t)n];
    if (total > b->routeCap) {
        const size_t cap = (size_t)total + total / 4 + 4096;
        if (b->routeCap == 0) { if ((rc = balloc(b, &b->dRouteOcc, cap * 3))) return rc; }
        else if ((rc = bgrow(b, &b->dRouteOcc, cap * 3, 0))) return rc;
        b->routeCap = cap;
    }
    if (total) {
        a.out = b->dRouteOcc;
        hipLaunchKernelGGL(routeWriteKernel, dim3((n + 255) / 256), dim3(256), 0, b->stream, a);
    }
    HIPCHK(hipGetLastError());
    out->nDefault = b->hRouteOffs[n] - b->hRouteOffs[0];
    out->nNewDefault = b->hRouteOffs[2 * (size_t)n] - b->hRouteOffs[n];
    out->nOther = total - b->hRouteOffs[2 * (size_t)n];
    for (uint32_t p = 0; p < np; p++) {
        const uint8_t x = b->hPairRoute[p];
        if (!x) continue;
        out->nUnpaired++;
        if (x & S3_ROUTE_DEEP) out->nDeepPairs++;
        if (x & S3_ROUTE_DEFAULT) out->nDefaultPairs++;
        if (x & S3_ROUTE_NEW_DEFAULT) out->nNewDefaultPairs++;
    }
    b->routing = *out;
    b->routed = true;
    return S3_OK;
}

extern "C" int s3_batch_fetch_routing(s3_batch *b, uint8_t *pairRoute, s3_occ *defaultOcc, size_t defaultCap, s3_occ *newDefaultOcc, size_t newDefaultCap,
                                      s3_occ *otherOcc, size_t otherCap)
{
    if (!b || !b->routed) return fail(S3_EINVAL, "s3_batch_fetch_routing: call s3_batch_route_pairs first");
    HIPCHK(hipSetDevice(b->ix->dev));
    const uint32_t n = b->n;
    if (pairRoute && n / 2) memcpy(pairRoute, b->hPairRoute.data(), n / 2);
    struct { s3_occ *dst; size_t cap; uint64_t cnt; size_t at; } L[3] = {
        {defaultOcc, defaultCap, b->routing.nDefault, n ? b->hRouteOffs[0] : 0u},
        {newDefaultOcc, newDefaultCap, b->routing.nNewDefault, n ? b->hRouteOffs[n] : 0u},
        {otherOcc, otherCap, b->routing.nOther, n ? b->hRouteOffs[2 * (size_t)n] : 0u}};
    for (auto &l : L) {
        if (!l.dst || !l.cnt) continue;
        if (l.cnt > l.cap) return fail(S3_EOVERFLOW, "s3_batch_fetch_routing: output array too small");
        HIPCHK(hipMemcpyAsync(l.dst, b->dRouteOcc + l.at * 3, l.cnt * sizeof(s3_occ), hipMemcpyDeviceToHost, b->stream));
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    return S3_OK;
}

extern "C" int s3_batch_fetch_kept(s3_batch *b, const uint32_t *reads, size_t nReads, uint64_t *offsets, s3_occ **occOut, uint64_t *nOccOut)
{
    if (!b || !b->routed || (!reads && nReads) || !offsets || !occOut) return fail(S3_EINVAL, "s3_batch_fetch_kept: bad argument (or no routing pending)");
    HIPCHK(hipSetDevice(b->ix->dev));
    const uint32_t n = b->n;
    *occOut = nullptr;
    std::vector<uint32_t> src(nReads ? nReads : 1), dst(nReads + 1, 0);
    // a read's kept list sits in exactly one of the three lists; its length is the difference to the next read's offset in that list
    uint64_t total = 0;
    for (size_t i = 0; i < nReads; i++) {
        const uint32_t r = reads[i];
        if (r >= n) return fail(S3_EINVAL, "s3_batch_fetch_kept: read index out of range");
        uint32_t c = 0, s = 0;
        for (size_t l = 0; l < 3; l++) {
            const uint32_t cl = b->hRouteOffs[l * n + r + 1] - b->hRouteOffs[l * n + r];
            if (cl) { c = cl; s = b->hRouteOffs[l * n + r]; }
        }
        src[i] = s; dst[i] = (uint32_t)total;
        offsets[i] = total;
        total += c;
    }
    dst[nReads] = (uint32_t)total;
    offsets[nReads] = total;
    if (nOccOut) *nOccOut = total;
    s3_occ *res = (s3_occ *)malloc((size_t)total * sizeof(s3_occ) + 16);
    if (!res) return fail(S3_ENOMEM, "s3_batch_fetch_kept: out of host memory");
    if (total) {
        uint32_t *dSrc = nullptr, *dDst = nullptr, *dOut = nullptr;
        hipError_t e = scratchAlloc((void **)&dSrc, nReads * 4 + 16, b->stream);
        if (e == hipSuccess) e = scratchAlloc((void **)&dDst, (nReads + 1) * 4 + 16, b->stream);
        if (e == hipSuccess) e = scratchAlloc((void **)&dOut, (size_t)total * 12 + 16, b->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dSrc, src.data(), nReads * 4, hipMemcpyHostToDevice, b->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dDst, dst.data(), (nReads + 1) * 4, hipMemcpyHostToDevice, b->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(routeGatherKernel, dim3((unsigned)nReads), dim3(64), 0, b->stream, (uint32_t)nReads, dSrc, dDst, b->dRouteOcc, dOut);
            e = hipMemcpyAsync(res, dOut, (size_t)total * 12, hipMemcpyDeviceToHost, b->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        if (dSrc) scratchFree(dSrc, b->stream);
        if (dDst) scratchFree(dDst, b->stream);
        if (dOut) scratchFree(dOut, b->stream);
        if (e != hipSuccess) { free(res); return fail(S3_EHIP, std::string("s3_batch_fetch_kept: ") + hipGetErrorString(e)); }
    }
    *occOut = res;
    return S3_OK;
}

// one kernel of this file's code object, asked for its attributes: the runtime loads a code object at the first use of one of its kernels (preloadKernels, s3dp_kernels.hip)
void s3::preloadRoute() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)routeCountKernel); (void)hipGetLastError(); }
