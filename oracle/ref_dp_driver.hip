// ref_dp_driver.hip -- TEST INFRASTRUCTURE ONLY.  Host driver around the REFERENCE's own semi-global DP device code
// (DV-DPfunctions.cu:35-512: DPScoreNHitPos, GenerateDPTable, SemiGlobalAligntment, GPUBacktrack), compiled for gfx950 with hipcc from the
// sources where they lie (oracle/ref.mk streams the region into oracle/_ref/obj/dp_device.inc; nothing of it enters the repository).
// CUDA and HIP share the kernel language, so the device code builds unchanged except for two stream edits, both forced by the hardware:
//   * CDNA has no texture units: the `texture <uint>` declarations are dropped and MC_ReadUnpack reads its array argument instead of
//     tex1Dfetch(texPatterns, .) -- the texture was bound to that very array (SemiGlobalAligner::init, DV-DPfunctions.cu:660);
//   * three lines of DV-DPfunctions.h (typedefs uint / uchar, DP_THREADS_PER_BLOCK) are streamed in, because the header itself pulls <cuda.h>.
// This file mirrors SemiGlobalAligner::init / performAlignment (DV-DPfunctions.cu:636-721): same buffers, same launches, same table size.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
using std::max;
using std::min;
#include "_ref/obj/dp_header_lines.inc"
#include "_ref/obj/dp_device.inc"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "dpref: %s: %s\n", #x, hipGetErrorString(e_)); return -1; } } while (0)

extern "C" int dpref_align(const uint32_t *packedDNA, const uint32_t *DNALengths, uint32_t maxDNALength, uint32_t maxDPTableLength,
                           const uint32_t *packedReads, const uint32_t *readLengths, uint32_t maxReadLength, uint32_t n,
                           const uint32_t *clipLt, uint32_t *clipRt /* in/out */, const uint32_t *anchorLeft, const uint32_t *anchorRight,
                           int match, int mismatch, int gapOpen, int gapExtend, const int32_t *cutoffs, int scheme,
                           int32_t *scores, uint32_t *hitLocs, uint32_t *startOffsets, uint32_t *maxScoreCounts, uint8_t *pattern)
{
    const size_t batch = ((size_t)n + 31) / 32 * 32;
    const size_t dnaW = batch * ((maxDNALength + 15) / 16), rdW = batch * ((maxReadLength + 15) / 16), patLen = maxReadLength + maxDPTableLength;
    void *table = nullptr;
    uint *dDna, *dRd, *dDnaLen, *dRdLen, *dHit, *dStart, *dCnt, *dCl = nullptr, *dCr = nullptr, *dAl = nullptr, *dAr = nullptr;
    int *dScores, *dCut;
    uchar *dPat;
    CK(hipMalloc(&table, (size_t)2 * maxDPTableLength * maxReadLength * batch * sizeof(short)));
    CK(hipMalloc((void **)&dDna, dnaW * 4)); CK(hipMalloc((void **)&dRd, rdW * 4));
    CK(hipMalloc((void **)&dDnaLen, batch * 4)); CK(hipMalloc((void **)&dRdLen, batch * 4));
    CK(hipMalloc((void **)&dHit, batch * 4)); CK(hipMalloc((void **)&dStart, batch * 4)); CK(hipMalloc((void **)&dCnt, batch * 4));
    CK(hipMalloc((void **)&dScores, batch * 4)); CK(hipMalloc((void **)&dCut, batch * 4)); CK(hipMalloc((void **)&dPat, batch * patLen));
    CK(hipMemset(dPat, 0, batch * patLen)); CK(hipMemset(dDnaLen, 0, batch * 4)); CK(hipMemset(dRdLen, 0, batch * 4));
    CK(hipMemcpy(dDna, packedDNA, dnaW * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dRd, packedReads, rdW * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dDnaLen, DNALengths, (size_t)n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dRdLen, readLengths, (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dCut, cutoffs, (size_t)n * 4, hipMemcpyHostToDevice));
    auto up = [&](const uint32_t *h, uint **d) -> int { if (!h) return 0; CK(hipMalloc((void **)d, batch * 4)); CK(hipMemcpy(*d, h, (size_t)n * 4, hipMemcpyHostToDevice)); return 0; };
    if (up(clipLt, &dCl) || up(clipRt, &dCr) || up(anchorLeft, &dAl) || up(anchorRight, &dAr)) return -1;
    const int blocks = ((int)n + DP_THREADS_PER_BLOCK - 1) / DP_THREADS_PER_BLOCK;
    SemiGlobalAligntment<<<blocks, DP_THREADS_PER_BLOCK>>>(dDna, dDnaLen, maxDNALength, maxDPTableLength, dRd, dRdLen, maxReadLength, dScores, dHit, dStart, dCl, dCr, dAl,
                                                          dAr, n, match, mismatch, gapOpen, gapExtend, table, dCnt, scheme);
    GPUBacktrack<<<blocks, DP_THREADS_PER_BLOCK>>>(dDna, dDnaLen, maxDNALength, maxDPTableLength, dRd, dRdLen, maxReadLength, dScores, dHit, dStart, dCl, dCr, dAl, n, match,
                                                  mismatch, gapOpen, gapExtend, dCut, table, dPat);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(scores, dScores, (size_t)n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hitLocs, dHit, (size_t)n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(startOffsets, dStart, (size_t)n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(maxScoreCounts, dCnt, (size_t)n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(pattern, dPat, (size_t)n * patLen, hipMemcpyDeviceToHost));
    if (clipRt) CK(hipMemcpy(clipRt, dCr, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (void *p : {table, (void *)dDna, (void *)dRd, (void *)dDnaLen, (void *)dRdLen, (void *)dHit, (void *)dStart, (void *)dCnt, (void *)dScores, (void *)dCut, (void *)dPat,
                    (void *)dCl, (void *)dCr, (void *)dAl, (void *)dAr}) if (p) (void)hipFree(p);
    return 0;
}
