// s3dp_dpapi.hip -- the semi-global DP part of the C-ABI: the aligner object (s3_dp_*) and the engines around the kernel for half-aligned
// pairs and single reads (window geometry, on-device packing, traceback decode).  Kernels: s3dp_dp.hip.
#include <algorithm>
#include <hipcub/hipcub.hpp>
#include <thread>
#include <chrono>
#include "s3dp_internal.h"

using namespace s3;
using namespace s3i;

// ---------------------------------------------------------------------------------------------
// semi-global DP
// ---------------------------------------------------------------------------------------------

extern "C" int s3_dp_config(int maxReadLength, int maxDNALength, int wantScheme, int *scheme, int *maxDPTableLength, int *patternLength)
{
    if (maxReadLength < 1 || maxDNALength < 1) return fail(S3_EINVAL, "s3_dp_config: bad lengths");
    int sch = wantScheme == 2 ? 2 : 1;
    int tab = sch == 1 ? maxDNALength : maxReadLength + maxReadLength / 2 + 8;      // DP_Effective_Region (:574)
    if (sch == 2 && tab > maxDNALength) { sch = 1; tab = maxDNALength; }
    if (scheme) *scheme = sch;
    if (maxDPTableLength) *maxDPTableLength = tab;
    if (patternLength) *patternLength = maxReadLength + tab;
    return S3_OK;
}

extern "C" void s3_dp_free(s3_dp *h)
{
    if (!h) return;
    (void)hipSetDevice(h->dev);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); scratchStreamGone(h->stream); }
    for (void *p : h->allocs) (void)hipFree(p);
    for (auto e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

template <class T>
static int dalloc(s3_dp *h, T **p, size_t count)
{
    void *q = nullptr;
    hipError_t e = s3i::mallocOrRelease(h->dev, &q, count * sizeof(T) + 64);    // (an aligner's tables before the flank tables of check-and-extend: s3_index::flankMu)
    if (e != hipSuccess) return fail(S3_ENOMEM, std::string("hipMalloc(dp): ") + hipGetErrorString(e));
    h->allocs.push_back(q);
    *p = (T *)q;
    return S3_OK;
}

extern "C" int s3_dp_create(int device, uint32_t maxProblems, uint32_t maxReadLength, uint32_t maxDNALength, uint32_t maxDPTableLength, int scheme,
                            const s3_dp_scores *sc, s3_dp **outp)
{
    if (!sc || !outp || !maxProblems || !maxReadLength || !maxDNALength || !maxDPTableLength) return fail(S3_EINVAL, "s3_dp_create: bad argument");
    if (maxReadLength > 4096 || maxDNALength > 12000) return fail(S3_EINVAL, "s3_dp_create: lengths too large (reads up to 4096, windows up to 12000 columns)");
    if (scheme == 1 && maxDPTableLength < maxDNALength) return fail(S3_EINVAL, "s3_dp_create: scheme 1 needs maxDPTableLength >= maxDNALength");
    // a wave keeps its window (1 byte per column) and one boundary row (12 bytes per column) in LDS: 160 KB per CU bound the window (s3dp_dp.hip::launchDp)
    if ((size_t)(((maxDNALength + 1u + 15u) & ~15u) + (maxDNALength + 1u) * 12u + 15u) > 160u * 1024u)
        return fail(S3_EINVAL, "s3_dp_create: maxDNALength above 12,000 columns does not fit the kernel's LDS-resident window");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(S3_ENODEV, "no HIP device available (libs3dp has no CPU path)");
    if (device < 0 || device >= ndev) return fail(S3_EINVAL, "bad device ordinal");
    HIPCHK(hipSetDevice(device));
    s3_dp *h = new s3_dp();
    h->dev = device; h->scheme = scheme == 2 ? 2 : 1; h->maxProblems = maxProblems; h->maxRead = maxReadLength; h->maxDNA = maxDNALength;
    h->maxTable = maxDPTableLength; h->patternLen = maxReadLength + maxDPTableLength; h->sc = *sc;
    const size_t rnd = ((size_t)maxProblems + 31) / 32 * 32;
    const size_t dnaW = rnd * ((maxDNALength + 15) / 16), rdW = rnd * ((maxReadLength + 15) / 16);
    h->tableWords = (size_t)((maxReadLength + 63) / 64) * dpTableChunkWords(maxDPTableLength + 64u);      // per 64-row chunk (s3dp_device.h)
    int rc;
#define DA(p, c) if ((rc = dalloc(h, &h->p, (c)))) { s3_dp_free(h); return rc; }
    DA(dDna, dnaW); DA(dReads, rdW); DA(dDnaLen, rnd); DA(dReadLen, rnd); DA(dClipLt, rnd); DA(dClipRt, rnd); DA(dAL, rnd); DA(dAR, rnd);
    DA(dCut, rnd); DA(dScores, rnd); DA(dHit, rnd); DA(dStart, rnd); DA(dCnt, rnd); DA(dCols, rnd);
    DA(dPattern, (size_t)maxProblems * h->patternLen); DA(dPatMax, 4); DA(dCells, 2);
    DA(dTable, h->tableWords * maxProblems);
#undef DA
#define HIPD(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { s3_dp_free(h); return fail(S3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    // (The aligner's stream at the device's lowest priority -- its many small kernels run beside the persistent search kernels of the next chunks -- was measured
    //  in round 5 on ten million pairs: 3.41 / 3.55 s with it, 3.42 s without.  Taken out in round 6.)
    HIPD(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPD(hipEventCreate(&h->ev[0])); HIPD(hipEventCreate(&h->ev[1]));
    HIPD(hipMemset(h->dCells, 0, 8));
#undef HIPD
    *outp = h;
    return S3_OK;
}

__global__ void dpGatherPatternsKernel(uint32_t n, const unsigned char *rows, uint32_t rowPitch, unsigned char *tight, uint32_t pitch)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (uint64_t)n * pitch) return;
    const uint32_t p = (uint32_t)(k / pitch), o = (uint32_t)(k - (uint64_t)p * pitch);
    // (the last byte of a row is always the terminator: a pattern that filled its whole row -- the traceback drops the terminator then -- must not
    //  run the decoders' loops into the next row)
    tight[k] = o + 1u == pitch ? (unsigned char)0 : rows[(size_t)p * rowPitch + o];
}

// ---- decode of the traceback on the device (CigarStringEncoder, edit-distance counts: DV-DPfunctions.cu:2363-2403, :3768-3818) -----------
// One lane per problem reads the pattern the traceback left (reversed string of M m D I S and 'V'<n> run markers) and writes a DECODED ROW into the
// problem's part of the traceback table, which nobody needs any more:  [int32 insertions][int32 deletions][int32 soft clips][int32 gap penalty]
// [CIGAR text, 0-terminated].  The CIGAR text is what the engines' host code used to build per problem: runs of equal operations merged in pattern
// order, printed from the last run to the first (the pattern is reversed), a zero-length run separating what it stands between.
__global__ void dpDecodeKernel(uint32_t n, const unsigned char *pattern, uint32_t patternLen, const int32_t *scores, const int32_t *cutoffs, int gapOpen, int gapExtend,
                               uint32_t *table, size_t tableWordsPerProblem, uint32_t *rowMax)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t *base = table + (size_t)p * tableWordsPerProblem;
    int32_t *hdr = (int32_t *)base;
    char *text = (char *)(base + 4);
    uint32_t *runs = base + 4 + (2u * patternLen + 8u) / 4u + 1u;             // (a run of one operation prints as two characters: the text never outgrows 2 x pattern)
    hdr[0] = hdr[1] = hdr[2] = hdr[3] = 0;
    text[0] = 0;
    if (scores[p] < cutoffs[p]) return;
    const unsigned char *pat = pattern + (size_t)p * patternLen;
    uint32_t nRuns = 0;
    int lastType = 'N', lastCnt = 0, last = 'N';
    auto append = [&](int type, int cnt) {
        if (lastType == type) lastCnt += cnt;
        else { if (lastCnt > 0) runs[nRuns++] = (uint32_t)lastType << 24 | (uint32_t)lastCnt; lastType = type; lastCnt = cnt; }
    };
    for (uint32_t k = 0; k < patternLen && pat[k] != 0; k++) {
        if (pat[k] == 'V') { k++; if (k < patternLen) append(last, (int)pat[k] - 1); }
        else { append((int)pat[k], 1); last = (int)pat[k]; }
    }
    append('N', 0);
    uint32_t o = 0;
    int nI = 0, nD = 0, nS = 0, gap = 0;
    for (uint32_t r = nRuns; r-- > 0;) {
        const int type = (int)(runs[r] >> 24), cnt = (int)(runs[r] & 0xFFFFFFu);
        char dig[12];
        int nd = 0, v = cnt;
        do { dig[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (nd) text[o++] = dig[--nd];
        text[o++] = (char)type;
        if (type == 'I') { nI += cnt; gap += gapOpen + (cnt - 1) * gapExtend; }
        else if (type == 'D') { nD += cnt; gap += gapOpen + (cnt - 1) * gapExtend; }
        else if (type == 'S') nS += cnt;
    }
    text[o] = 0;
    hdr[0] = nI; hdr[1] = nD; hdr[2] = nS; hdr[3] = gap;
    atomicMax(rowMax, 16u + o + 1u);
}

__global__ void dpGatherRowsKernel(uint32_t n, const uint32_t *table, size_t tableWordsPerProblem, unsigned char *tight, uint32_t pitch)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (uint64_t)n * pitch) return;
    const uint32_t p = (uint32_t)(k / pitch), o = (uint32_t)(k - (uint64_t)p * pitch);
    tight[k] = o + 1u == pitch ? (unsigned char)0 : ((const unsigned char *)(table + (size_t)p * tableWordsPerProblem))[o];
}

// the decoded rows of the last dpRun, `pitch` bytes each (see dpDecodeKernel).  A problem below its cutoff has an empty text.
int s3i::dpFetchDecoded(s3_dp *h, uint32_t n, std::vector<uint8_t> &rows, uint32_t &pitch)
{
    if (h->tableWords * 4u < 64u + 8u * (size_t)h->patternLen) return fail(S3_EINVAL, "dpFetchDecoded: the aligner's table is too small to decode in");
    hipStream_t s = h->stream;
    HIPCHK(hipMemsetAsync(h->dPatMax, 0, 4, s));
    if (n) hipLaunchKernelGGL(dpDecodeKernel, dim3((n + 63) / 64), dim3(64), 0, s, n, h->dPattern, h->patternLen, h->dScores, h->dCut, h->sc.gapOpenScore, h->sc.gapExtendScore,
                              h->dTable, h->tableWords, h->dPatMax);
    uint32_t mx = 0;
    HIPCHK(d2h(&mx, h->dPatMax, 4, s));
    if (mx < 20u) mx = 20u;
    pitch = ((mx + 3u) & ~3u) + 4u;
    rows.resize((size_t)n * pitch);
    if (n) {
        if ((size_t)n * pitch > (size_t)h->maxProblems * h->patternLen) return fail(S3_EINVAL, "dpFetchDecoded: no room to gather the rows");
        const uint64_t total = (uint64_t)n * pitch;
        hipLaunchKernelGGL(dpGatherRowsKernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, n, h->dTable, h->tableWords, h->dPattern, pitch);     // (the patterns are spent)
        HIPCHK(hipMemcpyAsync(rows.data(), h->dPattern, (size_t)total, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    return S3_OK;
}

int s3i::dpFetchPatterns(s3_dp *h, uint32_t n, std::vector<uint8_t> &pat, uint32_t &pitch)
{
    uint32_t mx = 0;
    HIPCHK(d2h(&mx, h->dPatMax, 4, h->stream));                        // (synchronises the stream: the results queued before it have arrived too)
    if (mx > h->patternLen) mx = h->patternLen;
    if (mx == 0) mx = 1;                                               // no problem of the launch reached its cutoff: nobody reads a pattern
    pitch = ((mx + 3u) & ~3u) + 4u;                                    // (room for the forced terminator behind the longest pattern)
    if (pitch > h->patternLen) pitch = h->patternLen;
    pat.resize((size_t)n * pitch);
    if (n) {
        // the rows are gathered on the device (into the traceback table, which nobody needs any more) and come over in one piece: a strided copy
        // to pageable memory goes row by row (measured: 850 ms for 30,000 rows)
        unsigned char *tight = (unsigned char *)h->dTable;
        if ((size_t)n * pitch > h->tableWords * (size_t)h->maxProblems * 4u) return fail(S3_EINVAL, "dpFetchPatterns: no room to gather the patterns");
        const uint64_t total = (uint64_t)n * pitch;
        hipLaunchKernelGGL(dpGatherPatternsKernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, n, h->dPattern, h->patternLen, tight, pitch);
        HIPCHK(hipMemcpyAsync(pat.data(), tight, (size_t)total, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return S3_OK;
}

// S3DP_DUMP_DP=<file>: every DP problem the engines submit, inputs as the packers built them and outputs as the decode stage receives them, one
// line per problem in the format of the reference-side dump (oracle/ref_dp_glue.cpp; tests/golden/*/dp_problems*.txt.gz):
//   P <readLen> <windowLen> <clipLt> <clipRt> <anchorL> <anchorR> <cutoff> <read> <window> | <score> <hitLoc> <ties> <ops>
static FILE *dpDumpFile()
{
    static FILE *f = [] { const char *p = getenv("S3DP_DUMP_DP"); return p && *p ? fopen(p, "w") : (FILE *)nullptr; }();     // (opened once, whichever thread comes first)
    return f;
}

void s3i::dpDumpNote(const char *line)
{
    if (FILE *f = dpDumpFile()) { fprintf(f, "# %s\n", line); fflush(f); }
}

int s3i::dpDump(s3_dp *h, uint32_t n, bool hasClipLt, bool hasClipRt, bool hasAL, bool hasAR)
{
    FILE *f = dpDumpFile();
    if (!f || !n) return S3_OK;
    const size_t rnd = ((size_t)n + 31) / 32 * 32;
    const uint32_t wD = (h->maxDNA + 15) / 16, wR = (h->maxRead + 15) / 16;
    std::vector<uint32_t> dna(rnd * wD), rd(rnd * wR), dl(n), rl(n), cl(n, 0), al(n, h->maxDNA), ar(n, 0), hit(n), cnt(n);
    std::vector<int32_t> cut(n), sc(n);
    std::vector<uint8_t> pat((size_t)n * h->patternLen);
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(dna.data(), h->dDna, dna.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(rd.data(), h->dReads, rd.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(dl.data(), h->dDnaLen, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(rl.data(), h->dReadLen, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (hasClipLt) HIPCHK(hipMemcpyAsync(cl.data(), h->dClipLt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (hasAL) HIPCHK(hipMemcpyAsync(al.data(), h->dAL, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (hasAR) HIPCHK(hipMemcpyAsync(ar.data(), h->dAR, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(cut.data(), h->dCut, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(sc.data(), h->dScores, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hit.data(), h->dHit, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(cnt.data(), h->dCnt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(pat.data(), h->dPattern, pat.size(), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    flockfile(f);                                                       // (two aligners may dump at once: the command line runs two DP stage threads)
    for (uint32_t t = 0; t < n; t++) {
        const size_t rb = (size_t)(t >> 5) * (wR << 5) + (t & 31), db = (size_t)(t >> 5) * (wD << 5) + (t & 31);
        fprintf(f, "P %u %u %u %u %u %u %d ", rl[t], dl[t], cl[t], hasClipRt && t < h->dumpClipRt.size() ? h->dumpClipRt[t] : 0u, al[t], ar[t], cut[t]);
        for (uint32_t i = 1; i <= rl[t]; i++) fputc("ACGT"[(rd[rb + ((size_t)(i >> 4) << 5)] >> ((15 - (i & 15)) << 1)) & 3], f);
        fputc(' ', f);
        for (uint32_t i = 1; i <= dl[t]; i++) fputc("ACGT"[(dna[db + ((size_t)(i >> 4) << 5)] >> ((15 - (i & 15)) << 1)) & 3], f);
        fprintf(f, " | %d %u %u ", sc[t], hit[t], cnt[t]);
        if (sc[t] >= cut[t]) {
            const uint8_t *p = pat.data() + (size_t)t * h->patternLen;
            for (uint32_t k = 0; p[k]; k++) {
                if (p[k] == 'V') { fprintf(f, "(%u)", (unsigned)p[++k]); if (p[k] == 0) break; }
                else fputc(p[k], f);
            }
        } else fputc('*', f);
        fputc('\n', f);
    }
    fflush(f);
    funlockfile(f);
    return S3_OK;
}

// launch the DP kernels on what is in the aligner's device buffers
// cells of a launch (read length x window length per problem), summed into the aligner's running total
__global__ void dpCellsKernel(uint32_t n, const uint32_t *dnaLen, const uint32_t *readLen, unsigned long long *total)
{
    unsigned long long c = 0;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) c += (unsigned long long)dnaLen[k] * readLen[k];
    for (int d = 32; d; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63u) == 0 && c) atomicAdd(total, c);
}

// the device time of the last launch joins the running total (its events are about to be recorded again)
static int dpFoldTiming(s3_dp *h)
{
    if (!h->timed || h->folded) return S3_OK;
    float ms = 0;
    HIPCHK(hipEventSynchronize(h->ev[1]));
    HIPCHK(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    h->totalMs += ms;
    h->folded = true;
    return S3_OK;
}

int s3i::dpRun(s3_dp *h, uint32_t n, bool hasClipLt, bool hasClipRt, bool hasAL, bool hasAR)
{
    hipStream_t s = h->stream;
    { int rc = dpFoldTiming(h); if (rc) return rc; }
    h->totalProblems += n;
    if (n) hipLaunchKernelGGL(dpCellsKernel, dim3(std::min<uint32_t>((n + 255u) / 256u, 1024u)), dim3(256), 0, s, n, h->dDnaLen, h->dReadLen, h->dCells);
    DpArgs a{};
    a.dna = h->dDna; a.dnaLengths = h->dDnaLen; a.reads = h->dReads; a.readLengths = h->dReadLen;
    a.clipLt = hasClipLt ? h->dClipLt : nullptr; a.clipRt = hasClipRt ? h->dClipRt : nullptr;
    a.anchorLeft = hasAL ? h->dAL : nullptr; a.anchorRight = hasAR ? h->dAR : nullptr;
    a.cutoffs = h->dCut; a.scores = h->dScores; a.hitLocs = h->dHit; a.startOffsets = h->dStart; a.maxScoreCounts = h->dCnt; a.tableCols = h->dCols;
    a.pattern = h->dPattern; a.table = h->dTable; a.tableWordsPerProblem = h->tableWords; a.patMax = h->dPatMax;
    HIPCHK(hipMemsetAsync(h->dPatMax, 0, 4, h->stream));
    a.n = n; a.maxDNALength = h->maxDNA; a.maxReadLength = h->maxRead; a.maxDPTableLength = h->maxTable; a.patternLen = h->patternLen;
    a.ldsPerWave = (((h->maxDNA + 1u + 15u) & ~15u) + (h->maxDNA + 1u) * 12u + 15u) & ~15u;
    a.ldsPerWavePair = ((h->maxDNA + 1u) * 20u + 15u) & ~15u;
    a.first = 0;
    { const char *e = getenv("S3DP_DP_PAIRED"); a.pairMode = !(e && *e == '0'); }
    { const char *e = getenv("S3DP_DP_FUSED"); a.fuseMode = e && *e == '1'; a.fuseTraceback = 0; }     // (measured slower on full launches: off unless asked for)
    a.match = h->sc.matchScore; a.mismatch = h->sc.mismatchScore; a.gapOpen = h->sc.gapOpenScore; a.gapExtend = h->sc.gapExtendScore; a.scheme = h->scheme;
    if ((size_t)a.ldsPerWave > 160 * 1024) return fail(S3_EINVAL, "s3_dp_align: maxDNALength too large for the LDS-resident window");
    if (dpDumpFile()) {
        h->dumpClipRt.assign(n, 0);
        if (hasClipRt) { HIPCHK(hipMemcpyAsync(h->dumpClipRt.data(), h->dClipRt, (size_t)n * 4, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s)); }
    }
    HIPCHK(hipEventRecord(h->ev[0], s));
    {
        const hipError_t le = (hipError_t)launchDp(a, s);
        if (le != hipSuccess) return fail(S3_EHIP, std::string("DP kernel launch: ") + hipGetErrorString(le));
    }
    HIPCHK(hipEventRecord(h->ev[1], s));
    HIPCHK(hipGetLastError());
    h->timed = true; h->folded = false;
    return S3_OK;
}

extern "C" int s3_dp_align(s3_dp *h, const uint32_t *packedDNA, const uint32_t *DNALengths, const uint32_t *packedReads, const uint32_t *readLengths,
                           const int32_t *cutoffThresholds, uint32_t n, const uint32_t *clipLtSizes, uint32_t *clipRtSizes,
                           const uint32_t *anchorLeftLocs, const uint32_t *anchorRightLocs, int32_t *scores, uint32_t *hitLocs,
                           uint32_t *maxScoreCounts, uint8_t *pattern)
{
    if (!h || !packedDNA || !DNALengths || !packedReads || !readLengths || !cutoffThresholds || !scores || !hitLocs || !maxScoreCounts || !pattern)
        return fail(S3_EINVAL, "s3_dp_align: null argument");
    if (n > h->maxProblems) return fail(S3_EINVAL, "s3_dp_align: too many problems for this aligner");
    for (uint32_t t = 0; t < n; t++) {
        if (readLengths[t] < 1 || readLengths[t] > h->maxRead) return fail(S3_EINVAL, "s3_dp_align: read length out of range");
        if (DNALengths[t] < 1 || DNALengths[t] > h->maxDNA) return fail(S3_EINVAL, "s3_dp_align: DNA length out of range");
        if (clipRtSizes && clipRtSizes[t] > readLengths[t]) return fail(S3_EINVAL, "s3_dp_align: clipRt larger than the read");
    }
    HIPCHK(hipSetDevice(h->dev));
    const size_t rnd = ((size_t)n + 31) / 32 * 32;
    const size_t dnaW = rnd * ((h->maxDNA + 15) / 16), rdW = rnd * ((h->maxRead + 15) / 16);
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(h->dDna, packedDNA, dnaW * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->dReads, packedReads, rdW * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->dDnaLen, DNALengths, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->dReadLen, readLengths, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->dCut, cutoffThresholds, (size_t)n * 4, hipMemcpyHostToDevice, s));
    if (clipLtSizes) HIPCHK(hipMemcpyAsync(h->dClipLt, clipLtSizes, (size_t)n * 4, hipMemcpyHostToDevice, s));
    if (clipRtSizes) HIPCHK(hipMemcpyAsync(h->dClipRt, clipRtSizes, (size_t)n * 4, hipMemcpyHostToDevice, s));
    if (anchorLeftLocs) HIPCHK(hipMemcpyAsync(h->dAL, anchorLeftLocs, (size_t)n * 4, hipMemcpyHostToDevice, s));
    if (anchorRightLocs) HIPCHK(hipMemcpyAsync(h->dAR, anchorRightLocs, (size_t)n * 4, hipMemcpyHostToDevice, s));
    {
        int rc = dpRun(h, n, clipLtSizes != nullptr, clipRtSizes != nullptr, anchorLeftLocs != nullptr, anchorRightLocs != nullptr);
        if (rc) return rc;
    }
    HIPCHK(hipMemcpyAsync(scores, h->dScores, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hitLocs, h->dHit, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(maxScoreCounts, h->dCnt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(pattern, h->dPattern, (size_t)n * h->patternLen, hipMemcpyDeviceToHost, s));
    if (clipRtSizes) HIPCHK(hipMemcpyAsync(clipRtSizes, h->dClipRt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    h->timed = true;
    return S3_OK;
}

// ---------------------------------------------------------------------------------------------
// Default DP for half-aligned pairs: one end has BWT hits, the other is aligned by DP inside the window the insert size allows.
// Replaces, of the reference's HalfEndAlignmentEngine (DV-DPfunctions.cu:1888-2540):
//   HalfEndAlgnBatch::pack      :2027-2109  window geometry, clip sizes, anchors               -> host loop below
//   packRead / repackDNA        :2111-2167  1-based MSB-first 32-interleaved words             -> dpPackKernel, from the RESIDENT text and reads
//   DP_Space::algnmtCPUThread   :2329-2441  pattern -> CIGAR, edit distance, insert size       -> decode loop below
//   CigarStringEncoder          DV-DPfunctions.h:548-597
// The reference packs windows on the host and copies them to the card; here text and reads already live in HBM, only the candidate
// list travels.  Pinned: the reference's own engine runs in the oracle harness (tests/test_gpu_dp_pipeline.py).
// ---------------------------------------------------------------------------------------------
struct DpCand { uint32_t readId, strand, start, dnaLen; };     // read to align (global index), strand to pack it in, window start / length

__global__ void dpPackKernel(uint32_t n, const DpCand *cand, const uint32_t *text, const uint32_t *queries, uint32_t wpq, const uint32_t *readLens,
                             uint32_t wordsPerDNA, uint32_t wordsPerRead, uint32_t *dna, uint32_t *reads)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per = wordsPerDNA + wordsPerRead;
    if (g >= (uint64_t)n * per) return;
    const uint32_t t = (uint32_t)(g / per), w = (uint32_t)(g % per);
    const DpCand c = cand[t];
    uint32_t out = 0;
    if (w < wordsPerDNA) {
        // characters i = 16 w .. 16 w + 15 of the 1-based sequence; character i sits at bit 2 (15 - (i & 15)); i = 0 stays empty
        for (uint32_t k = 0; k < 16u; k++) {
            const uint32_t i = w * 16u + k;
            if (i >= 1u && i <= c.dnaLen) {
                const uint64_t p = (uint64_t)c.start + i - 1u;
                out |= ((text[p >> 4] >> (2u * (15u - (uint32_t)(p & 15u)))) & 3u) << (2u * (15u - k));
            }
        }
        dna[(size_t)(t >> 5) * 32u * wordsPerDNA + (t & 31u) + (size_t)w * 32u] = out;
    } else {
        const uint32_t wr = w - wordsPerDNA, len = readLens[c.readId];
        const uint32_t *src = queries + (size_t)(c.readId >> 5) * 32u * wpq + (c.readId & 31u);
        for (uint32_t k = 0; k < 16u; k++) {
            const uint32_t i = wr * 16u + k;
            if (i >= 1u && i <= len) {
                const uint32_t q = c.strand == 1u ? i - 1u : len - i;
                uint32_t ch = (src[(size_t)(q >> 4) * 32u] >> ((q & 15u) * 2u)) & 3u;
                if (c.strand != 1u) ch = 3u - ch;
                out |= ch << (2u * (15u - k));
            }
        }
        reads[(size_t)(t >> 5) * 32u * wordsPerRead + (t & 31u) + (size_t)wr * 32u] = out;
    }
}

namespace {
// CigarStringEncoder (DV-DPfunctions.h:548-597): runs are collected in traceback order and written out reversed
// a decoded row (dpDecodeKernel): the counts the engines need and the CIGAR text
struct DecodedRow {
    int nI, nD, nS, gapPenalty;
    const char *text;
    DecodedRow(const std::vector<uint8_t> &rows, uint32_t pitch, uint32_t id)
    {
        const uint8_t *r = rows.data() + (size_t)id * pitch;
        int32_t hdr[4];
        memcpy(hdr, r, 16);
        nI = hdr[0]; nD = hdr[1]; nS = hdr[2]; gapPenalty = hdr[3];
        text = (const char *)r + 16;
    }
};
}  // namespace

// Decode of one DP batch by the host threads: fn(id, results, pool) appends the results of problem id with CIGAR offsets relative to the pool it is
// given; the parts are joined in id order (fix adds the part's base to the offsets), so the output is what a serial loop would produce.
template <class R, class Fn, class Fix>
static void decodeParallel(uint32_t n, std::vector<R> &results, std::string &pool, Fn &&fn, Fix &&fix)
{
    // (eight threads: the boxes show 256 processors and grant sixteen, and these calls run beside the command line's own pools; a part's vector and string
    //  sit on cache lines of their own -- neighbouring headers updated by neighbouring threads cost more than the threads brought)
    unsigned T = std::thread::hardware_concurrency();
    if (T > 8) T = 8;
    if (T < 2 || n < 4096) { for (uint32_t id = 0; id < n; id++) fn(id, results, pool); return; }
    struct alignas(128) Part { std::vector<R> res; std::string pool; };
    std::vector<Part> parts(T);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&, t] {
            const uint32_t lo = (uint32_t)((uint64_t)n * t / T), hi = (uint32_t)((uint64_t)n * (t + 1) / T);
            parts[t].res.reserve(hi - lo);
            parts[t].pool.reserve((size_t)(hi - lo) * 16);
            for (uint32_t id = lo; id < hi; id++) fn(id, parts[t].res, parts[t].pool);
        });
    for (auto &x : th) x.join();
    size_t total = results.size(), chars = pool.size();
    for (unsigned t = 0; t < T; t++) { total += parts[t].res.size(); chars += parts[t].pool.size(); }
    results.reserve(total); pool.reserve(chars);
    for (unsigned t = 0; t < T; t++) {
        const uint32_t base = (uint32_t)pool.size();
        for (R &r : parts[t].res) fix(r, base);
        results.insert(results.end(), parts[t].res.begin(), parts[t].res.end());
        pool += parts[t].pool;
    }
}

extern "C" int s3_dp_half_end(s3_dp *h, s3_batch *b, const uint32_t *readLengths, const s3_occ *occ, size_t nOcc, int insertLow, int insertHigh,
                              const int cutoffThreshold[2], int softClipLeft, int softClipRight, s3_dp_half_result **resultsOut, size_t *nResultsOut,
                              char **cigarPoolOut)
{
    if (!h || !b || !readLengths || (!occ && nOcc) || !cutoffThreshold || !resultsOut || !nResultsOut || !cigarPoolOut)
        return fail(S3_EINVAL, "s3_dp_half_end: null argument");
    if (h->dev != b->ix->dev) return fail(S3_EINVAL, "s3_dp_half_end: aligner and batch live on different devices");
    HIPCHK(hipSetDevice(h->dev));
    const uint32_t fullDNALength = b->ix->d.n;
    const int leftLeg = 1, rightLeg = 2;                                  // StrandArrangement "+/-" (IniParam.cpp:317-330)
    std::vector<s3_dp_half_result> results;
    std::string pool;
    std::vector<DpCand> cand;
    std::vector<uint32_t> dnaLens, rdLens, clipLt, clipRt, aL, aR, startLocs;
    std::vector<int32_t> cut;
    std::vector<uint8_t> leftOrRight;
    std::vector<s3_occ> refer;
    DpCand *dCand = nullptr;
    {
        const size_t cap = std::min<size_t>(h->maxProblems, 2 * nOcc + 2);
        cand.reserve(cap); dnaLens.reserve(cap); rdLens.reserve(cap); clipLt.reserve(cap); clipRt.reserve(cap); aL.reserve(cap); aR.reserve(cap); startLocs.reserve(cap);
        cut.reserve(cap); leftOrRight.reserve(cap); refer.reserve(cap);
    }
    const uint32_t wordsPerDNA = (h->maxDNA + 15) / 16, wordsPerRead = (h->maxRead + 15) / 16;
    static const bool verbose = getenv("S3DP_VERBOSE") != nullptr;
    auto tLap = std::chrono::steady_clock::now();
    double lapMs[6] = {0, 0, 0, 0, 0, 0};                                    // candidate lists, uploads + pack, kernels, fetch, records, (rest)
    auto lap = [&](int k) { const auto now = std::chrono::steady_clock::now(); lapMs[k] += std::chrono::duration<double, std::milli>(now - tLap).count(); tLap = now; };
    auto flush = [&]() -> int {
        const uint32_t n = (uint32_t)cand.size();
        if (!n) return S3_OK;
        hipStream_t s = h->stream;
        lap(0);
        HIPCHK(hipMemcpyAsync(dCand, cand.data(), (size_t)n * sizeof(DpCand), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(h->dDnaLen, dnaLens.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(h->dReadLen, rdLens.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(h->dCut, cut.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(h->dClipLt, clipLt.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(h->dClipRt, clipRt.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(h->dAL, aL.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(h->dAR, aR.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        const uint64_t threads = (uint64_t)n * (wordsPerDNA + wordsPerRead);
        hipLaunchKernelGGL(dpPackKernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, n, dCand, b->ix->d.pac, b->dQueries, b->wpq, b->dLens,
                           wordsPerDNA, wordsPerRead, h->dDna, h->dReads);
        int rc = dpRun(h, n, true, true, true, true);
        if (rc) return rc;
        if (verbose) { HIPCHK(hipStreamSynchronize(s)); }
        lap(1);
        std::vector<int32_t> scores(n);
        std::vector<uint32_t> hit(n), cnt(n);
        std::vector<uint8_t> pat;
        uint32_t pitch = 0;
        HIPCHK(hipMemcpyAsync(scores.data(), h->dScores, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(hit.data(), h->dHit, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(cnt.data(), h->dCnt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        if ((rc = dpDump(h, n, true, true, true, true))) return rc;
        if ((rc = dpFetchDecoded(h, n, pat, pitch))) return rc;
        lap(3);
        // algnmtCPUThread (:2329-2441)
        decodeParallel<s3_dp_half_result>(n, results, pool, [&](uint32_t id, std::vector<s3_dp_half_result> &results, std::string &pool) {
            s3_dp_half_result r{};
            const uint32_t alignedID = refer[id].readID, alignedIsMate = alignedID & 1u;
            r.readID = alignedID - alignedIsMate;
            r.cigarOffset = 0xFFFFFFFFu;
            uint32_t dpPos;
            if (scores[id] >= cutoffThreshold[1 - alignedIsMate]) {
                const DecodedRow enc(pat, pitch, id);
                r.cigarOffset = (uint32_t)pool.size();
                pool += enc.text;
                pool.push_back('\0');
                const int L = (int)rdLens[id] - enc.nI - enc.nS;
                const int numMis = (L * h->sc.matchScore + enc.gapPenalty - scores[id]) / (h->sc.matchScore - h->sc.mismatchScore);
                r.editdist = (uint32_t)(enc.nI + enc.nD + numMis);
                r.whichFromDP = (uint8_t)(1 - alignedIsMate);
                dpPos = startLocs[id] + hit[id];
                if (dpPos < refer[id].pos) r.insertSize = refer[id].pos - dpPos + readLengths[alignedID];
                else r.insertSize = dpPos - refer[id].pos + rdLens[id] + (uint32_t)enc.nD - (uint32_t)enc.nI - (uint32_t)enc.nS;
                r.num_sameScore = cnt[id];
            } else { r.whichFromDP = 2; dpPos = 0xFFFFFFFFu; }
            const uint8_t dpStrand = (uint8_t)(leftOrRight[id] == 0 ? leftLeg : rightLeg);
            if (!alignedIsMate) {
                r.algnmt_1 = refer[id].pos; r.algnmt_2 = dpPos; r.score_1 = refer[id].mismatchCount; r.score_2 = scores[id];
                r.strand_1 = refer[id].strand; r.strand_2 = dpStrand;
            } else {
                r.algnmt_1 = dpPos; r.algnmt_2 = refer[id].pos; r.score_1 = scores[id]; r.score_2 = refer[id].mismatchCount;
                r.strand_1 = dpStrand; r.strand_2 = refer[id].strand;
            }
            results.push_back(r);
        }, [](s3_dp_half_result &r, uint32_t base) { if (r.cigarOffset != 0xFFFFFFFFu) r.cigarOffset += base; });
        cand.clear(); dnaLens.clear(); rdLens.clear(); clipLt.clear(); clipRt.clear(); aL.clear(); aR.clear(); startLocs.clear(); cut.clear();
        leftOrRight.clear(); refer.clear();
        lap(4);
        return S3_OK;
    };
    {
        void *q = nullptr;
        hipError_t e = scratchAlloc(&q, (size_t)h->maxProblems * sizeof(DpCand) + 64, h->stream);
        if (e != hipSuccess) return fail(S3_ENOMEM, std::string("hipMalloc(dp candidates): ") + hipGetErrorString(e));
        dCand = (DpCand *)q;
    }
    int rc = S3_OK;
    for (size_t k = 0; k < nOcc && rc == S3_OK; k++) {
        const s3_occ &cur = occ[k];
        if (cur.strand != leftLeg && cur.strand != rightLeg) continue;
        if (cur.readID >= b->n) { rc = fail(S3_EINVAL, "s3_dp_half_end: read index out of range"); break; }
        const uint32_t alignedLen = readLengths[cur.readID];
        const int unalignedIsMate = 1 - (int)(cur.readID & 1u);
        const uint32_t unalignedID = unalignedIsMate == 0 ? cur.readID - 1 : cur.readID + 1;
        if (unalignedID >= b->n) { rc = fail(S3_EINVAL, "s3_dp_half_end: the mate is not in the batch"); break; }
        const uint32_t unalignedLen = readLengths[unalignedID];
        if (unalignedLen > h->maxRead) { rc = fail(S3_EINVAL, "s3_dp_half_end: read longer than the aligner's maxReadLength"); break; }
        auto add = [&](uint32_t start, uint32_t end, uint8_t lor, uint32_t anchorL, uint32_t anchorR, uint32_t strand) {
            DpCand c{unalignedID, strand, start, end - start};
            cand.push_back(c);
            dnaLens.push_back(end - start); rdLens.push_back(unalignedLen); startLocs.push_back(start);
            cut.push_back(cutoffThreshold[unalignedIsMate]);
            clipLt.push_back(strand == 1 ? (uint32_t)softClipLeft : (uint32_t)softClipRight);
            clipRt.push_back(strand == 1 ? (uint32_t)softClipRight : (uint32_t)softClipLeft);
            aL.push_back(anchorL); aR.push_back(anchorR);
            leftOrRight.push_back(lor); refer.push_back(cur);
        };
        if (cur.strand == leftLeg) {                                        // aligned end at the left, the other end somewhere to its right
            const uint32_t rightEnd = cur.pos + (uint32_t)insertHigh;
            uint32_t rightStart = cur.pos + (uint32_t)insertLow - unalignedLen;
            if (rightStart < cur.pos) rightStart = cur.pos;
            if (rightStart < fullDNALength && rightEnd <= fullDNALength && rightEnd - rightStart <= h->maxDNA)
                add(rightStart, rightEnd, 1, h->maxDNA, unalignedLen, (uint32_t)rightLeg);
        }
        if (cur.strand == rightLeg) {                                       // aligned end at the right
            const uint32_t leftStart = cur.pos + alignedLen - (uint32_t)insertHigh;
            uint32_t leftEnd = cur.pos + alignedLen - (uint32_t)insertLow + unalignedLen;
            if (leftEnd >= cur.pos + alignedLen) leftEnd = cur.pos + alignedLen - 1;
            if (leftStart < fullDNALength && leftEnd <= fullDNALength && leftEnd - leftStart <= h->maxDNA)
                add(leftStart, leftEnd, 0, (uint32_t)(insertHigh - insertLow + 1), 0, (uint32_t)leftLeg);
        }
        if (cand.size() + 1 > h->maxProblems) rc = flush();
    }
    if (rc == S3_OK) rc = flush();
    scratchFree(dCand, h->stream);
    if (rc) return rc;
    if (verbose) fprintf(stderr, "[s3dp] default DP call: %zu results: candidate lists %.2f ms, uploads + pack + kernels (synchronised) %.2f, fetch of scores and decoded rows %.2f, records %.2f\n",
                         results.size(), lapMs[0], lapMs[1], lapMs[3], lapMs[4]);
    s3_dp_half_result *out = (s3_dp_half_result *)malloc(results.size() * sizeof(s3_dp_half_result) + 1);
    char *cp = (char *)malloc(pool.size() + 1);
    if (!out || !cp) { free(out); free(cp); return fail(S3_ENOMEM, "s3_dp_half_end: out of host memory"); }
    memcpy(out, results.data(), results.size() * sizeof(s3_dp_half_result));
    memcpy(cp, pool.data(), pool.size());
    *resultsOut = out; *nResultsOut = results.size(); *cigarPoolOut = cp;
    return S3_OK;
}

// Single-end DP (SingleEndAlignmentEngine, DV-DPfunctions.cu:1373-1770): every seeding candidate gets the window
// [pos - margin, pos - margin + readLength + 2 margin) (SingleEndAlgnBatch::pack :1425-1467, margin = DPS_MARGIN), the read is packed
// in the candidate's strand, no anchors; results below the cutoff are dropped (SingleDP_Space::algnmtCPUThread :1682-1741).
extern "C" int s3_dp_single_end(s3_dp *h, s3_batch *b, const uint32_t *readLengths, const s3_seed_candidate *candidates, size_t nCand, int cutoffThreshold,
                                int softClipLeft, int softClipRight, s3_dp_single_result **resultsOut, size_t *nResultsOut, char **cigarPoolOut)
{
    if (!h || !b || !readLengths || (!candidates && nCand) || !resultsOut || !nResultsOut || !cigarPoolOut) return fail(S3_EINVAL, "s3_dp_single_end: null argument");
    if (h->dev != b->ix->dev) return fail(S3_EINVAL, "s3_dp_single_end: aligner and batch live on different devices");
    HIPCHK(hipSetDevice(h->dev));
    const uint32_t fullDNALength = b->ix->d.n;
    const uint32_t wordsPerDNA = (h->maxDNA + 15) / 16, wordsPerRead = (h->maxRead + 15) / 16;
    std::vector<s3_dp_single_result> results;
    std::string pool;
    DpCand *dCand = nullptr;
    {
        void *q = nullptr;
        hipError_t e = scratchAlloc(&q, (size_t)h->maxProblems * sizeof(DpCand) + 64, h->stream);
        if (e != hipSuccess) return fail(S3_ENOMEM, std::string("hipMalloc(dp candidates): ") + hipGetErrorString(e));
        dCand = (DpCand *)q;
    }
    int rc = S3_OK;
    for (size_t base = 0; base < nCand && rc == S3_OK; base += h->maxProblems) {
        const uint32_t n = (uint32_t)(nCand - base < h->maxProblems ? nCand - base : h->maxProblems);
        std::vector<DpCand> cand(n);
        std::vector<uint32_t> dnaLens(n), rdLens(n), clipLt(n), clipRt(n);
        std::vector<int32_t> cut(n, cutoffThreshold);
        for (uint32_t t = 0; t < n && rc == S3_OK; t++) {
            const s3_seed_candidate &c = candidates[base + t];
            if (c.readID >= b->n || (c.strand != 1 && c.strand != 2)) { rc = fail(S3_EINVAL, "s3_dp_single_end: bad candidate"); break; }
            const uint32_t L = readLengths[c.readID];
            const uint32_t margin = L > 100 ? L / 4 : 25;                      // DPS_MARGIN, DV-DPfunctions.h
            uint32_t start = c.pos - margin;
            if (start >= fullDNALength) start = 0;
            uint32_t len = L + 2 * margin;
            if (start + len > fullDNALength) len = fullDNALength - start;
            if (L > h->maxRead || len > h->maxDNA) { rc = fail(S3_EINVAL, "s3_dp_single_end: the aligner is too small for this read (maxReadLength / maxDNALength)"); break; }
            cand[t] = DpCand{c.readID, c.strand, start, len};
            dnaLens[t] = len; rdLens[t] = L;
            clipLt[t] = c.strand == 1 ? (uint32_t)softClipLeft : (uint32_t)softClipRight;
            clipRt[t] = c.strand == 1 ? (uint32_t)softClipRight : (uint32_t)softClipLeft;
        }
        if (rc) break;
        hipStream_t s = h->stream;
#define ECHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { scratchFree(dCand, h->stream); return fail(S3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
        ECHK(hipMemcpyAsync(dCand, cand.data(), (size_t)n * sizeof(DpCand), hipMemcpyHostToDevice, s));
        ECHK(hipMemcpyAsync(h->dDnaLen, dnaLens.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        ECHK(hipMemcpyAsync(h->dReadLen, rdLens.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        ECHK(hipMemcpyAsync(h->dCut, cut.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        ECHK(hipMemcpyAsync(h->dClipLt, clipLt.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        ECHK(hipMemcpyAsync(h->dClipRt, clipRt.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        const uint64_t threads = (uint64_t)n * (wordsPerDNA + wordsPerRead);
        hipLaunchKernelGGL(dpPackKernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, n, dCand, b->ix->d.pac, b->dQueries, b->wpq, b->dLens,
                           wordsPerDNA, wordsPerRead, h->dDna, h->dReads);
        rc = dpRun(h, n, true, true, false, false);
        if (rc) break;
        std::vector<int32_t> scores(n);
        std::vector<uint32_t> hit(n), cnt(n);
        std::vector<uint8_t> pat;
        uint32_t pitch = 0;
        ECHK(hipMemcpyAsync(scores.data(), h->dScores, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        ECHK(hipMemcpyAsync(hit.data(), h->dHit, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        ECHK(hipMemcpyAsync(cnt.data(), h->dCnt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
#undef ECHK
        if ((rc = dpDump(h, n, true, true, false, false))) break;
        if ((rc = dpFetchDecoded(h, n, pat, pitch))) break;
        decodeParallel<s3_dp_single_result>(n, results, pool, [&](uint32_t i, std::vector<s3_dp_single_result> &results, std::string &pool) {
            if (scores[i] < cutoffThreshold) return;
            const DecodedRow enc(pat, pitch, i);
            s3_dp_single_result r{};
            r.readID = cand[i].readId; r.strand = (uint8_t)cand[i].strand; r.algnmt = cand[i].start + hit[i]; r.score = scores[i];
            const int L = (int)rdLens[i] - enc.nI - enc.nS;
            const int numMis = (L * h->sc.matchScore + enc.gapPenalty - scores[i]) / (h->sc.matchScore - h->sc.mismatchScore);
            r.editdist = (uint32_t)(enc.nI + enc.nD + numMis);
            r.num_sameScore = cnt[i];
            r.cigarOffset = (uint32_t)pool.size();
            pool += enc.text;
            pool.push_back('\0');
            results.push_back(r);
        }, [](s3_dp_single_result &r, uint32_t base) { r.cigarOffset += base; });
    }
    scratchFree(dCand, h->stream);
    if (rc) return rc;
    s3_dp_single_result *out = (s3_dp_single_result *)malloc(results.size() * sizeof(s3_dp_single_result) + 1);
    char *cp = (char *)malloc(pool.size() + 1);
    if (!out || !cp) { free(out); free(cp); return fail(S3_ENOMEM, "s3_dp_single_end: out of host memory"); }
    memcpy(out, results.data(), results.size() * sizeof(s3_dp_single_result));
    memcpy(cp, pool.data(), pool.size());
    *resultsOut = out; *nResultsOut = results.size(); *cigarPoolOut = cp;
    return S3_OK;
}

// ---------------------------------------------------------------------------------------------
// Deep DP for pairs whose ends both failed the BWT stage.  Replaces, of the reference's DeepDP_Space (DV-DPfunctions.cu:2552-3900):
//   PairEndSeedingBatch::packSeeds :2681-2705          seeds of both ends of a pair (tables: s3_seed_positions)
//   decodePositions :2885-2958, findRevStart :2626     per end: estimated positions sorted by (strand, pair, position)
//   pairEndMerge :2780-2883                            '+' positions of one end against '-' positions of the other inside the insert window
//   decodeMergePositions :2960-2990                    both orientations, candidates ordered by the left read
//   PairEndAlgnBatch::packLeft / packRight :3374-3472  left end first; the right end's window is cut by the left hit and the insert bounds
//   DP2CPUAlgnThread :3731-3828                        both scores above their cutoffs -> CIGARs, edit distances, insert size
// Seeds and DP run on the device (the seed batch is a second s3_batch); the sequential merge runs on the host.  Pinned through the reference's own engines (tests/test_gpu_dp_pipeline.py).
// ---------------------------------------------------------------------------------------------
namespace {
struct PairCand { uint32_t pos[2]; uint32_t readIDLeft; };
}

// shared by the deep DP (both ends seeded) and the new default DP (one end's positions given by the BWT stage, only its mate seeded)
static int deepImpl(s3_dp *h, s3_batch *b, s3_batch *seedBatch, const uint32_t *readLengths, const uint32_t *pairFirstReads, size_t nPairs, int stage,
                    uint32_t maxHitNum, int insertLow, int insertHigh, const int cutoffThreshold[2], int softClipLeft, int softClipRight,
                    s3_dp_deep_result **resultsOut, size_t *nResultsOut, char **cigarPoolOut, uint8_t **tooManyHitsOut, const s3_occ *given, size_t nGiven,
                    const uint8_t *givenEndOnDevice = nullptr);

extern "C" int s3_dp_deep(s3_dp *h, s3_batch *b, s3_batch *seedBatch, const uint32_t *readLengths, const uint32_t *pairFirstReads, size_t nPairs, int stage,
                          uint32_t maxHitNum, int insertLow, int insertHigh, const int cutoffThreshold[2], int softClipLeft, int softClipRight,
                          s3_dp_deep_result **resultsOut, size_t *nResultsOut, char **cigarPoolOut, uint8_t **tooManyHitsOut)
{
    return deepImpl(h, b, seedBatch, readLengths, pairFirstReads, nPairs, stage, maxHitNum, insertLow, insertHigh, cutoffThreshold, softClipLeft, softClipRight, resultsOut,
                    nResultsOut, cigarPoolOut, tooManyHitsOut, nullptr, 0);
}

extern "C" int s3_dp_new_default(s3_dp *h, s3_batch *b, s3_batch *seedBatch, const uint32_t *readLengths, const s3_occ *occ, size_t nOcc, uint32_t maxHitNum,
                                 int insertLow, int insertHigh, const int cutoffThreshold[2], int softClipLeft, int softClipRight, s3_dp_deep_result **resultsOut,
                                 size_t *nResultsOut, char **cigarPoolOut, uint8_t **pairFlagsOut, uint32_t **pairFirstReadsOut, size_t *nPairsOut)
{
    if (!occ && nOcc) return fail(S3_EINVAL, "s3_dp_new_default: null argument");
    std::vector<uint32_t> pairs;
    for (size_t i = 0; i < nOcc; i++) {
        const uint32_t even = occ[i].readID & ~1u;
        if (!pairs.empty() && even < pairs.back()) return fail(S3_EINVAL, "s3_dp_new_default: occurrences must be grouped by pair, pairs ascending");
        if (pairs.empty() || pairs.back() != even) pairs.push_back(even);
    }
    int rc = deepImpl(h, b, seedBatch, readLengths, pairs.data(), pairs.size(), S3_STAGE_NEW_DEFAULT_DP, maxHitNum, insertLow, insertHigh, cutoffThreshold, softClipLeft,
                      softClipRight, resultsOut, nResultsOut, cigarPoolOut, pairFlagsOut, occ, nOcc);
    if (rc) return rc;
    if (pairFirstReadsOut) {
        uint32_t *p = (uint32_t *)malloc(pairs.size() * 4 + 4);
        if (!p) return fail(S3_ENOMEM, "s3_dp_new_default: out of host memory");
        memcpy(p, pairs.data(), pairs.size() * 4);
        *pairFirstReadsOut = p;
    }
    if (nPairsOut) *nPairsOut = pairs.size();
    return S3_OK;
}

// The same with the aligned ends' hits where s3_batch_route_pairs left them on the device: the list never comes to the host (11 M entries per chunk of 512 Ki
// pairs on a genome that is half repeats: 0.11 s to upload, as much again to walk on the host)
extern "C" int s3_dp_new_default_routed(s3_dp *h, s3_batch *b, s3_batch *seedBatch, const uint32_t *readLengths, uint32_t maxHitNum, int insertLow, int insertHigh,
                                        const int cutoffThreshold[2], int softClipLeft, int softClipRight, s3_dp_deep_result **resultsOut, size_t *nResultsOut,
                                        char **cigarPoolOut, uint8_t **pairFlagsOut, uint32_t **pairFirstReadsOut, size_t *nPairsOut)
{
    if (!b || !b->routed) return fail(S3_EINVAL, "s3_dp_new_default_routed: call s3_batch_route_pairs first");
    std::vector<uint32_t> pairs;
    std::vector<uint8_t> givenEnd;
    for (uint32_t p = 0; p < b->n / 2; p++)
        if (b->hPairRoute[p] & S3_ROUTE_NEW_DEFAULT) { pairs.push_back(2 * p); givenEnd.push_back((b->hPairRoute[p] & S3_ROUTE_GIVEN_END_2) ? 1 : 0); }
    const s3_occ *dGiven = (const s3_occ *)(b->dRouteOcc + (size_t)(b->n ? b->hRouteOffs[b->n] : 0u) * 3);
    int rc = deepImpl(h, b, seedBatch, readLengths, pairs.data(), pairs.size(), S3_STAGE_NEW_DEFAULT_DP, maxHitNum, insertLow, insertHigh, cutoffThreshold, softClipLeft,
                      softClipRight, resultsOut, nResultsOut, cigarPoolOut, pairFlagsOut, dGiven, (size_t)b->routing.nNewDefault, givenEnd.data());
    if (rc) return rc;
    if (pairFirstReadsOut) {
        uint32_t *p = (uint32_t *)malloc(pairs.size() * 4 + 4);
        if (!p) return fail(S3_ENOMEM, "s3_dp_new_default_routed: out of host memory");
        memcpy(p, pairs.data(), pairs.size() * 4);
        *pairFirstReadsOut = p;
    }
    if (nPairsOut) *nPairsOut = pairs.size();
    return S3_OK;
}

static int deepImpl(s3_dp *h, s3_batch *b, s3_batch *seedBatch, const uint32_t *readLengths, const uint32_t *pairFirstReads, size_t nPairs, int stage,
                    uint32_t maxHitNum, int insertLow, int insertHigh, const int cutoffThreshold[2], int softClipLeft, int softClipRight,
                    s3_dp_deep_result **resultsOut, size_t *nResultsOut, char **cigarPoolOut, uint8_t **tooManyHitsOut, const s3_occ *given, size_t nGiven,
                    const uint8_t *givenEndOnDevice)
{
    if (!h || !b || !seedBatch || !readLengths || (!pairFirstReads && nPairs) || !cutoffThreshold || !resultsOut || !nResultsOut || !cigarPoolOut)
        return fail(S3_EINVAL, "s3_dp_deep: null argument");
    // S3DP_VERBOSE: where a call's time goes (stderr)
    static const bool verbose = getenv("S3DP_VERBOSE") != nullptr;
    auto tLap = std::chrono::steady_clock::now();
    double lapMs[6] = {0, 0, 0, 0, 0, 0};                                     // seeds listed, seeds searched, positions sorted + merged, DP rounds, decode, rest
    auto lap = [&](int k) { const auto now = std::chrono::steady_clock::now(); lapMs[k] += std::chrono::duration<double, std::milli>(now - tLap).count(); tLap = now; };
    // new default DP: BOTH ends of a pair are seeded, the end whose hits are given too -- the reference's seeding driver calls packSeeds ( occ, stage )
    // (DV-DPfunctions.cu:3071 -> :2709-2731: the pair's first occurrence packs the seeds of both reads), not packSeedsOneSide (:2733-2776, which nothing calls).
    // Until round 6 the given end was left out here: a pair whose given list was cut at the hit limit before it reached the end's true place -- an end inside
    // a family of more than 8,000 near-identical copies -- found its place through that end's own seeds in the reference and not at all here (6 of 501 such
    // pairs of the benchmark's sample; found by bench.py's DP parity leg at the benchmark's size, which compares the DP problems of both sides).
    // (givenEndOnDevice: `given` is a device pointer and the given end of every pair comes with it)
    const bool givenOnDevice = givenEndOnDevice != nullptr;
    std::vector<uint8_t> givenEnd(nPairs, 255);
    if (givenOnDevice) givenEnd.assign(givenEndOnDevice, givenEndOnDevice + nPairs);
    else {
        const uint32_t *it = pairFirstReads;                                    // (the occurrences are grouped by pair, pairs ascending: the search resumes where it was)
        for (size_t i = 0; i < nGiven; i++) {
            const uint32_t even = given[i].readID & ~1u;
            if (it == pairFirstReads + nPairs || *it != even) it = std::lower_bound(pairFirstReads, pairFirstReads + nPairs, even);
            if (it == pairFirstReads + nPairs || *it != even) return fail(S3_EINVAL, "s3_dp_new_default: occurrence of a pair that is not listed");
            uint8_t &g = givenEnd[(size_t)(it - pairFirstReads)];
            if (g == 255) g = (uint8_t)(given[i].readID & 1u);
        }
    }
    if (b->ix != seedBatch->ix || b == seedBatch || h->dev != b->ix->dev) return fail(S3_EINVAL, "s3_dp_deep: batches / aligner do not belong together");
    const int leftLeg = 1, rightLeg = 2;
    // ---- seeds of both ends; key = end << 31 | strand << 30 | first read of the pair (30 bits), then the estimated position --------------
    std::vector<SeedDesc> seeds;
    for (size_t k = 0; k < nPairs; k++) {
        const uint32_t even = pairFirstReads[k];
        if ((even & 1u) || even + 1 >= b->n) return fail(S3_EINVAL, "s3_dp_deep: bad pair index");
        if (even >= (1u << 30)) return fail(S3_EINVAL, "s3_dp_deep: read index too large");
        for (uint32_t e = 0; e < 2; e++) {
            const uint32_t r = even + e;
            int seedLength = 0, num = 0, pos[128];
            if (s3_seed_positions(stage, (int)readLengths[r], &seedLength, pos, &num) != S3_OK) return fail(S3_EINVAL, "s3_dp_deep: bad stage");
            for (int i = 0; i < num; i++) {
                if (pos[i] < 0 || seedLength < 16 || (uint32_t)(pos[i] + seedLength) > readLengths[r]) continue;
                seeds.push_back(SeedDesc{r, (uint32_t)pos[i], (uint32_t)seedLength, e << 31 | even});
            }
        }
    }
    std::vector<uint64_t> keys;
    std::vector<uint8_t> status;
    uint64_t *dKeys = nullptr;
    uint32_t nKeys = 0;
    lap(0);
    int rc = seedSearchKeys(b, seedBatch, seeds, maxHitNum, 30, keys, status, &dKeys, &nKeys);
    if (rc) return rc;
    lap(1);
    // a pair with a seed dropped for too many hits is flagged (tooManyHitFlags, :2940-2944)
    std::vector<uint8_t> tooMany(nPairs, 0);
    if (tooManyHitsOut) {
        size_t j = 0;
        for (size_t k = 0; k < nPairs; k++)
            for (; j < seeds.size() && (seeds[j].keyHi & 0x3FFFFFFFu) == pairFirstReads[k]; j++) if (status[j] == 4) tooMany[k] = 1;
    }
    if (!givenOnDevice)
        for (size_t i = 0; i < nGiven; i++) if (given[i].strand < 1 || given[i].strand > 2) { scratchFree(dKeys, seedBatch->stream); return fail(S3_EINVAL, "s3_dp_new_default: bad strand"); }
    // ---- pairEndMerge on the device (s3dp_seed.hip): the positions never come to the host; the candidates do, ordered by the left read --------------------
    std::vector<PairCandHost> candsDev;
    std::vector<uint32_t> perPair;
    rc = pairMergeOnDevice(b, dKeys, nKeys, given, nGiven, givenOnDevice, pairFirstReads, nPairs, insertLow, insertHigh, candsDev, perPair);
    scratchFree(dKeys, seedBatch->stream);
    if (rc) return rc;
    std::vector<PairCand> cands(candsDev.size());
    for (size_t i = 0; i < candsDev.size(); i++) cands[i] = PairCand{{candsDev[i].pos[0], candsDev[i].pos[1]}, candsDev[i].readIDLeft};
    // bit 1 of the per-pair flag: the pair has candidates (alignFlags of performSeeding, :3110-3121 -- only pairs WITHOUT candidates are
    // seeded again in round 2 or handed on as unseeded)
    if (tooManyHitsOut) for (size_t k = 0; k < nPairs; k++) if (perPair[k]) tooMany[k] |= 2;
    lap(2);
    // ---- two DP rounds per batch of candidates ---------------------------------------------------------------------------------------
    HIPCHK(hipSetDevice(h->dev));
    const uint32_t fullDNALength = b->ix->d.n;
    const uint32_t wordsPerDNA = (h->maxDNA + 15) / 16, wordsPerRead = (h->maxRead + 15) / 16;
    std::vector<s3_dp_deep_result> results;
    std::string pool;
    DpCand *dCand = nullptr;
    {
        void *q = nullptr;
        hipError_t e = scratchAlloc(&q, (size_t)h->maxProblems * sizeof(DpCand) + 64, h->stream);
        if (e != hipSuccess) return fail(S3_ENOMEM, std::string("hipMalloc(dp candidates): ") + hipGetErrorString(e));
        dCand = (DpCand *)q;
    }
#define ECHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { scratchFree(dCand, h->stream); return fail(S3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    struct Round { std::vector<int32_t> scores; std::vector<uint32_t> hit, cnt; std::vector<uint8_t> pat; uint32_t pitch = 0; };
    auto runRound = [&](const std::vector<DpCand> &cand, const std::vector<uint32_t> &dnaLens, const std::vector<uint32_t> &rdLens, const std::vector<int32_t> &cut,
                        const std::vector<uint32_t> &clipLt, const std::vector<uint32_t> &clipRt, const std::vector<uint32_t> &aL, const std::vector<uint32_t> &aR,
                        Round &R) -> int {
        const uint32_t n = (uint32_t)cand.size();
        hipStream_t s = h->stream;
        ECHK(hipMemcpyAsync(dCand, cand.data(), (size_t)n * sizeof(DpCand), hipMemcpyHostToDevice, s));
        ECHK(hipMemcpyAsync(h->dDnaLen, dnaLens.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        ECHK(hipMemcpyAsync(h->dReadLen, rdLens.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        ECHK(hipMemcpyAsync(h->dCut, cut.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        ECHK(hipMemcpyAsync(h->dClipLt, clipLt.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        ECHK(hipMemcpyAsync(h->dClipRt, clipRt.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        ECHK(hipMemcpyAsync(h->dAL, aL.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        ECHK(hipMemcpyAsync(h->dAR, aR.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        const uint64_t threads = (uint64_t)n * (wordsPerDNA + wordsPerRead);
        hipLaunchKernelGGL(dpPackKernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, n, dCand, b->ix->d.pac, b->dQueries, b->wpq, b->dLens,
                           wordsPerDNA, wordsPerRead, h->dDna, h->dReads);
        int rc2 = dpRun(h, n, true, true, true, true);
        if (rc2) return rc2;
        R.scores.resize(n); R.hit.resize(n); R.cnt.resize(n);
        ECHK(hipMemcpyAsync(R.scores.data(), h->dScores, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        ECHK(hipMemcpyAsync(R.hit.data(), h->dHit, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        ECHK(hipMemcpyAsync(R.cnt.data(), h->dCnt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        { const int rc3 = dpDump(h, n, true, true, true, true); if (rc3) return rc3; }
        return dpFetchDecoded(h, n, R.pat, R.pitch);
    };
    auto window = [&](uint32_t pos, uint32_t readLength, uint32_t &start, uint32_t &len) {
        const uint32_t margin = readLength > 100 ? readLength >> 2 : 25;
        start = pos - margin;
        if (start >= fullDNALength) start = 0;
        len = readLength + margin * 2;
        if (start + len > fullDNALength) len = fullDNALength - start;
    };
    for (size_t base = 0; base < cands.size() && rc == S3_OK; base += h->maxProblems) {
        const uint32_t n = (uint32_t)(cands.size() - base < h->maxProblems ? cands.size() - base : h->maxProblems);
        // left ends (packLeft): no anchors
        std::vector<DpCand> cand(n);
        std::vector<uint32_t> dnaLens(n), rdLens(n), clipLt(n), clipRt(n), aL(n, h->maxDNA), aR(n, 0), startL(n);
        std::vector<int32_t> cut(n);
        for (uint32_t i = 0; i < n && rc == S3_OK; i++) {
            const PairCand &c = cands[base + i];
            const uint32_t L = readLengths[c.readIDLeft];
            uint32_t st, len;
            window(c.pos[0], L, st, len);
            if (L > h->maxRead || len > h->maxDNA) { rc = fail(S3_EINVAL, "s3_dp_deep: the aligner is too small for this read"); break; }
            cand[i] = DpCand{c.readIDLeft, (uint32_t)leftLeg, st, len};
            dnaLens[i] = len; rdLens[i] = L; startL[i] = st;
            cut[i] = cutoffThreshold[c.readIDLeft & 1u];
            clipLt[i] = leftLeg == 1 ? (uint32_t)softClipLeft : (uint32_t)softClipRight;
            clipRt[i] = leftLeg == 1 ? (uint32_t)softClipRight : (uint32_t)softClipLeft;
        }
        if (rc) break;
        Round RL, RR;
        if ((rc = runRound(cand, dnaLens, rdLens, cut, clipLt, clipRt, aL, aR, RL))) break;
        // right ends (packRight) of the candidates whose left end passed; the others are not aligned again
        std::vector<uint32_t> sel;
        for (uint32_t i = 0; i < n; i++) if (RL.scores[i] >= cutoffThreshold[cands[base + i].readIDLeft & 1u]) sel.push_back(i);
        const uint32_t m = (uint32_t)sel.size();
        std::vector<DpCand> candR(m);
        std::vector<uint32_t> dnaR(m), rdR(m), clipLtR(m), clipRtR(m), aLR(m, h->maxDNA), aRR(m), startR(m);
        std::vector<int32_t> cutR(m);
        for (uint32_t k = 0; k < m && rc == S3_OK; k++) {
            const uint32_t i = sel[k];
            const PairCand &c = cands[base + i];
            const uint32_t leftIsOdd = c.readIDLeft & 1u, right = leftIsOdd ? c.readIDLeft - 1 : c.readIDLeft + 1;
            const uint32_t L = readLengths[right];
            uint32_t st, len;
            window(c.pos[1], L, st, len);
            const uint32_t hitPosLeft = startL[i] + RL.hit[i];
            const uint32_t bounded = hitPosLeft + (uint32_t)insertHigh - st;            // restrict the maximum insert size
            if (bounded < len) len = bounded;
            const int rightAnchor = (int)(hitPosLeft + (uint32_t)insertLow - st);      // restrict the minimum insert size
            if (L > h->maxRead || len > h->maxDNA || len == 0) { rc = fail(S3_EINVAL, "s3_dp_deep: right window out of range for the aligner"); break; }
            candR[k] = DpCand{right, (uint32_t)rightLeg, st, len};
            dnaR[k] = len; rdR[k] = L; startR[k] = st; aRR[k] = rightAnchor > 0 ? (uint32_t)rightAnchor : 0u;
            cutR[k] = cutoffThreshold[right & 1u];
            clipLtR[k] = rightLeg == 1 ? (uint32_t)softClipLeft : (uint32_t)softClipRight;
            clipRtR[k] = rightLeg == 1 ? (uint32_t)softClipRight : (uint32_t)softClipLeft;
        }
        if (rc) break;
        if (m && (rc = runRound(candR, dnaR, rdR, cutR, clipLtR, clipRtR, aLR, aRR, RR))) break;
        lap(3);
        // DP2CPUAlgnThread: both ends above their cutoffs
        decodeParallel<s3_dp_deep_result>(m, results, pool, [&](uint32_t k, std::vector<s3_dp_deep_result> &results, std::string &pool) {
            const uint32_t i = sel[k];
            const PairCand &c = cands[base + i];
            const int readSide = (int)(c.readIDLeft & 1u), mateSide = 1 - readSide;
            if (RR.scores[k] < cutoffThreshold[mateSide]) return;
            // index 0 = left end, 1 = right end.  Both edit distances use the RIGHT end's length (batch->lengths[] after packRight), as the reference
            const Round *rd[2] = {&RL, &RR};
            const uint32_t idx[2] = {i, k};
            std::string cig[2];
            int editdist[2], DIS[2];
            for (int lOr = 0; lOr < 2; lOr++) {
                const DecodedRow enc(rd[lOr]->pat, rd[lOr]->pitch, idx[lOr]);
                cig[lOr] = enc.text;
                const int L = (int)rdR[k] - enc.nI - enc.nS;
                const int numMis = (L * h->sc.matchScore + enc.gapPenalty - rd[lOr]->scores[idx[lOr]]) / (h->sc.matchScore - h->sc.mismatchScore);
                editdist[lOr] = enc.nI + enc.nD + numMis;
                DIS[lOr] = enc.nD - enc.nI - enc.nS;
            }
            const uint32_t posOf[2] = {startL[i] + RL.hit[i], startR[k] + RR.hit[k]};
            s3_dp_deep_result r{};
            r.readID = c.readIDLeft - (uint32_t)readSide;
            r.strand_1 = (uint8_t)(readSide == 0 ? leftLeg : rightLeg);
            r.strand_2 = (uint8_t)(mateSide == 0 ? leftLeg : rightLeg);
            r.algnmt_1 = posOf[readSide]; r.algnmt_2 = posOf[mateSide];
            r.score_1 = rd[readSide]->scores[idx[readSide]]; r.score_2 = rd[mateSide]->scores[idx[mateSide]];
            r.editdist_1 = (uint32_t)editdist[readSide]; r.editdist_2 = (uint32_t)editdist[mateSide];
            r.insertSize = r.algnmt_1 < r.algnmt_2 ? r.algnmt_2 - r.algnmt_1 + rdR[k] + (uint32_t)DIS[mateSide] : r.algnmt_1 - r.algnmt_2 + rdR[k] + (uint32_t)DIS[readSide];
            r.num_sameScore_1 = rd[readSide]->cnt[idx[readSide]]; r.num_sameScore_2 = rd[mateSide]->cnt[idx[mateSide]];
            r.cigarOffset_1 = (uint32_t)pool.size(); pool += cig[readSide]; pool.push_back('\0');
            r.cigarOffset_2 = (uint32_t)pool.size(); pool += cig[mateSide]; pool.push_back('\0');
            results.push_back(r);
        }, [](s3_dp_deep_result &r, uint32_t base) { r.cigarOffset_1 += base; r.cigarOffset_2 += base; });
        lap(4);
    }
#undef ECHK
    scratchFree(dCand, h->stream);
    if (rc) return rc;
    s3_dp_deep_result *out = (s3_dp_deep_result *)malloc(results.size() * sizeof(s3_dp_deep_result) + 1);
    char *cp = (char *)malloc(pool.size() + 1);
    uint8_t *tm = tooManyHitsOut ? (uint8_t *)malloc(nPairs + 1) : nullptr;
    if (!out || !cp || (tooManyHitsOut && !tm)) { free(out); free(cp); free(tm); return fail(S3_ENOMEM, "s3_dp_deep: out of host memory"); }
    memcpy(out, results.data(), results.size() * sizeof(s3_dp_deep_result));
    memcpy(cp, pool.data(), pool.size());
    if (tm) { memcpy(tm, tooMany.data(), nPairs); *tooManyHitsOut = tm; }
    lap(5);
    if (verbose)
        fprintf(stderr, "[s3dp] deep DP stage %d: %zu pairs, %zu given hits, %zu seeds, %zu positions, %zu candidates, %zu results: seeds listed %.1f ms, searched %.1f, sorted + merged %.1f, "
                        "DP rounds %.1f, decode %.1f, rest %.1f\n", stage, nPairs, nGiven, seeds.size(), (size_t)nKeys, cands.size(), results.size(), lapMs[0], lapMs[1], lapMs[2], lapMs[3],
                lapMs[4], lapMs[5]);
    *resultsOut = out; *nResultsOut = results.size(); *cigarPoolOut = cp;
    return S3_OK;
}

extern "C" int s3_dp_totals(s3_dp *h, double *kernelMs, uint64_t *problems, uint64_t *cells)
{
    if (!h) return fail(S3_EINVAL, "null aligner");
    HIPCHK(hipSetDevice(h->dev));
    HIPCHK(hipStreamSynchronize(h->stream));
    { int rc = dpFoldTiming(h); if (rc) return rc; }
    unsigned long long c = 0;
    HIPCHK(hipMemcpy(&c, h->dCells, 8, hipMemcpyDeviceToHost));
    if (kernelMs) *kernelMs = h->totalMs;
    if (problems) *problems = h->totalProblems;
    if (cells) *cells = c;
    return S3_OK;
}

extern "C" int s3_dp_last_timing(s3_dp *h, float *ms)
{
    if (!h || !h->timed || !ms) return fail(S3_EINVAL, "no timed DP call");
    HIPCHK(hipSetDevice(h->dev));
    HIPCHK(hipEventElapsedTime(ms, h->ev[0], h->ev[1]));
    return S3_OK;
}

// one kernel of this file's code object, asked for its attributes: the runtime loads a code object at the first use of one of its kernels (preloadKernels, s3dp_kernels.hip)
void s3::preloadDpApi() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)dpCellsKernel); (void)hipGetLastError(); }
