// s3dp_dpapi.hip -- the semi-global DP part of the C-ABI: the aligner object (s3_dp_*) and the engines around the kernel for half-aligned
// pairs and single reads (window geometry, on-device packing, traceback decode).  Kernels: s3dp_dp.hip.
#include <algorithm>
#include <hipcub/hipcub.hpp>
#include <thread>
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
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void *p : h->allocs) (void)hipFree(p);
    for (auto e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

template <class T>
static int dalloc(s3_dp *h, T **p, size_t count)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T) + 64);
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
    tight[k] = rows[(size_t)p * rowPitch + o];
}

int s3i::dpFetchPatterns(s3_dp *h, uint32_t n, std::vector<uint8_t> &pat, uint32_t &pitch)
{
    uint32_t mx = 0;
    HIPCHK(d2h(&mx, h->dPatMax, 4, h->stream));                        // (synchronises the stream: the results queued before it have arrived too)
    if (mx == 0 || mx > h->patternLen) mx = h->patternLen;
    pitch = (mx + 3u) & ~3u;
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
    static FILE *f = nullptr;
    static bool tried = false;
    if (!tried) { const char *p = getenv("S3DP_DUMP_DP"); f = p && *p ? fopen(p, "w") : nullptr; tried = true; }
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
struct CigarEnc {
    std::vector<std::pair<char, int>> runs;
    char lastType = 'N';
    int lastCnt = 0;
    int count[128];
    int gapPenalty = 0;
    std::string text;
    void append(char type, int cnt)
    {
        if (lastType == type) lastCnt += cnt;
        else { runs.emplace_back(lastType, lastCnt); lastType = type; lastCnt = cnt; }
    }
    void encode(int gapOpen, int gapExtend)
    {
        memset(count, 0, sizeof count);
        gapPenalty = 0;
        append('N', 0);
        for (size_t i = runs.size(); i-- > 1;) {
            const char type = runs[i].first;
            const int cnt = runs[i].second;
            if (cnt > 0) {
                text += std::to_string(cnt);
                text.push_back(type);
                count[(int)type] += cnt;
                if (type == 'I' || type == 'D') gapPenalty += gapOpen + (cnt - 1) * gapExtend;
            }
        }
    }
};
}  // namespace

// Decode of one DP batch by the host threads: fn(id, results, pool) appends the results of problem id with CIGAR offsets relative to the pool it is
// given; the parts are joined in id order (fix adds the part's base to the offsets), so the output is what a serial loop would produce.
template <class R, class Fn, class Fix>
static void decodeParallel(uint32_t n, std::vector<R> &results, std::string &pool, Fn &&fn, Fix &&fix)
{
    unsigned T = std::thread::hardware_concurrency();
    if (T > 32) T = 32;
    if (T < 2 || n < 4096) { for (uint32_t id = 0; id < n; id++) fn(id, results, pool); return; }
    struct Part { std::vector<R> res; std::string pool; };
    std::vector<Part> parts(T);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&, t] {
            const uint32_t lo = (uint32_t)((uint64_t)n * t / T), hi = (uint32_t)((uint64_t)n * (t + 1) / T);
            for (uint32_t id = lo; id < hi; id++) fn(id, parts[t].res, parts[t].pool);
        });
    for (auto &x : th) x.join();
    for (unsigned t = 0; t < T; t++) {
        const uint32_t base = (uint32_t)pool.size();
        for (R &r : parts[t].res) { fix(r, base); results.push_back(r); }
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
    const uint32_t wordsPerDNA = (h->maxDNA + 15) / 16, wordsPerRead = (h->maxRead + 15) / 16;
    auto flush = [&]() -> int {
        const uint32_t n = (uint32_t)cand.size();
        if (!n) return S3_OK;
        hipStream_t s = h->stream;
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
        std::vector<int32_t> scores(n);
        std::vector<uint32_t> hit(n), cnt(n);
        std::vector<uint8_t> pat;
        uint32_t pitch = 0;
        HIPCHK(hipMemcpyAsync(scores.data(), h->dScores, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(hit.data(), h->dHit, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(cnt.data(), h->dCnt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        if ((rc = dpFetchPatterns(h, n, pat, pitch))) return rc;
        if ((rc = dpDump(h, n, true, true, true, true))) return rc;
        // algnmtCPUThread (:2329-2441)
        decodeParallel<s3_dp_half_result>(n, results, pool, [&](uint32_t id, std::vector<s3_dp_half_result> &results, std::string &pool) {
            s3_dp_half_result r{};
            const uint32_t alignedID = refer[id].readID, alignedIsMate = alignedID & 1u;
            r.readID = alignedID - alignedIsMate;
            r.cigarOffset = 0xFFFFFFFFu;
            uint32_t dpPos;
            if (scores[id] >= cutoffThreshold[1 - alignedIsMate]) {
                CigarEnc enc;
                char last = 'N';
                for (const uint8_t *p = pat.data() + (size_t)id * pitch; *p != 0; p++) {
                    if (*p == 'V') enc.append(last, (int)*(++p) - 1);
                    else { enc.append((char)*p, 1); last = (char)*p; }
                }
                enc.encode(h->sc.gapOpenScore, h->sc.gapExtendScore);
                r.cigarOffset = (uint32_t)pool.size();
                pool += enc.text;
                pool.push_back('\0');
                const int L = (int)rdLens[id] - enc.count['I'] - enc.count['S'];
                const int numMis = (L * h->sc.matchScore + enc.gapPenalty - scores[id]) / (h->sc.matchScore - h->sc.mismatchScore);
                r.editdist = (uint32_t)(enc.count['I'] + enc.count['D'] + numMis);
                r.whichFromDP = (uint8_t)(1 - alignedIsMate);
                dpPos = startLocs[id] + hit[id];
                if (dpPos < refer[id].pos) r.insertSize = refer[id].pos - dpPos + readLengths[alignedID];
                else r.insertSize = dpPos - refer[id].pos + rdLens[id] + (uint32_t)enc.count['D'] - (uint32_t)enc.count['I'] - (uint32_t)enc.count['S'];
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
        return S3_OK;
    };
    {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, (size_t)h->maxProblems * sizeof(DpCand) + 64);
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
    (void)hipFree(dCand);
    if (rc) return rc;
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
        hipError_t e = hipMalloc(&q, (size_t)h->maxProblems * sizeof(DpCand) + 64);
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
#define ECHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { (void)hipFree(dCand); return fail(S3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
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
        if ((rc = dpFetchPatterns(h, n, pat, pitch))) break;
#undef ECHK
        if ((rc = dpDump(h, n, true, true, false, false))) break;
        decodeParallel<s3_dp_single_result>(n, results, pool, [&](uint32_t i, std::vector<s3_dp_single_result> &results, std::string &pool) {
            if (scores[i] < cutoffThreshold) return;
            CigarEnc enc;
            char last = 'N';
            for (const uint8_t *p = pat.data() + (size_t)i * pitch; *p != 0; p++) {
                if (*p == 'V') enc.append(last, (int)*(++p) - 1);
                else { enc.append((char)*p, 1); last = (char)*p; }
            }
            enc.encode(h->sc.gapOpenScore, h->sc.gapExtendScore);
            s3_dp_single_result r{};
            r.readID = cand[i].readId; r.strand = (uint8_t)cand[i].strand; r.algnmt = cand[i].start + hit[i]; r.score = scores[i];
            const int L = (int)rdLens[i] - enc.count['I'] - enc.count['S'];
            const int numMis = (L * h->sc.matchScore + enc.gapPenalty - scores[i]) / (h->sc.matchScore - h->sc.mismatchScore);
            r.editdist = (uint32_t)(enc.count['I'] + enc.count['D'] + numMis);
            r.num_sameScore = cnt[i];
            r.cigarOffset = (uint32_t)pool.size();
            pool += enc.text;
            pool.push_back('\0');
            results.push_back(r);
        }, [](s3_dp_single_result &r, uint32_t base) { r.cigarOffset += base; });
    }
    (void)hipFree(dCand);
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
                    s3_dp_deep_result **resultsOut, size_t *nResultsOut, char **cigarPoolOut, uint8_t **tooManyHitsOut, const s3_occ *given, size_t nGiven);

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

static int deepImpl(s3_dp *h, s3_batch *b, s3_batch *seedBatch, const uint32_t *readLengths, const uint32_t *pairFirstReads, size_t nPairs, int stage,
                    uint32_t maxHitNum, int insertLow, int insertHigh, const int cutoffThreshold[2], int softClipLeft, int softClipRight,
                    s3_dp_deep_result **resultsOut, size_t *nResultsOut, char **cigarPoolOut, uint8_t **tooManyHitsOut, const s3_occ *given, size_t nGiven)
{
    if (!h || !b || !seedBatch || !readLengths || (!pairFirstReads && nPairs) || !cutoffThreshold || !resultsOut || !nResultsOut || !cigarPoolOut)
        return fail(S3_EINVAL, "s3_dp_deep: null argument");
    // new default DP: the end whose hits are given is not seeded (packSeedsOneSide seeds the mate of the pair's FIRST occurrence, DV-DPfunctions.cu:2756-2776)
    std::vector<uint8_t> givenEnd(nPairs, 255);
    for (size_t i = 0; i < nGiven; i++) {
        const uint32_t even = given[i].readID & ~1u;
        const uint32_t *it = std::lower_bound(pairFirstReads, pairFirstReads + nPairs, even);
        if (it == pairFirstReads + nPairs || *it != even) return fail(S3_EINVAL, "s3_dp_new_default: occurrence of a pair that is not listed");
        uint8_t &g = givenEnd[(size_t)(it - pairFirstReads)];
        if (g == 255) g = (uint8_t)(given[i].readID & 1u);
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
            if (givenEnd[k] != 255 && e == givenEnd[k]) continue;
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
    int rc = seedSearchKeys(b, seedBatch, seeds, maxHitNum, 30, keys, status);
    if (rc) return rc;
    // a pair with a seed dropped for too many hits is flagged (tooManyHitFlags, :2940-2944)
    std::vector<uint8_t> tooMany(nPairs, 0);
    if (tooManyHitsOut) {
        size_t j = 0;
        for (size_t k = 0; k < nPairs; k++)
            for (; j < seeds.size() && (seeds[j].keyHi & 0x3FFFFFFFu) == pairFirstReads[k]; j++) if (status[j] == 4) tooMany[k] = 1;
    }
    // four sorted runs: [end][strand], entries = (pair's first read, position)
    struct Pos { uint32_t id, pos; };
    std::vector<Pos> arr[2][2];
    for (uint64_t key : keys) {
        const uint32_t hi = (uint32_t)(key >> 32);
        arr[hi >> 31][(hi >> 30) & 1u].push_back(Pos{hi & 0x3FFFFFFFu, (uint32_t)key});
    }
    if (nGiven) {
        // the given hits stand in front of the seed hits (decodePositions :2896-2899) and everything is sorted by (pair, position) again
        for (size_t i = 0; i < nGiven; i++) {
            if (given[i].strand < 1 || given[i].strand > 2) return fail(S3_EINVAL, "s3_dp_new_default: bad strand");
            arr[given[i].readID & 1u][given[i].strand - 1].push_back(Pos{given[i].readID & ~1u, given[i].pos});
        }
        for (int e = 0; e < 2; e++)
            for (int st2 = 0; st2 < 2; st2++)
                std::stable_sort(arr[e][st2].begin(), arr[e][st2].end(), [](const Pos &x, const Pos &y) { return x.id != y.id ? x.id < y.id : x.pos < y.pos; });
    }
    // ---- pairEndMerge: left = '+' hits of one end (compressed by DP2_DIVIDE_GAP), right = '-' hits of the other end --------------------
    std::vector<PairCand> cands;
    auto merge = [&](std::vector<Pos> &left, const std::vector<Pos> &right, uint32_t leftIsMate) {
        size_t li = 0, ri = 0;
        while (li < left.size() && ri < right.size()) {
            if (left[li].id < right[ri].id) { li++; continue; }
            if (right[ri].id < left[li].id) { ri++; continue; }
            const uint32_t id = left[li].id;
            size_t le = li, re = ri;
            while (le < left.size() && left[le].id == id) le++;
            while (re < right.size() && right[re].id == id) re++;
            // MC_Compress on the left run
            size_t w = li;
            uint32_t prev = left[li].pos;
            for (size_t p = li + 1; p < le; p++) if (prev + 50u < left[p].pos) { left[++w] = left[p]; prev = left[p].pos; }
            const size_t lend = w + 1;
            const int readLength = (int)readLengths[id];                            // readLengths[readID] with the pair's first read, as the reference
            const int margin = readLength > 100 ? readLength >> 2 : 25;            // DP2_MARGIN
            int lengthLow = insertLow - readLength - margin;
            if (lengthLow < 0) lengthLow = 0;
            const int lengthHigh = insertHigh - readLength + margin;
            size_t lp = li, rp = ri;
            while (lp < lend && rp < re) {
                const uint32_t readLoc = left[lp].pos, mateLoc = right[rp].pos;
                if (readLoc + (uint32_t)lengthLow > mateLoc) rp++;
                else if (readLoc + (uint32_t)lengthHigh < mateLoc) lp++;
                else { cands.push_back(PairCand{{readLoc, mateLoc}, id + leftIsMate}); lp++; }
            }
            li = le; ri = re;
        }
    };
    merge(arr[0][leftLeg - 1], arr[1][rightLeg - 1], 0);                            // read left, mate right
    merge(arr[1][leftLeg - 1], arr[0][rightLeg - 1], 1);                            // mate left, read right
    std::stable_sort(cands.begin(), cands.end(), [](const PairCand &x, const PairCand &y) { return x.readIDLeft < y.readIDLeft; });
    // bit 1 of the per-pair flag: the pair has candidates (alignFlags of performSeeding, :3110-3121 -- only pairs WITHOUT candidates are
    // seeded again in round 2 or handed on as unseeded)
    if (tooManyHitsOut)
        for (const PairCand &c : cands) {
            const uint32_t even = c.readIDLeft & ~1u;
            const uint32_t *it = std::lower_bound(pairFirstReads, pairFirstReads + nPairs, even);
            if (it != pairFirstReads + nPairs && *it == even) tooMany[(size_t)(it - pairFirstReads)] |= 2;
        }
    // ---- two DP rounds per batch of candidates ---------------------------------------------------------------------------------------
    HIPCHK(hipSetDevice(h->dev));
    const uint32_t fullDNALength = b->ix->d.n;
    const uint32_t wordsPerDNA = (h->maxDNA + 15) / 16, wordsPerRead = (h->maxRead + 15) / 16;
    std::vector<s3_dp_deep_result> results;
    std::string pool;
    DpCand *dCand = nullptr;
    {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, (size_t)h->maxProblems * sizeof(DpCand) + 64);
        if (e != hipSuccess) return fail(S3_ENOMEM, std::string("hipMalloc(dp candidates): ") + hipGetErrorString(e));
        dCand = (DpCand *)q;
    }
#define ECHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { (void)hipFree(dCand); return fail(S3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
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
        { const int rc3 = dpFetchPatterns(h, n, R.pat, R.pitch); if (rc3) return rc3; }
        return dpDump(h, n, true, true, true, true);
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
                CigarEnc enc;
                char last = 'N';
                for (const uint8_t *p = rd[lOr]->pat.data() + (size_t)idx[lOr] * rd[lOr]->pitch; *p != 0; p++) {
                    if (*p == 'V') enc.append(last, (int)*(++p) - 1);
                    else { enc.append((char)*p, 1); last = (char)*p; }
                }
                enc.encode(h->sc.gapOpenScore, h->sc.gapExtendScore);
                cig[lOr] = enc.text;
                const int L = (int)rdR[k] - enc.count['I'] - enc.count['S'];
                const int numMis = (L * h->sc.matchScore + enc.gapPenalty - rd[lOr]->scores[idx[lOr]]) / (h->sc.matchScore - h->sc.mismatchScore);
                editdist[lOr] = enc.count['I'] + enc.count['D'] + numMis;
                DIS[lOr] = enc.count['D'] - enc.count['I'] - enc.count['S'];
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
    }
#undef ECHK
    (void)hipFree(dCand);
    if (rc) return rc;
    s3_dp_deep_result *out = (s3_dp_deep_result *)malloc(results.size() * sizeof(s3_dp_deep_result) + 1);
    char *cp = (char *)malloc(pool.size() + 1);
    uint8_t *tm = tooManyHitsOut ? (uint8_t *)malloc(nPairs + 1) : nullptr;
    if (!out || !cp || (tooManyHitsOut && !tm)) { free(out); free(cp); free(tm); return fail(S3_ENOMEM, "s3_dp_deep: out of host memory"); }
    memcpy(out, results.data(), results.size() * sizeof(s3_dp_deep_result));
    memcpy(cp, pool.data(), pool.size());
    if (tm) { memcpy(tm, tooMany.data(), nPairs); *tooManyHitsOut = tm; }
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

