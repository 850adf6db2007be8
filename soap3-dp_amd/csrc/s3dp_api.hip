// s3dp_api.hip -- the C-ABI of libs3dp.so (include/s3dp.h): index residency, batch objects, the
// k-mismatch search call and the single-end all-best ladder.  Host logic only; kernels are in
// s3dp_kernels.hip.  No CPU fallback exists: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/s3dp.h"
#include "s3dp_device.h"
#include "s3dp_model.h"

using namespace s3;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                                          \
    do {                                                                                                   \
        hipError_t e_ = (x);                                                                               \
        if (e_ != hipSuccess) return fail(S3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_));        \
    } while (0)

extern "C" const char *s3_last_error(void) { return g_err.c_str(); }
extern "C" const char *s3_version(void) { return "s3dp 0.1 (gfx950)"; }
extern "C" void s3_free(void *p) { free(p); }

extern "C" int s3_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int s3_model_describe(int readLength, int maxError, char *buf, size_t bufLen)
{
    if (readLength < 1 || readLength > S3_MAX_READ_LENGTH || maxError < 0 || maxError > S3_MAX_MISMATCH) return fail(S3_EINVAL, "bad model args");
    return s3_model_text(readLength, maxError, buf, (unsigned long)bufLen);
}

// ---------------------------------------------------------------------------------------------
struct s3_index {
    int       dev = 0;
    DevIndex  d{};
    std::vector<void *> allocs;
    S3Model  *models[S3_MAX_MISMATCH + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // device, indexed by read length
    size_t    bytes = 0;
};

template <class T>
static int devAlloc(s3_index *ix, T **p, size_t count)
{
    void *q = nullptr;
    size_t b = count * sizeof(T);
    if (b == 0) b = sizeof(T);
    hipError_t e = hipMalloc(&q, b);
    if (e != hipSuccess) return fail(S3_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    ix->allocs.push_back(q);
    ix->bytes += b;
    *p = (T *)q;
    return S3_OK;
}

static int uploadHalf(s3_index *ix, const s3_bwt_desc &h, DevBwt *out, hipStream_t s)
{
    if (!h.bwt || !h.occ) return fail(S3_EINVAL, "s3_index_upload: missing bwt/occ array");
    const size_t expectRecs = ((size_t)h.cumFreq[4] + 127) / 128 + 1;
    if (h.occRecs != expectRecs) return fail(S3_EINVAL, "s3_index_upload: occRecs does not match textLength");
    uint32_t *tb = nullptr, *to = nullptr;
    uint4 *lines = nullptr;
    HIPCHK(hipMalloc((void **)&tb, h.bwtWords * 4 + 16));
    HIPCHK(hipMalloc((void **)&to, h.occRecs * 16));
    int rc = devAlloc(ix, &lines, h.occRecs * 4);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(tb, h.bwt, h.bwtWords * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(to, h.occ, h.occRecs * 16, hipMemcpyHostToDevice, s));
    launchRelayout(tb, h.bwtWords, to, h.occRecs, lines, s);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(tb));
    HIPCHK(hipFree(to));
    out->lines = lines;
    out->inverseSa0 = h.inverseSa0;
    out->n = h.cumFreq[4];
    return S3_OK;
}

extern "C" int s3_index_upload(int device, const s3_index_desc *desc, s3_index **outp)
{
    if (!desc || !outp) return fail(S3_EINVAL, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(S3_ENODEV, "no HIP device available (libs3dp has no CPU path)");
    if (device < 0 || device >= ndev) return fail(S3_EINVAL, "bad device ordinal");
    if (!desc->sa || !desc->packedDNA || !desc->lkt || !desc->revLkt) return fail(S3_EINVAL, "sa, packedDNA and both lookup tables are required");
    if (desc->lktSize < 1 || desc->lktSize > 15) return fail(S3_EINVAL, "bad lookup table size");
    HIPCHK(hipSetDevice(device));
    s3_index *ix = new s3_index();
    ix->dev = device;
    hipStream_t s;
    HIPCHK(hipStreamCreate(&s));
    int rc = uploadHalf(ix, desc->fwd, &ix->d.fwd, s);
    if (!rc) rc = uploadHalf(ix, desc->rev, &ix->d.rev, s);
    if (rc) { s3_index_free(ix); return rc; }
    const uint32_t n = desc->textLength;
    uint32_t *sa = nullptr, *pac = nullptr, *lkt = nullptr, *rlkt = nullptr;
    const size_t ltWords = (size_t)1 << (2 * desc->lktSize);
    const size_t pacWords = desc->packedWords + 4;                     // + zero tail so 3-word windows never leave the buffer
    if ((rc = devAlloc(ix, &sa, (size_t)n + 1)) || (rc = devAlloc(ix, &pac, pacWords)) || (rc = devAlloc(ix, &lkt, ltWords)) ||
        (rc = devAlloc(ix, &rlkt, ltWords))) { s3_index_free(ix); return rc; }
    HIPCHK(hipMemcpyAsync(sa, desc->sa, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(pac, 0, pacWords * 4, s));
    HIPCHK(hipMemcpyAsync(pac, desc->packedDNA, desc->packedWords * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(lkt, desc->lkt, ltWords * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(rlkt, desc->revLkt, ltWords * 4, hipMemcpyHostToDevice, s));
    ix->d.sa = sa; ix->d.pac = pac; ix->d.pacWords = pacWords; ix->d.lkt = lkt; ix->d.rlkt = rlkt;
    ix->d.ltSize = desc->lktSize; ix->d.n = n;
    // model tables for every read length and mismatch budget (built on the host, see s3dp_model.cpp)
    std::vector<S3Model> host(S3_MAX_READ_LENGTH + 1);
    for (int e = 0; e <= S3_MAX_MISMATCH; e++) {
        memset(host.data(), 0, host.size() * sizeof(S3Model));
        for (int L = 1; L <= S3_MAX_READ_LENGTH; L++) s3_model_build(L, e, &host[L]);
        if ((rc = devAlloc(ix, &ix->models[e], host.size()))) { s3_index_free(ix); return rc; }
        HIPCHK(hipMemcpyAsync(ix->models[e], host.data(), host.size() * sizeof(S3Model), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipStreamDestroy(s));
    *outp = ix;
    return S3_OK;
}

extern "C" void s3_index_free(s3_index *ix)
{
    if (!ix) return;
    (void)hipSetDevice(ix->dev);
    for (void *p : ix->allocs) (void)hipFree(p);
    delete ix;
}

extern "C" size_t s3_index_device_bytes(const s3_index *ix) { return ix ? ix->bytes : 0; }

// ---------------------------------------------------------------------------------------------
struct s3_batch {
    s3_index *ix = nullptr;
    hipStream_t stream = nullptr;
    uint32_t maxReads = 0, wpq = 0, n = 0;
    size_t   maxItems = 0;
    uint32_t recCap = 0, occCap = 0;
    // inputs
    uint32_t *dQueries = nullptr, *dLens = nullptr, *dReadsT = nullptr, *dSeedLens = nullptr;
    bool longMode = false;          // seed lengths set: the reference's long-read mode
    uint32_t fullUniformLen = 0;
    // search / ordering work buffers
    uint4    *dRecs = nullptr, *dOrdered = nullptr;
    uint32_t *dRecCount = nullptr, *dFlags = nullptr, *dItemCounts = nullptr, *dItemOffsets = nullptr;
    unsigned long long *dCounters = nullptr, *dWorkHead = nullptr;
    uint32_t uniformLen = 0;
    uint32_t *dKept = nullptr, *dOccCounts = nullptr, *dCeCounts = nullptr, *dTotalOcc = nullptr, *dWithError = nullptr;
    uint32_t *dHitOffsets = nullptr;
    DevHit   *dHits = nullptr;
    char     *dTemp = nullptr; size_t tempBytes = 0;
    // ladder
    uint32_t *dActive[2] = {nullptr, nullptr}, *dNumSel = nullptr, *dIota = nullptr;
    uint8_t  *dState = nullptr, *dSelFlags = nullptr;
    uint32_t *dOccOffsets = nullptr, *dMasked = nullptr;
    uint32_t *dOccPos = nullptr, *dOccMeta = nullptr;
    uint8_t  *dPhase = nullptr;
    uint32_t *dFinalWithError = nullptr, *dOccStart = nullptr, *dOccCount = nullptr;
    uint32_t *dOutOffsets = nullptr, *dOutPos = nullptr, *dOutMeta = nullptr;
    // paired-end ladder (allocated by the first s3_batch_align_pair)
    bool pairReady = false;
    DevPair  *dPairs = nullptr;
    uint8_t  *dPairMin = nullptr, *dPairNext = nullptr;
    uint32_t *dReadList = nullptr, *dPairActive[2] = {nullptr, nullptr}, *dBestIdx = nullptr, *dXaCount = nullptr, *dXaOffsets = nullptr;
    uint32_t *dXaStart = nullptr, *dXaCountOut = nullptr, *dXaOutOffsets = nullptr, *dOccCountIn = nullptr, *dSegBegin = nullptr, *dSegEnd = nullptr;
    DevPairXa *dXa = nullptr, *dXaOut = nullptr;
    uint32_t xaCap = 0, xaUsed = 0;
    uint64_t *dSortKeys[2] = {nullptr, nullptr};
    uint32_t *dSortMeta = nullptr;
    char     *dSortTemp = nullptr;
    size_t    sortItems = 0, sortTempBytes = 0;
    std::vector<void *> allocs;
    // state of the last call
    int lastKind = 0;              // 1 = search, 2 = align (single), 3 = align (pairs)
    uint32_t lastNumCases = 0;
    bool counting = false;
    hipEvent_t evTotal[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> evSearch;
    uint32_t searchLaunches = 0;
    uint32_t occUsed = 0;
    uint64_t recordsTotal = 0;      // records of all search passes of the last call (read back when counters are enabled)
};

template <class T>
static int balloc(s3_batch *b, T **p, size_t count)
{
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    if (!bytes) bytes = 16;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(S3_ENOMEM, std::string("hipMalloc(batch): ") + hipGetErrorString(e));
    b->allocs.push_back(q);
    *p = (T *)q;
    return S3_OK;
}

// replace a batch buffer by a larger one, keeping the first `keep` elements
template <class T>
static int bgrow(s3_batch *b, T **p, size_t newCount, size_t keep)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, newCount * sizeof(T));
    if (e != hipSuccess) return fail(S3_ENOMEM, std::string("hipMalloc(batch grow): ") + hipGetErrorString(e));
    if (keep) {
        e = hipMemcpyAsync(q, *p, keep * sizeof(T), hipMemcpyDeviceToDevice, b->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        if (e != hipSuccess) { (void)hipFree(q); return fail(S3_EHIP, std::string("grow copy: ") + hipGetErrorString(e)); }
    }
    for (auto &a : b->allocs) if (a == (void *)*p) a = q;
    (void)hipFree(*p);
    *p = (T *)q;
    return S3_OK;
}

static const size_t kMaxEntries = 0xFFFFFF00ull;         // record / occurrence indices are 32 bit

static int growRecords(s3_batch *b)
{
    if (b->recCap >= kMaxEntries) return fail(S3_EOVERFLOW, "record buffer overflow at the 32-bit limit");
    size_t cap = (size_t)b->recCap * 2;
    if (cap > kMaxEntries) cap = kMaxEntries;
    int rc;
    if ((rc = bgrow(b, &b->dRecs, cap, 0))) return rc;
    if ((rc = bgrow(b, &b->dOrdered, cap, 0))) return rc;
    if ((rc = bgrow(b, &b->dHits, cap, 0))) return rc;
    b->recCap = (uint32_t)cap;
    return S3_OK;
}

static int growOccurrences(s3_batch *b, uint64_t need)
{
    if (need > kMaxEntries) return fail(S3_EOVERFLOW, "occurrence buffer overflow at the 32-bit limit");
    size_t cap = (size_t)b->occCap * 2;
    if (cap < need + need / 4) cap = need + need / 4;
    if (cap > kMaxEntries) cap = kMaxEntries;
    int rc;
    if ((rc = bgrow(b, &b->dOccPos, cap, b->occUsed))) return rc;
    if ((rc = bgrow(b, &b->dOccMeta, cap, b->occUsed))) return rc;
    if ((rc = bgrow(b, &b->dOutPos, cap, 0))) return rc;
    if ((rc = bgrow(b, &b->dOutMeta, cap, 0))) return rc;
    b->occCap = (uint32_t)cap;
    return S3_OK;
}

static size_t envSize(const char *name, size_t dflt)
{
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    return (size_t)strtoull(v, nullptr, 10);
}

extern "C" int s3_batch_create(s3_index *ix, uint32_t maxReads, uint32_t wordPerQuery, s3_batch **outp)
{
    if (!ix || !outp || maxReads == 0 || wordPerQuery == 0 || wordPerQuery > 64) return fail(S3_EINVAL, "s3_batch_create: bad argument");
    HIPCHK(hipSetDevice(ix->dev));
    s3_batch *b = new s3_batch();
    b->ix = ix; b->maxReads = maxReads; b->wpq = wordPerQuery;
    const size_t roundUp = ((size_t)maxReads + 31) / 32 * 32;
    b->maxItems = (size_t)maxReads * 20;                               // 10 cases x 2 strands at 4 mismatches
    const size_t recPerRead = envSize("S3DP_REC_PER_READ", 6), occPerRead = envSize("S3DP_OCC_PER_READ", 12);
    // every wave of the persistent search grid may hold one partly filled block of record slots: that is the floor.  Both buffers
    // grow on demand (growRecords / growOccurrences); the sizes here only avoid regrowth in the common case.
    const size_t recFloor = envSize("S3DP_REC_FLOOR", 65536 + (size_t)S3_V2_REC_BLOCK * 8192);
    size_t recCap = (size_t)maxReads * recPerRead + recFloor, occCap = (size_t)maxReads * occPerRead + 65536;
    if (recCap < 4096) recCap = 4096;
    if (recCap > 0xFFFFFF00ull) recCap = 0xFFFFFF00ull;
    if (occCap > 0xFFFFFF00ull) occCap = 0xFFFFFF00ull;
    b->recCap = (uint32_t)recCap; b->occCap = (uint32_t)occCap;
    HIPCHK(hipStreamCreate(&b->stream));
    int rc;
#define BA(p, cnt) if ((rc = balloc(b, &b->p, (cnt)))) { s3_batch_free(b); return rc; }
    BA(dQueries, roundUp * wordPerQuery); BA(dLens, roundUp); BA(dSeedLens, roundUp); BA(dReadsT, roundUp * 2 * wordPerQuery + 16);
    BA(dRecs, recCap); BA(dOrdered, recCap);
    BA(dRecCount, 4); BA(dFlags, 4); BA(dCounters, 8); BA(dWorkHead, 2);
    BA(dItemCounts, b->maxItems + 1); BA(dItemOffsets, b->maxItems + 1);
    BA(dKept, (size_t)maxReads + 1); BA(dOccCounts, (size_t)maxReads + 1); BA(dCeCounts, (size_t)maxReads + 1); BA(dTotalOcc, (size_t)maxReads + 1);
    BA(dWithError, (size_t)maxReads * 8);
    BA(dHitOffsets, (size_t)maxReads + 1); BA(dHits, recCap);
    BA(dActive[0], maxReads); BA(dActive[1], maxReads); BA(dNumSel, 4); BA(dIota, maxReads);
    BA(dState, maxReads); BA(dSelFlags, maxReads);
    BA(dOccOffsets, (size_t)maxReads + 1); BA(dMasked, (size_t)maxReads + 1);
    BA(dOccPos, occCap); BA(dOccMeta, occCap);
    BA(dPhase, maxReads); BA(dFinalWithError, (size_t)maxReads * 8); BA(dOccStart, maxReads); BA(dOccCount, (size_t)maxReads + 1);
    BA(dOutOffsets, (size_t)maxReads + 1); BA(dOutPos, occCap); BA(dOutMeta, occCap);
#undef BA
    // scan / select scratch sized for the largest problem
    size_t t1 = 0, t2 = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t1, b->dItemCounts, b->dItemOffsets, (int)(b->maxItems + 1), b->stream);
    (void)hipcub::DeviceSelect::Flagged(nullptr, t2, b->dIota, b->dSelFlags, b->dActive[0], b->dNumSel, (int)maxReads, b->stream);
    b->tempBytes = (t1 > t2 ? t1 : t2) + 256;
    if ((rc = balloc(b, &b->dTemp, b->tempBytes))) { s3_batch_free(b); return rc; }
    HIPCHK(hipEventCreate(&b->evTotal[0])); HIPCHK(hipEventCreate(&b->evTotal[1]));
    b->evSearch.resize(16);
    for (auto &e : b->evSearch) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipMemsetAsync(b->dCounters, 0, 64, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    *outp = b;
    return S3_OK;
}

extern "C" void s3_batch_free(s3_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->ix->dev);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (void *p : b->allocs) (void)hipFree(p);
    for (auto e : b->evSearch) if (e) (void)hipEventDestroy(e);
    for (auto e : b->evTotal) if (e) (void)hipEventDestroy(e);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

extern "C" int s3_batch_upload(s3_batch *b, const uint32_t *queries, const uint32_t *readLengths, uint32_t n)
{
    if (!b || !queries || !readLengths) return fail(S3_EINVAL, "s3_batch_upload: null argument");
    if (n > b->maxReads) return fail(S3_EINVAL, "s3_batch_upload: n exceeds batch capacity");
    for (uint32_t i = 0; i < n; i++)
        if (readLengths[i] < 16 || readLengths[i] > S3_MAX_READ_LENGTH || readLengths[i] > b->wpq * 16u)
            return fail(S3_EINVAL, "s3_batch_upload: read length out of range for wordPerQuery");
    HIPCHK(hipSetDevice(b->ix->dev));
    const size_t roundUp = ((size_t)n + 31) / 32 * 32;
    HIPCHK(hipMemcpyAsync(b->dQueries, queries, roundUp * b->wpq * 4, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->dLens, readLengths, (size_t)n * 4, hipMemcpyHostToDevice, b->stream));
    launchOrientReads(b->dQueries, b->dLens, b->dReadsT, n, b->wpq, b->wpq, b->stream);
    HIPCHK(hipStreamSynchronize(b->stream));
    b->n = n;
    b->lastKind = 0;
    b->uniformLen = n ? readLengths[0] : 0;
    for (uint32_t i = 1; i < n; i++) if (readLengths[i] != readLengths[0]) { b->uniformLen = 0; break; }
    b->fullUniformLen = b->uniformLen;
    b->longMode = false;
    return S3_OK;
}

// Long-read mode: the seedLengths argument of all_valid_alignment (alignment.cu:534; built at :2473-2491).  The search then runs on
// the first seedLengths[i] bases of read i and the ladder verifies the remaining bases (validateAlignments, CPUfunctions.cpp:1129).
extern "C" int s3_batch_set_seed_lengths(s3_batch *b, const uint32_t *seedLengths)
{
    if (!b) return fail(S3_EINVAL, "null batch");
    HIPCHK(hipSetDevice(b->ix->dev));
    if (!seedLengths) {
        if (b->longMode) {
            launchOrientReads(b->dQueries, b->dLens, b->dReadsT, b->n, b->wpq, b->wpq, b->stream);
            HIPCHK(hipStreamSynchronize(b->stream));
        }
        b->longMode = false; b->uniformLen = b->fullUniformLen;
        return S3_OK;
    }
    const uint32_t n = b->n;
    std::vector<uint32_t> lens(n);
    HIPCHK(hipMemcpy(lens.data(), b->dLens, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++)
        if (seedLengths[i] < 16 || seedLengths[i] > lens[i] || seedLengths[i] > 128)
            return fail(S3_EINVAL, "s3_batch_set_seed_lengths: a seed must have 16..128 bases and not exceed its read");
    HIPCHK(hipMemcpyAsync(b->dSeedLens, seedLengths, (size_t)n * 4, hipMemcpyHostToDevice, b->stream));
    launchOrientReads(b->dQueries, b->dSeedLens, b->dReadsT, n, b->wpq, b->wpq > 8 ? 8u : b->wpq, b->stream);   // seeds have at most 128 bases
    HIPCHK(hipStreamSynchronize(b->stream));
    b->uniformLen = n ? seedLengths[0] : 0;
    for (uint32_t i = 1; i < n; i++) if (seedLengths[i] != seedLengths[0]) { b->uniformLen = 0; break; }
    b->longMode = true;
    b->lastKind = 0;
    return S3_OK;
}

extern "C" int s3_batch_enable_counters(s3_batch *b, int on)
{
    if (!b) return fail(S3_EINVAL, "null batch");
    b->counting = on != 0;
    return S3_OK;
}

// search + order + merge for `nActive` slots.  Leaves per-slot results in dTotalOcc / dWithError / dKept / ...
static int searchPass(s3_batch *b, const uint32_t *active, uint32_t nActive, uint32_t modelMm, uint32_t maxOut)
{
    const int nCases = s3_num_cases((int)modelMm);
    const size_t nItems = (size_t)nActive * nCases * 2;
    SearchIn in{};
    in.queries = b->dQueries; in.readsT = b->dReadsT; in.readLengths = b->longMode ? b->dSeedLens : b->dLens; in.active = active; in.nActive = nActive;
    in.wpq = b->wpq;
    in.wpqLoad = (b->longMode && b->wpq > 8) ? 8 : b->wpq;                // seeds have at most 128 bases
    in.models = b->ix->models[modelMm]; in.nCases = nCases; in.maxOut = maxOut; in.uniformLen = b->uniformLen;
    {
        S3Model m;
        s3_model_build(100, (int)modelMm, &m);
        in.nStepsPacked = 0;
        for (int c = 0; c < m.nCases; c++) in.nStepsPacked |= (uint32_t)m.cs[c].nSteps << (3 * c);
    }
    unsigned long long counterSnap[8];
    if (b->counting) {
        HIPCHK(hipMemcpyAsync(counterSnap, b->dCounters, 64, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    for (;;) {
        HIPCHK(hipMemsetAsync(b->dRecCount, 0, 4, b->stream));
        SearchOut out{};
        out.recs = b->dRecs; out.recCap = b->recCap; out.recCount = b->dRecCount; out.itemCounts = b->dItemCounts; out.flags = b->dFlags;
        out.counters = b->counting ? b->dCounters : nullptr;
        out.workHead = b->dWorkHead;
        HIPCHK(hipMemsetAsync(b->dWorkHead, 0, 8, b->stream));
        if (b->searchLaunches * 2 + 1 < b->evSearch.size()) HIPCHK(hipEventRecord(b->evSearch[b->searchLaunches * 2], b->stream));
        {
            // S3DP_KERNEL=v1 selects the first-generation one-thread-per-item kernel (kept for A/B measurements)
            static const bool useV1 = [] { const char *e = getenv("S3DP_KERNEL"); return e && strcmp(e, "v1") == 0; }();
            static const unsigned blocksPerCu = (unsigned)envSize("S3DP_BLOCKS_PER_CU", S3_V2_WAVES_PER_SIMD);
            if (useV1) launchSearch(b->ix->d, in, out, 256u * 8u * 4u, b->stream);
            else launchSearchV2(b->ix->d, in, out, 256u * blocksPerCu, b->stream);
        }
        if (b->searchLaunches * 2 + 1 < b->evSearch.size()) HIPCHK(hipEventRecord(b->evSearch[b->searchLaunches * 2 + 1], b->stream));
        // a record that did not fit was dropped but counted: the ordering below must not run on such a pass.  Grow and search again.
        uint32_t flags = 0;
        HIPCHK(hipMemcpyAsync(&flags, b->dFlags, 4, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        if (!(flags & S3_FLAG_REC_OVERFLOW)) break;
        int rc = growRecords(b);
        if (rc) return rc;
        flags &= ~S3_FLAG_REC_OVERFLOW;
        HIPCHK(hipMemcpyAsync(b->dFlags, &flags, 4, hipMemcpyHostToDevice, b->stream));
        if (b->counting) HIPCHK(hipMemcpyAsync(b->dCounters, counterSnap, 64, hipMemcpyHostToDevice, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    b->searchLaunches++;
    HIPCHK(hipMemsetAsync(b->dItemCounts + nItems, 0, 4, b->stream));
    size_t tb = b->tempBytes;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->dTemp, tb, b->dItemCounts, b->dItemOffsets, (int)(nItems + 1), b->stream));
    launchScatter(b->dRecs, b->dRecCount, b->recCap, b->dItemOffsets, b->dOrdered, b->stream);
    MergeArgs m{};
    m.nActive = nActive; m.nCases = (uint32_t)nCases; m.maxOut = maxOut; m.itemOffsets = b->dItemOffsets; m.ordered = b->dOrdered;
    m.keptRecs = b->dKept; m.occCounts = b->dOccCounts; m.ceCounts = b->dCeCounts; m.totalOcc = b->dTotalOcc; m.withError = b->dWithError;
    launchMergeCount(m, b->stream);
    b->lastNumCases = (uint32_t)nCases;
    if (b->counting) {
        uint32_t t = 0;
        HIPCHK(hipMemcpyAsync(&t, b->dItemOffsets + nItems, 4, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        b->recordsTotal += t;
    }
    HIPCHK(hipGetLastError());
    return S3_OK;
}

static int checkFlags(s3_batch *b)
{
    uint32_t flags = 0;
    HIPCHK(hipMemcpyAsync(&flags, b->dFlags, 4, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (flags & S3_FLAG_STACK_OVERFLOW) return fail(S3_EOVERFLOW, "search stack overflow (internal bound exceeded)");
    if (flags & S3_FLAG_REC_OVERFLOW) return fail(S3_EOVERFLOW, "record buffer overflow: raise S3DP_REC_PER_READ");
    if (flags & S3_FLAG_OCC_OVERFLOW) return fail(S3_EOVERFLOW, "occurrence buffer overflow: raise S3DP_OCC_PER_READ");
    return S3_OK;
}

extern "C" int s3_batch_search(s3_batch *b, uint32_t numMismatch, uint32_t maxOutputPerRead)
{
    if (!b) return fail(S3_EINVAL, "null batch");
    if (numMismatch > S3_MAX_MISMATCH) return fail(S3_EINVAL, "numMismatch must be 0..4");
    if (maxOutputPerRead == 0 || maxOutputPerRead > 0xFFFFFFu) return fail(S3_EINVAL, "maxOutputPerRead must be 1..16777215");
    HIPCHK(hipSetDevice(b->ix->dev));
    b->searchLaunches = 0;
    b->recordsTotal = 0;
    HIPCHK(hipMemsetAsync(b->dFlags, 0, 4, b->stream));
    if (b->counting) HIPCHK(hipMemsetAsync(b->dCounters, 0, 64, b->stream));
    HIPCHK(hipEventRecord(b->evTotal[0], b->stream));
    int rc = searchPass(b, nullptr, b->n, numMismatch, maxOutputPerRead);
    if (rc) return rc;
    // hits in the ABI format
    HIPCHK(hipMemsetAsync(b->dKept + b->n, 0, 4, b->stream));
    size_t tb = b->tempBytes;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->dTemp, tb, b->dKept, b->dHitOffsets, (int)(b->n + 1), b->stream));
    MergeArgs m{};
    m.nActive = b->n; m.nCases = b->lastNumCases; m.maxOut = maxOutputPerRead; m.itemOffsets = b->dItemOffsets; m.ordered = b->dOrdered;
    m.keptRecs = b->dKept; m.occCounts = b->dOccCounts; m.ceCounts = b->dCeCounts; m.totalOcc = b->dTotalOcc; m.withError = b->dWithError;
    launchMergeFill(m, b->dHitOffsets, b->dHits, b->stream);
    HIPCHK(hipEventRecord(b->evTotal[1], b->stream));
    b->lastKind = 1;
    return S3_OK;
}

extern "C" int s3_batch_sync(s3_batch *b)
{
    if (!b) return fail(S3_EINVAL, "null batch");
    HIPCHK(hipSetDevice(b->ix->dev));
    HIPCHK(hipStreamSynchronize(b->stream));
    return checkFlags(b);
}

extern "C" int s3_batch_fetch_hits(s3_batch *b, uint64_t *hitOffsets, s3_hit *hits, size_t hitCap, uint32_t *totalOcc, uint32_t *withError,
                                   uint64_t *nHitsOut)
{
    if (!b || b->lastKind != 1) return fail(S3_EINVAL, "s3_batch_fetch_hits: no search result pending");
    int rc = s3_batch_sync(b);
    if (rc) return rc;
    const uint32_t n = b->n;
    std::vector<uint32_t> off((size_t)n + 1);
    HIPCHK(hipMemcpy(off.data(), b->dHitOffsets, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
    const uint64_t total = off[n];
    if (nHitsOut) *nHitsOut = total;
    if (hitOffsets) for (size_t i = 0; i <= n; i++) hitOffsets[i] = off[i];
    if (hits) {
        if (total > hitCap) return fail(S3_EOVERFLOW, "s3_batch_fetch_hits: hit buffer too small");
        static_assert(sizeof(s3_hit) == sizeof(DevHit), "hit layout");
        HIPCHK(hipMemcpy(hits, b->dHits, total * sizeof(DevHit), hipMemcpyDeviceToHost));
    }
    if (totalOcc) HIPCHK(hipMemcpy(totalOcc, b->dTotalOcc, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (withError) HIPCHK(hipMemcpy(withError, b->dWithError, (size_t)n * 32, hipMemcpyDeviceToHost));
    return S3_OK;
}

// ---------------------------------------------------------------------------------------------
// small helper kernels of the ladder (plumbing, not hot)
// ---------------------------------------------------------------------------------------------
__global__ void iotaKernel(uint32_t *p, uint32_t n) { uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = i; }

__global__ void maskKernel(uint32_t n, const uint8_t *state, uint8_t want, const uint32_t *occCounts, uint32_t *masked, uint8_t *selFlags,
                           uint8_t selectWhen /*0: state!=1, 2: state==2*/)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { masked[i] = 0; return; }
    masked[i] = state[i] == want ? occCounts[i] : 0u;
    selFlags[i] = selectWhen == 0 ? (state[i] != 1) : (state[i] == 2);
}

// record the resolved reads of this pass in the per-read (global index) result arrays
__global__ void publishKernel(uint32_t nActive, const uint32_t *active, const uint8_t *state, uint8_t want, uint8_t phase, const uint32_t *withError,
                              const uint32_t *occOffsets, uint32_t occBase, uint8_t *outPhase, uint32_t *outWithError, uint32_t *outOccStart,
                              uint32_t *outOccCount)
{
    uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nActive || state[slot] != want) return;
    const uint32_t r = active ? active[slot] : slot;
    outPhase[r] = phase;
    for (int k = 0; k < 8; k++) outWithError[(size_t)r * 8 + k] = withError[(size_t)slot * 8 + k];
    outOccStart[r] = occBase + occOffsets[slot];
    outOccCount[r] = occOffsets[slot + 1] - occOffsets[slot];
}

__global__ void stateAllKernel(uint32_t n, const uint32_t *totalOcc, uint8_t *state)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) state[i] = totalOcc[i] > 0 ? 1 : 0;
}

__global__ void gatherOccKernel(uint32_t n, const uint32_t *occStart, const uint32_t *outOffsets, const uint32_t *pos, const uint32_t *meta,
                                uint32_t *outPos, uint32_t *outMeta)
{
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint32_t cnt = outOffsets[r + 1] - outOffsets[r], src = occStart[r], dst = outOffsets[r];
    for (uint32_t k = 0; k < cnt; k++) { outPos[dst + k] = pos[src + k]; outMeta[dst + k] = meta[src + k]; }
}

// Long-read mode (validateAlignments, CPUfunctions.cpp:1129-1222, called at :1832): the published occurrences are seed hits; every one
// is extended over the rest of the read by counting mismatching characters against the packed text (numMismatchNew, PE.cpp:148),
// kept when seed + extension mismatches <= maxMismatch and moved to the leftmost text position of the whole read; best modes keep
// only the hits with the fewest mismatches.  The list is filtered in place in the reference's order, one lane per read.  A read whose
// list comes out empty is "nohit": unaligned, and carried to the next phase when there is one.
struct ValidateArgs {
    uint32_t nActive;
    const uint32_t *active;
    const uint8_t *state;
    uint8_t want;
    const uint32_t *occOffsets;
    uint32_t occBase;
    uint32_t *occPos, *occMeta;
    const uint32_t *queries;      // packed reads, reference layout (32-interleaved, LSB-first characters)
    uint32_t wpq;
    const uint32_t *lens, *seedLens;
    const uint32_t *text;
    uint32_t textLength;
    int onlyBest, minSeedMis, needMapq;
    uint32_t maxHit;
    int carryOn;                  // not the terminal phase: empty lists go on
    uint8_t *outPhase;
    uint32_t *outOccCount;
    uint8_t *selFlags;
};

__global__ void validateKernel(ValidateArgs a)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= a.nActive || a.state[slot] != a.want) return;
    const uint32_t r = a.active ? a.active[slot] : slot;
    const uint32_t readLen = a.lens[r], seedLen = a.seedLens[r];
    const uint32_t base = a.occBase + a.occOffsets[slot];
    const uint32_t cnt = a.occOffsets[slot + 1] - a.occOffsets[slot];
    uint32_t out = cnt;
    if (readLen > seedLen) {
        const uint32_t ext = readLen - seedLen;
        int maxMis = (int)ceil(0.02 * (double)readLen);                  // MISMATCH_RATIO_FOR_LONG, CPUfunctions.cpp:1819
        if (a.needMapq) maxMis *= 2;
        int pre = maxMis;
        const uint32_t *rd = a.queries + (size_t)(r >> 5) * 32u * a.wpq + (r & 31u);
        auto readChar = [&](uint32_t q) -> uint32_t { return (rd[(size_t)(q >> 4) * 32u] >> ((q & 15u) * 2u)) & 3u; };
        auto textChar = [&](uint64_t p) -> uint32_t { return (a.text[p >> 4] >> (2u * (15u - (uint32_t)(p & 15u)))) & 3u; };
        out = 0;
        for (uint32_t i = 0; i < cnt; i++) {
            const uint32_t pos = a.occPos[base + i], meta = a.occMeta[base + i];
            const uint32_t strand = meta & 0xFFu;
            const int seedMis = (int)((meta >> 8) & 0xFFu);
            if (seedMis < a.minSeedMis) continue;
            int mis = maxMis + 1;
            if (strand == 1u && (uint64_t)pos + readLen <= a.textLength) {
                mis = 0;
                for (uint32_t t = 0; t < ext; t++) mis += readChar(seedLen + t) != textChar((uint64_t)pos + seedLen + t);
            }
            if (strand == 2u && pos >= ext) {
                mis = 0;
                for (uint32_t t = 0; t < ext; t++) mis += (3u - readChar(readLen - 1u - t)) != textChar((uint64_t)pos - ext + t);
            }
            const int tot = mis + seedMis;
            if (tot > maxMis) continue;
            if (a.onlyBest && tot > pre) continue;
            if (a.onlyBest && tot < pre) { out = 0; pre = tot; }
            a.occPos[base + out] = strand == 1u ? pos : pos - ext;
            a.occMeta[base + out] = strand | (uint32_t)tot << 8;
            out++;
            if (pre == a.minSeedMis && out >= a.maxHit) break;
        }
    }
    if (out > a.maxHit) out = a.maxHit;                                  // origHitLimit, CPUfunctions.cpp:1838
    a.outOccCount[r] = out;
    if (out == 0) {
        a.outPhase[r] = 0;
        if (a.carryOn) a.selFlags[slot] = 1;
    }
}

// decode + publish the slots whose state == want; the occurrence region is appended at *dOccBase
struct LongArgs { bool on; int onlyBest, minSeedMis, needMapq; uint32_t maxHit; int carryOn; };

static int decodeAndPublish(s3_batch *b, const uint32_t *active, uint32_t nActive, uint8_t want, uint8_t phase, uint32_t maxOut, uint8_t selectWhen,
                            const LongArgs &lg)
{
    if (!nActive) return S3_OK;
    const unsigned g = (nActive + 1 + 255) / 256;
    hipLaunchKernelGGL(maskKernel, dim3(g), dim3(256), 0, b->stream, nActive, b->dState, want, b->dOccCounts, b->dMasked, b->dSelFlags, selectWhen);
    size_t tb = b->tempBytes;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->dTemp, tb, b->dMasked, b->dOccOffsets, (int)(nActive + 1), b->stream));
    uint32_t added = 0;
    HIPCHK(hipMemcpyAsync(&added, b->dOccOffsets + nActive, 4, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    if ((uint64_t)b->occUsed + added > b->occCap) {
        int rc = growOccurrences(b, (uint64_t)b->occUsed + added);
        if (rc) return rc;
    }
    const uint32_t base = b->occUsed;
    hipLaunchKernelGGL(publishKernel, dim3((nActive + 255) / 256), dim3(256), 0, b->stream, nActive, active, b->dState, want, phase, b->dWithError,
                       b->dOccOffsets, base, b->dPhase, b->dFinalWithError, b->dOccStart, b->dOccCount);
    MergeArgs m{};
    m.nActive = nActive; m.nCases = b->lastNumCases; m.maxOut = maxOut; m.itemOffsets = b->dItemOffsets; m.ordered = b->dOrdered;
    m.keptRecs = b->dKept; m.occCounts = b->dOccCounts; m.ceCounts = b->dCeCounts; m.totalOcc = b->dTotalOcc; m.withError = b->dWithError;
    m.decodeMask = b->dState; m.decodeWant = want;
    launchDecode(m, b->ix->d, b->dOccOffsets, base, b->occCap, b->dOccPos, b->dOccMeta, b->dFlags, b->stream);
    if (lg.on) {
        ValidateArgs v{};
        v.nActive = nActive; v.active = active; v.state = b->dState; v.want = want; v.occOffsets = b->dOccOffsets; v.occBase = base;
        v.occPos = b->dOccPos; v.occMeta = b->dOccMeta; v.queries = b->dQueries; v.wpq = b->wpq; v.lens = b->dLens; v.seedLens = b->dSeedLens;
        v.text = b->ix->d.pac; v.textLength = b->ix->d.n;
        v.onlyBest = lg.onlyBest; v.minSeedMis = lg.minSeedMis; v.needMapq = lg.needMapq; v.maxHit = lg.maxHit; v.carryOn = lg.carryOn;
        v.outPhase = b->dPhase; v.outOccCount = b->dOccCount; v.selFlags = b->dSelFlags;
        hipLaunchKernelGGL(validateKernel, dim3((nActive + 255) / 256), dim3(256), 0, b->stream, v);
    }
    b->occUsed += added;
    return S3_OK;
}

// reportType as the reference's -h option: 1 all valid, 2 all best, 3 unique best, 4 random best.
//   all best   : all_best_single_alignment (alignment.cu:1422): phases 1..k, reads whose best == phase are deferred / re-searched
//   all valid  : one phase at k (soap3_dp_single_align -> all_valid_alignment, alignment.cu:2530), terminal, same MAPQ special case
//   unique/random best: best_single_alignment (alignment.cu:1377): phases 0..k, a read is reported in the first phase with a hit
extern "C" int s3_batch_align_single(s3_batch *b, uint32_t numMismatch, uint32_t maxOutputPerRead, int reportType)
{
    if (reportType < 1 || reportType > 4) return fail(S3_EINVAL, "reportType must be 1..4");
    const bool bestMode = reportType >= 3, allValid = reportType == 1;
    if (!b) return fail(S3_EINVAL, "null batch");
    if (numMismatch > S3_MAX_MISMATCH) return fail(S3_EINVAL, "numMismatch must be 0..4");
    if (maxOutputPerRead == 0 || maxOutputPerRead > 0xFFFFFFu) return fail(S3_EINVAL, "maxOutputPerRead must be 1..16777215");
    HIPCHK(hipSetDevice(b->ix->dev));
    const uint32_t n = b->n;
    b->searchLaunches = 0;
    b->occUsed = 0;
    b->recordsTotal = 0;
    HIPCHK(hipMemsetAsync(b->dFlags, 0, 4, b->stream));
    if (b->counting) HIPCHK(hipMemsetAsync(b->dCounters, 0, 64, b->stream));
    HIPCHK(hipEventRecord(b->evTotal[0], b->stream));
    HIPCHK(hipMemsetAsync(b->dPhase, 0, n, b->stream));
    HIPCHK(hipMemsetAsync(b->dFinalWithError, 0, (size_t)n * 32, b->stream));
    HIPCHK(hipMemsetAsync(b->dOccStart, 0, (size_t)n * 4, b->stream));
    HIPCHK(hipMemsetAsync(b->dOccCount, 0, ((size_t)n + 1) * 4, b->stream));
    HIPCHK(hipMemsetAsync(b->dNumSel, 0, 16, b->stream));
    const uint32_t last = (bestMode || allValid) ? numMismatch : (numMismatch > 0 ? numMismatch : 1);   // alignment.cu:1439
    const uint32_t first = bestMode ? 0u : (allValid ? numMismatch : 1u);
    const uint32_t searchLimit = b->longMode ? 0xFFFFFFFFu : maxOutputPerRead;
    const uint32_t *active = nullptr;
    uint32_t nActive = n;
    int pp = 0;
    for (uint32_t mm = first; mm <= last && nActive > 0; mm++) {
        const bool terminal = mm >= numMismatch;
        // long-read mode lifts the hit limit during the search (CPUfunctions.cpp:1655-1656) and applies it after the verification
        LongArgs lg{b->longMode, bestMode ? 1 : 0, bestMode ? (int)mm : 0, bestMode ? 0 : 1, maxOutputPerRead, terminal ? 0 : 1};
        int rc = searchPass(b, active, nActive, mm, searchLimit);
        if (rc) return rc;
        // MAPQ special case needs the (mm+1) model, which exists only for mm < 4 (CPUfunctions.cpp:1756)
        launchClassify(nActive, b->dTotalOcc, b->dWithError, mm, (!bestMode && mm < 4) ? 1u : 0u, b->dState, b->stream);
        rc = decodeAndPublish(b, active, nActive, 1, (uint8_t)(bestMode ? mm + 1 : mm), searchLimit, terminal ? 2 : 0, lg);
        if (rc) return rc;
        // reads that go on: no hit or best == mm (non terminal: next phase; terminal: state 2 re-searched with mm+1 model)
        const uint32_t *idsIn = active;
        if (!idsIn) {
            hipLaunchKernelGGL(iotaKernel, dim3((nActive + 255) / 256), dim3(256), 0, b->stream, b->dIota, nActive);
            idsIn = b->dIota;
        }
        size_t tb = b->tempBytes;
        uint32_t *nextList = b->dActive[pp];
        HIPCHK(hipcub::DeviceSelect::Flagged(b->dTemp, tb, idsIn, b->dSelFlags, nextList, b->dNumSel, (int)nActive, b->stream));
        uint32_t nNext = 0;
        HIPCHK(hipMemcpyAsync(&nNext, b->dNumSel, 4, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        if (terminal) {
            if (!bestMode && mm < 4 && nNext > 0) {
                // ProcessOneMoreMismatchAllCases (CPUfunctions.cpp:624, called at :1790): the read is re-searched with the
                // (mm+1)-mismatch model and THAT result is what gets reported
                rc = searchPass(b, nextList, nNext, mm + 1, searchLimit);
                if (rc) return rc;
                hipLaunchKernelGGL(stateAllKernel, dim3((nNext + 255) / 256), dim3(256), 0, b->stream, nNext, b->dTotalOcc, b->dState);
                rc = decodeAndPublish(b, nextList, nNext, 1, (uint8_t)mm, searchLimit, 0, lg);
                if (rc) return rc;
            }
            break;
        }
        active = nextList;
        nActive = nNext;
        pp ^= 1;
    }
    // read-ordered CSR of the occurrences
    size_t tb = b->tempBytes;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->dTemp, tb, b->dOccCount, b->dOutOffsets, (int)(n + 1), b->stream));
    if (n) hipLaunchKernelGGL(gatherOccKernel, dim3((n + 255) / 256), dim3(256), 0, b->stream, n, b->dOccStart, b->dOutOffsets, b->dOccPos, b->dOccMeta,
                              b->dOutPos, b->dOutMeta);
    HIPCHK(hipEventRecord(b->evTotal[1], b->stream));
    HIPCHK(hipGetLastError());
    b->lastKind = 2;
    return S3_OK;
}

extern "C" int s3_batch_align_single_allbest(s3_batch *b, uint32_t numMismatch, uint32_t maxOutputPerRead)
{
    return s3_batch_align_single(b, numMismatch, maxOutputPerRead, 2);
}

extern "C" int s3_batch_fetch_alignments(s3_batch *b, uint8_t *phase, uint32_t *withError, uint64_t *occOffsets, uint32_t *occPos, uint8_t *occStrand,
                                         uint8_t *occMis, size_t occCap, uint64_t *nOccOut)
{
    if (!b || b->lastKind != 2) return fail(S3_EINVAL, "s3_batch_fetch_alignments: no alignment result pending");
    int rc = s3_batch_sync(b);
    if (rc) return rc;
    const uint32_t n = b->n;
    std::vector<uint32_t> off((size_t)n + 1);
    HIPCHK(hipMemcpy(off.data(), b->dOutOffsets, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
    const uint64_t total = off[n];
    if (nOccOut) *nOccOut = total;
    if (phase) HIPCHK(hipMemcpy(phase, b->dPhase, n, hipMemcpyDeviceToHost));
    if (withError) HIPCHK(hipMemcpy(withError, b->dFinalWithError, (size_t)n * 32, hipMemcpyDeviceToHost));
    if (occOffsets) for (size_t i = 0; i <= n; i++) occOffsets[i] = off[i];
    if (occPos || occStrand || occMis) {
        if (total > occCap) return fail(S3_EOVERFLOW, "s3_batch_fetch_alignments: output arrays too small");
        if (occPos) HIPCHK(hipMemcpy(occPos, b->dOutPos, total * 4, hipMemcpyDeviceToHost));
        if (occStrand || occMis) {
            std::vector<uint32_t> meta(total);
            HIPCHK(hipMemcpy(meta.data(), b->dOutMeta, total * 4, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < total; i++) {
                if (occStrand) occStrand[i] = (uint8_t)(meta[i] & 0xFF);
                if (occMis) occMis[i] = (uint8_t)((meta[i] >> 8) & 0xFF);
            }
        }
    }
    return S3_OK;
}

// ---------------------------------------------------------------------------------------------
// paired-end ladder with the DP stages off.  Replaces, for `soap3-dp pair ... -s k`:
//   soap3_dp_pair_align's choice of ladder (alignment.cu:1965-2013), all_best_alignment (:1236, phases 1, 2, k),
//   four_phases_alignment (:1119, phases 0, 1, 2, k), all_valid_alignment for pairs (:534), the PAIR_END_READ branch of hostKernel
//   (CPUfunctions.cpp:1980-2560) and PEMappingOccurrences / PEStatsPEOutput (PEAlgnmt.cpp:480,834).
// ---------------------------------------------------------------------------------------------
__global__ void pairInitKernel(uint32_t n, DevPair *pairs)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    DevPair z{};
    z.min1 = z.min2 = 255; z.minTotal = z.secMinTotal = 127;
    pairs[p] = z;
}

static int ensurePairBuffers(s3_batch *b)
{
    if (b->pairReady) return S3_OK;
    const size_t maxReads = b->maxReads, maxPairs = maxReads / 2 + 1;
    int rc;
#define BA(p, cnt) if ((rc = balloc(b, &b->p, (cnt)))) return rc;
    BA(dPairs, maxPairs); BA(dPairMin, maxReads + 2); BA(dPairNext, maxPairs); BA(dReadList, maxReads + 2);
    BA(dPairActive[0], maxPairs); BA(dPairActive[1], maxPairs); BA(dBestIdx, maxPairs); BA(dXaCount, maxPairs + 1); BA(dXaOffsets, maxPairs + 1);
    BA(dXaStart, maxPairs); BA(dXaCountOut, maxPairs + 1); BA(dXaOutOffsets, maxPairs + 1); BA(dOccCountIn, maxReads + 2);
    BA(dSegBegin, maxReads + 2); BA(dSegEnd, maxReads + 2);
    b->xaCap = (uint32_t)(maxPairs + 65536);
    BA(dXa, b->xaCap); BA(dXaOut, b->xaCap);
#undef BA
    b->pairReady = true;
    return S3_OK;
}

static int ensureSortBuffers(s3_batch *b, size_t items, int segments)
{
    int rc;
    if (items > b->sortItems) {
        const size_t cap = items + items / 2 + 1024;
        if (b->sortItems == 0) {
            if ((rc = balloc(b, &b->dSortKeys[0], cap)) || (rc = balloc(b, &b->dSortKeys[1], cap)) || (rc = balloc(b, &b->dSortMeta, cap))) return rc;
        } else {
            if ((rc = bgrow(b, &b->dSortKeys[0], cap, 0)) || (rc = bgrow(b, &b->dSortKeys[1], cap, 0)) || (rc = bgrow(b, &b->dSortMeta, cap, 0))) return rc;
        }
        b->sortItems = cap;
    }
    size_t tb = 0;
    HIPCHK(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, tb, b->dSortKeys[0], b->dSortKeys[1], b->dOccMeta, b->dSortMeta, (int)items, segments,
                                                       b->dSegBegin, b->dSegEnd, 0, 64, b->stream));
    if (tb > b->sortTempBytes) {
        const size_t cap = tb + tb / 2 + 256;
        if (b->sortTempBytes == 0) { if ((rc = balloc(b, &b->dSortTemp, cap))) return rc; }
        else if ((rc = bgrow(b, &b->dSortTemp, cap, 0))) return rc;
        b->sortTempBytes = cap;
    }
    return S3_OK;
}

extern "C" int s3_batch_align_pair(s3_batch *b, uint32_t numMismatch, uint32_t maxHitsPerEnd, int reportType, uint32_t insertLow, uint32_t insertHigh,
                                   uint32_t peMaxOutputPerPair)
{
    if (!b) return fail(S3_EINVAL, "null batch");
    if (reportType < 1 || reportType > 4) return fail(S3_EINVAL, "reportType must be 1..4");
    if (numMismatch > S3_MAX_MISMATCH) return fail(S3_EINVAL, "numMismatch must be 0..4");
    if (maxHitsPerEnd == 0 || maxHitsPerEnd > 0xFFFFFFu) return fail(S3_EINVAL, "maxHitsPerEnd must be 1..16777215");
    if (b->n & 1u) return fail(S3_EINVAL, "s3_batch_align_pair: the batch must hold an even number of reads (read 2p and 2p+1 are a pair)");
    if (b->longMode) return fail(S3_EINVAL, "s3_batch_align_pair: seed lengths are a single-end feature");
    HIPCHK(hipSetDevice(b->ix->dev));
    int rc = ensurePairBuffers(b);
    if (rc) return rc;
    const bool bestMode = reportType >= 3, mapq = !bestMode;
    const uint32_t n = b->n, nPairs = n / 2;
    b->searchLaunches = 0;
    b->occUsed = 0;
    b->xaUsed = 0;
    b->recordsTotal = 0;
    HIPCHK(hipMemsetAsync(b->dFlags, 0, 4, b->stream));
    if (b->counting) HIPCHK(hipMemsetAsync(b->dCounters, 0, 64, b->stream));
    HIPCHK(hipEventRecord(b->evTotal[0], b->stream));
    HIPCHK(hipMemsetAsync(b->dPhase, 0, n ? n : 1, b->stream));
    HIPCHK(hipMemsetAsync(b->dFinalWithError, 0, (size_t)(n ? n : 1) * 32, b->stream));
    HIPCHK(hipMemsetAsync(b->dOccStart, 0, (size_t)(n ? n : 1) * 4, b->stream));
    HIPCHK(hipMemsetAsync(b->dOccCount, 0, ((size_t)n + 1) * 4, b->stream));
    HIPCHK(hipMemsetAsync(b->dXaStart, 0, ((size_t)nPairs + 1) * 4, b->stream));
    HIPCHK(hipMemsetAsync(b->dXaCountOut, 0, ((size_t)nPairs + 1) * 4, b->stream));
    HIPCHK(hipMemsetAsync(b->dPairMin, 255, n ? n : 1, b->stream));
    HIPCHK(hipMemsetAsync(b->dNumSel, 0, 16, b->stream));
    if (nPairs) hipLaunchKernelGGL(pairInitKernel, dim3((nPairs + 255) / 256), dim3(256), 0, b->stream, nPairs, b->dPairs);
    uint32_t phases[4];
    int nPhases = 0;
    if (reportType == 2) { phases[nPhases++] = 1; if (numMismatch > 1) phases[nPhases++] = 2; if (numMismatch > 2) phases[nPhases++] = numMismatch; }
    else if (reportType == 1) phases[nPhases++] = numMismatch;
    else { phases[nPhases++] = 0; if (numMismatch > 0) phases[nPhases++] = 1; if (numMismatch > 1) phases[nPhases++] = 2; if (numMismatch > 2) phases[nPhases++] = numMismatch; }
    const LongArgs noLong{false, 0, 0, 0, 0, 0};
    const uint32_t *active = nullptr;
    uint32_t nA = nPairs;
    int pp = 0;
    for (int ph = 0; ph < nPhases && nA > 0; ph++) {
        const uint32_t mm = phases[ph];
        const bool terminal = ph + 1 == nPhases;
        launchPairExpand(nA, active, b->dReadList, b->stream);
        rc = searchPass(b, b->dReadList, 2 * nA, mm, maxHitsPerEnd);
        if (rc) return rc;
        PairClassifyArgs ca{};
        ca.nPairs = nA; ca.activePairs = active; ca.totalOcc = b->dTotalOcc; ca.withError = b->dWithError; ca.mm = mm; ca.maxHits = maxHitsPerEnd;
        ca.terminal = terminal; ca.mapq = mapq; ca.slotState = b->dState; ca.pairNext = b->dPairNext; ca.pairMin = b->dPairMin; ca.occCount = b->dOccCount;
        launchPairClassify(ca, b->stream);
        const uint32_t regionBase = b->occUsed;
        rc = decodeAndPublish(b, b->dReadList, 2 * nA, 1, 0, maxHitsPerEnd, 2, noLong);       // selFlags = ends to search again
        if (rc) return rc;
        if (mapq && mm < 4) {
            size_t tb = b->tempBytes;
            HIPCHK(hipcub::DeviceSelect::Flagged(b->dTemp, tb, b->dReadList, b->dSelFlags, b->dActive[0], b->dNumSel, (int)(2 * nA), b->stream));
            uint32_t nR = 0;
            HIPCHK(hipMemcpyAsync(&nR, b->dNumSel, 4, hipMemcpyDeviceToHost, b->stream));
            HIPCHK(hipStreamSynchronize(b->stream));
            if (nR) {
                // ProcessOneMoreMismatchAllCases on that end; THAT result is what the pairing sees
                rc = searchPass(b, b->dActive[0], nR, mm + 1, maxHitsPerEnd);
                if (rc) return rc;
                hipLaunchKernelGGL(stateAllKernel, dim3((nR + 255) / 256), dim3(256), 0, b->stream, nR, b->dTotalOcc, b->dState);
                rc = decodeAndPublish(b, b->dActive[0], nR, 1, 0, maxHitsPerEnd, 0, noLong);
                if (rc) return rc;
            }
        }
        // position-sort both lists of every pair with hits on both ends (PERadixSort), then pair them up
        const size_t items = (size_t)b->occUsed - regionBase;
        PairSortArgs sa{};
        sa.nPairs = nA; sa.activePairs = active; sa.pairNext = b->dPairNext; sa.occStart = b->dOccStart; sa.occCount = b->dOccCount;
        sa.regionBase = regionBase; sa.occPos = b->dOccPos; sa.occMeta = b->dOccMeta; sa.segBegin = b->dSegBegin; sa.segEnd = b->dSegEnd;
        if (items) {
            rc = ensureSortBuffers(b, items, (int)(2 * nA));
            if (rc) return rc;
            sa.keys = b->dSortKeys[0];
            launchPairSortKeys(sa, b->stream);
            size_t tb = b->sortTempBytes;
            HIPCHK(hipcub::DeviceSegmentedRadixSort::SortPairs(b->dSortTemp, tb, b->dSortKeys[0], b->dSortKeys[1], b->dOccMeta + regionBase, b->dSortMeta,
                                                               (int)items, (int)(2 * nA), b->dSegBegin, b->dSegEnd, 0, 64, b->stream));
            launchPairSortBack(sa, b->dSortKeys[1], b->dSortMeta, b->stream);
        }
        launchPairSnapshotCounts(nA, active, b->dOccCount, b->dOccCountIn, b->stream);
        PairSweepArgs sw{};
        sw.nPairs = nA; sw.activePairs = active; sw.pairNext = b->dPairNext; sw.pairMin = b->dPairMin; sw.finalWithError = b->dFinalWithError;
        sw.occStart = b->dOccStart; sw.occCount = b->dOccCount; sw.occCountIn = b->dOccCountIn; sw.occPos = b->dOccPos; sw.occMeta = b->dOccMeta;
        sw.readLengths = b->dLens; sw.insertLow = insertLow; sw.insertHigh = insertHigh; sw.peMaxOutputPerPair = peMaxOutputPerPair; sw.mm = mm;
        sw.terminal = terminal; sw.pairs = b->dPairs; sw.bestIdx = b->dBestIdx; sw.xaCount = b->dXaCount; sw.xaStart = b->dXaStart; sw.xaCountOut = b->dXaCountOut;
        launchPairStats(sw, b->stream);
        HIPCHK(hipMemsetAsync(b->dXaCount + nA, 0, 4, b->stream));
        size_t tb = b->tempBytes;
        HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->dTemp, tb, b->dXaCount, b->dXaOffsets, (int)(nA + 1), b->stream));
        uint32_t xaAdded = 0;
        HIPCHK(hipMemcpyAsync(&xaAdded, b->dXaOffsets + nA, 4, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        if ((uint64_t)b->xaUsed + xaAdded > b->xaCap) {
            uint64_t cap = (uint64_t)b->xaCap * 2;
            if (cap < (uint64_t)b->xaUsed + xaAdded) cap = (uint64_t)b->xaUsed + xaAdded + 65536;
            if (cap > kMaxEntries) return fail(S3_EOVERFLOW, "pair list overflow at the 32-bit limit");
            if ((rc = bgrow(b, &b->dXa, (size_t)cap, b->xaUsed)) || (rc = bgrow(b, &b->dXaOut, (size_t)cap, 0))) return rc;
            b->xaCap = (uint32_t)cap;
        }
        launchPairEmit(sw, b->dXaOffsets, b->xaUsed, b->dXa, b->stream);
        b->xaUsed += xaAdded;
        if (terminal) break;
        const uint32_t *idsIn = active;
        if (!idsIn) {
            hipLaunchKernelGGL(iotaKernel, dim3((nA + 255) / 256), dim3(256), 0, b->stream, b->dIota, nA);
            idsIn = b->dIota;
        }
        tb = b->tempBytes;
        uint32_t *nextList = b->dPairActive[pp];
        HIPCHK(hipcub::DeviceSelect::Flagged(b->dTemp, tb, idsIn, b->dPairNext, nextList, b->dNumSel, (int)nA, b->stream));
        uint32_t nNext = 0;
        HIPCHK(hipMemcpyAsync(&nNext, b->dNumSel, 4, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        active = nextList;
        nA = nNext;
        pp ^= 1;
    }
    // read-ordered CSR of the occurrence lists of the pairs without a proper pairing, pair-ordered CSR of the XA pairs
    size_t tb = b->tempBytes;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->dTemp, tb, b->dOccCount, b->dOutOffsets, (int)(n + 1), b->stream));
    if (n) hipLaunchKernelGGL(gatherOccKernel, dim3((n + 255) / 256), dim3(256), 0, b->stream, n, b->dOccStart, b->dOutOffsets, b->dOccPos, b->dOccMeta,
                              b->dOutPos, b->dOutMeta);
    tb = b->tempBytes;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->dTemp, tb, b->dXaCountOut, b->dXaOutOffsets, (int)(nPairs + 1), b->stream));
    launchPairGatherXa(nPairs, b->dXaStart, b->dXaOutOffsets, b->dXa, b->dXaOut, b->stream);
    HIPCHK(hipEventRecord(b->evTotal[1], b->stream));
    HIPCHK(hipGetLastError());
    b->lastKind = 3;
    return S3_OK;
}

extern "C" int s3_batch_fetch_pairs(s3_batch *b, s3_pair_result *pairs, uint64_t *xaOffsets, s3_pair_xa *xa, size_t xaCap, uint64_t *nXaOut,
                                    uint64_t *occOffsets, uint32_t *occPos, uint8_t *occStrand, uint8_t *occMis, size_t occCap, uint64_t *nOccOut)
{
    if (!b || b->lastKind != 3) return fail(S3_EINVAL, "s3_batch_fetch_pairs: no paired-end result pending");
    int rc = s3_batch_sync(b);
    if (rc) return rc;
    static_assert(sizeof(s3_pair_result) == sizeof(DevPair) && sizeof(s3_pair_xa) == sizeof(DevPairXa), "pair layouts");
    const uint32_t n = b->n, nPairs = n / 2;
    if (pairs && nPairs) HIPCHK(hipMemcpy(pairs, b->dPairs, (size_t)nPairs * sizeof(DevPair), hipMemcpyDeviceToHost));
    std::vector<uint32_t> off((size_t)nPairs + 1);
    HIPCHK(hipMemcpy(off.data(), b->dXaOutOffsets, ((size_t)nPairs + 1) * 4, hipMemcpyDeviceToHost));
    const uint64_t nXa = off[nPairs];
    if (nXaOut) *nXaOut = nXa;
    if (xaOffsets) for (size_t i = 0; i <= nPairs; i++) xaOffsets[i] = off[i];
    if (xa) {
        if (nXa > xaCap) return fail(S3_EOVERFLOW, "s3_batch_fetch_pairs: pair list array too small");
        if (nXa) HIPCHK(hipMemcpy(xa, b->dXaOut, nXa * sizeof(DevPairXa), hipMemcpyDeviceToHost));
    }
    std::vector<uint32_t> ooff((size_t)n + 1);
    HIPCHK(hipMemcpy(ooff.data(), b->dOutOffsets, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
    const uint64_t total = ooff[n];
    if (nOccOut) *nOccOut = total;
    if (occOffsets) for (size_t i = 0; i <= n; i++) occOffsets[i] = ooff[i];
    if (occPos || occStrand || occMis) {
        if (total > occCap) return fail(S3_EOVERFLOW, "s3_batch_fetch_pairs: occurrence arrays too small");
        if (occPos && total) HIPCHK(hipMemcpy(occPos, b->dOutPos, total * 4, hipMemcpyDeviceToHost));
        if ((occStrand || occMis) && total) {
            std::vector<uint32_t> meta(total);
            HIPCHK(hipMemcpy(meta.data(), b->dOutMeta, total * 4, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < total; i++) {
                if (occStrand) occStrand[i] = (uint8_t)(meta[i] & 0xFF);
                if (occMis) occMis[i] = (uint8_t)((meta[i] >> 8) & 0xFF);
            }
        }
    }
    return S3_OK;
}

extern "C" int s3_batch_last_timing(s3_batch *b, float *totalMs, float *searchMs, uint32_t *searchLaunches)
{
    if (!b || b->lastKind == 0) return fail(S3_EINVAL, "no timed call");
    HIPCHK(hipSetDevice(b->ix->dev));
    HIPCHK(hipStreamSynchronize(b->stream));
    float t = 0, s = 0;
    HIPCHK(hipEventElapsedTime(&t, b->evTotal[0], b->evTotal[1]));
    uint32_t timed = b->searchLaunches;
    if (timed * 2 > b->evSearch.size()) timed = (uint32_t)(b->evSearch.size() / 2);
    for (uint32_t i = 0; i < timed; i++) {
        float x = 0;
        HIPCHK(hipEventElapsedTime(&x, b->evSearch[2 * i], b->evSearch[2 * i + 1]));
        s += x;
    }
    if (totalMs) *totalMs = t;
    if (searchMs) *searchMs = s;
    if (searchLaunches) *searchLaunches = b->searchLaunches;
    return S3_OK;
}

extern "C" int s3_batch_counters(s3_batch *b, s3_counters *out)
{
    if (!b || !out) return fail(S3_EINVAL, "null argument");
    HIPCHK(hipSetDevice(b->ix->dev));
    unsigned long long c[8] = {0};
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(c, b->dCounters, 64, hipMemcpyDeviceToHost));
    out->rankQueries = c[0]; out->ltLookups = c[1]; out->saReads = c[2]; out->ceWindows = c[3];
    out->records = b->recordsTotal;
    return S3_OK;
}

extern "C" int s3_bwt_search(s3_index *ix, const uint32_t *queries, const uint32_t *readLengths, uint32_t n, uint32_t wordPerQuery,
                             uint32_t numMismatch, uint32_t maxOutputPerRead, uint64_t **hitOffsets, s3_hit **hits, uint32_t **totalOcc,
                             uint32_t **withError)
{
    if (!ix || !hitOffsets || !hits) return fail(S3_EINVAL, "null argument");
    s3_batch *b = nullptr;
    int rc = s3_batch_create(ix, n ? n : 1, wordPerQuery, &b);
    if (rc) return rc;
    rc = s3_batch_upload(b, queries, readLengths, n);
    if (!rc) rc = s3_batch_search(b, numMismatch, maxOutputPerRead);
    uint64_t nh = 0;
    uint64_t *off = (uint64_t *)malloc(((size_t)n + 1) * 8);
    uint32_t *tot = (uint32_t *)malloc((size_t)(n ? n : 1) * 4), *we = (uint32_t *)malloc((size_t)(n ? n : 1) * 32);
    if (!rc) rc = s3_batch_fetch_hits(b, off, nullptr, 0, tot, we, &nh);
    s3_hit *h = (s3_hit *)malloc((size_t)(nh ? nh : 1) * sizeof(s3_hit));
    if (!rc) rc = s3_batch_fetch_hits(b, nullptr, h, nh, nullptr, nullptr, nullptr);
    s3_batch_free(b);
    if (rc) { free(off); free(tot); free(we); free(h); return rc; }
    *hitOffsets = off; *hits = h;
    if (totalOcc) *totalOcc = tot; else free(tot);
    if (withError) *withError = we; else free(we);
    return S3_OK;
}

// measurement aid (tools/hbm_probe.py): random 64-byte-line gather rate of the device
extern "C" int s3_probe_gather(int device, size_t tableBytes, uint32_t blocks, uint32_t itersPerLane, int ilp, int loadsPerLine, int dependent,
                               int repeats, float *msOut, uint64_t *linesOut)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(S3_ENODEV, "no HIP device available");
    HIPCHK(hipSetDevice(device));
    uint4 *table = nullptr;
    uint32_t *sink = nullptr;
    const uint64_t nLines = tableBytes / 64;
    if (nLines == 0 || repeats < 1) return fail(S3_EINVAL, "bad probe arguments");
    HIPCHK(hipMalloc((void **)&table, nLines * 64));
    HIPCHK(hipMalloc((void **)&sink, 16));
    HIPCHK(hipMemset(table, 1, nLines * 64));
    hipStream_t s;
    hipEvent_t e0, e1;
    HIPCHK(hipStreamCreate(&s));
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    launchGatherProbe(table, nLines, blocks, itersPerLane, ilp, loadsPerLine, dependent, sink, s);     // warm-up
    HIPCHK(hipEventRecord(e0, s));
    for (int r = 0; r < repeats; r++) launchGatherProbe(table, nLines, blocks, itersPerLane, ilp, loadsPerLine, dependent, sink, s);
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    const int eff = ilp >= 8 ? 8 : ilp >= 4 ? 4 : ilp >= 2 ? 2 : 1;
    if (msOut) *msOut = ms / repeats;
    if (linesOut) *linesOut = (uint64_t)blocks * 256ull * itersPerLane * eff;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    (void)hipFree(table); (void)hipFree(sink);
    return S3_OK;
}

// ---------------------------------------------------------------------------------------------
// semi-global DP
// ---------------------------------------------------------------------------------------------
struct s3_dp {
    int dev = 0, scheme = 1;
    uint32_t maxProblems = 0, maxRead = 0, maxDNA = 0, maxTable = 0, patternLen = 0;
    s3_dp_scores sc{};
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    std::vector<void *> allocs;
    uint32_t *dDna = nullptr, *dReads = nullptr, *dDnaLen = nullptr, *dReadLen = nullptr, *dClipLt = nullptr, *dClipRt = nullptr, *dAL = nullptr, *dAR = nullptr;
    int32_t *dCut = nullptr, *dScores = nullptr;
    uint32_t *dHit = nullptr, *dStart = nullptr, *dCnt = nullptr, *dCols = nullptr, *dTable = nullptr;
    unsigned char *dPattern = nullptr;
    size_t tableWords = 0;
    bool timed = false;
};

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
    if (maxReadLength > 4096 || maxDNALength > 16384) return fail(S3_EINVAL, "s3_dp_create: lengths too large");
    if (scheme == 1 && maxDPTableLength < maxDNALength) return fail(S3_EINVAL, "s3_dp_create: scheme 1 needs maxDPTableLength >= maxDNALength");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(S3_ENODEV, "no HIP device available (libs3dp has no CPU path)");
    if (device < 0 || device >= ndev) return fail(S3_EINVAL, "bad device ordinal");
    HIPCHK(hipSetDevice(device));
    s3_dp *h = new s3_dp();
    h->dev = device; h->scheme = scheme == 2 ? 2 : 1; h->maxProblems = maxProblems; h->maxRead = maxReadLength; h->maxDNA = maxDNALength;
    h->maxTable = maxDPTableLength; h->patternLen = maxReadLength + maxDPTableLength; h->sc = *sc;
    const size_t rnd = ((size_t)maxProblems + 31) / 32 * 32;
    const size_t dnaW = rnd * ((maxDNALength + 15) / 16), rdW = rnd * ((maxReadLength + 15) / 16);
    h->tableWords = (size_t)((maxReadLength + 63) / 64) * (maxDPTableLength + 64) * 64;
    int rc;
#define DA(p, c) if ((rc = dalloc(h, &h->p, (c)))) { s3_dp_free(h); return rc; }
    DA(dDna, dnaW); DA(dReads, rdW); DA(dDnaLen, rnd); DA(dReadLen, rnd); DA(dClipLt, rnd); DA(dClipRt, rnd); DA(dAL, rnd); DA(dAR, rnd);
    DA(dCut, rnd); DA(dScores, rnd); DA(dHit, rnd); DA(dStart, rnd); DA(dCnt, rnd); DA(dCols, rnd);
    DA(dPattern, (size_t)maxProblems * h->patternLen);
    DA(dTable, h->tableWords * maxProblems);
#undef DA
    HIPCHK(hipStreamCreate(&h->stream));
    HIPCHK(hipEventCreate(&h->ev[0])); HIPCHK(hipEventCreate(&h->ev[1]));
    *outp = h;
    return S3_OK;
}

// launch the DP kernels on what is in the aligner's device buffers
static int dpRun(s3_dp *h, uint32_t n, bool hasClipLt, bool hasClipRt, bool hasAL, bool hasAR)
{
    hipStream_t s = h->stream;
    DpArgs a{};
    a.dna = h->dDna; a.dnaLengths = h->dDnaLen; a.reads = h->dReads; a.readLengths = h->dReadLen;
    a.clipLt = hasClipLt ? h->dClipLt : nullptr; a.clipRt = hasClipRt ? h->dClipRt : nullptr;
    a.anchorLeft = hasAL ? h->dAL : nullptr; a.anchorRight = hasAR ? h->dAR : nullptr;
    a.cutoffs = h->dCut; a.scores = h->dScores; a.hitLocs = h->dHit; a.startOffsets = h->dStart; a.maxScoreCounts = h->dCnt; a.tableCols = h->dCols;
    a.pattern = h->dPattern; a.table = h->dTable; a.tableWordsPerProblem = h->tableWords;
    a.n = n; a.maxDNALength = h->maxDNA; a.maxReadLength = h->maxRead; a.maxDPTableLength = h->maxTable; a.patternLen = h->patternLen;
    a.ldsPerWave = (((h->maxDNA + 1u + 15u) & ~15u) + (h->maxDNA + 1u) * 12u + 15u) & ~15u;
    a.match = h->sc.matchScore; a.mismatch = h->sc.mismatchScore; a.gapOpen = h->sc.gapOpenScore; a.gapExtend = h->sc.gapExtendScore; a.scheme = h->scheme;
    if ((size_t)a.ldsPerWave * 4 > 160 * 1024) return fail(S3_EINVAL, "s3_dp_align: maxDNALength too large for the LDS-resident window");
    HIPCHK(hipEventRecord(h->ev[0], s));
    launchDp(a, s);
    HIPCHK(hipEventRecord(h->ev[1], s));
    HIPCHK(hipGetLastError());
    h->timed = true;
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
// list travels.  PARITY UNPINNED (host code inside a CUDA source file): restated from the source, checked on simulated indel pairs.
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
        std::vector<uint8_t> pat((size_t)n * h->patternLen);
        HIPCHK(hipMemcpyAsync(scores.data(), h->dScores, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(hit.data(), h->dHit, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(cnt.data(), h->dCnt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(pat.data(), h->dPattern, pat.size(), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        // algnmtCPUThread (:2329-2441)
        for (uint32_t id = 0; id < n; id++) {
            s3_dp_half_result r{};
            const uint32_t alignedID = refer[id].readID, alignedIsMate = alignedID & 1u;
            r.readID = alignedID - alignedIsMate;
            r.cigarOffset = 0xFFFFFFFFu;
            uint32_t dpPos;
            if (scores[id] >= cutoffThreshold[1 - alignedIsMate]) {
                CigarEnc enc;
                char last = 'N';
                for (const uint8_t *p = pat.data() + (size_t)id * h->patternLen; *p != 0; p++) {
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
        }
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
        std::vector<uint8_t> pat((size_t)n * h->patternLen);
        ECHK(hipMemcpyAsync(scores.data(), h->dScores, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        ECHK(hipMemcpyAsync(hit.data(), h->dHit, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        ECHK(hipMemcpyAsync(cnt.data(), h->dCnt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        ECHK(hipMemcpyAsync(pat.data(), h->dPattern, pat.size(), hipMemcpyDeviceToHost, s));
        ECHK(hipStreamSynchronize(s));
#undef ECHK
        for (uint32_t i = 0; i < n; i++) {
            if (scores[i] < cutoffThreshold) continue;
            CigarEnc enc;
            char last = 'N';
            for (const uint8_t *p = pat.data() + (size_t)i * h->patternLen; *p != 0; p++) {
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
        }
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
// Seeding for the DP stages of single reads: seeds cut from the reads at the tabulated offsets, searched exactly, the seeds
// without a hit once more with one mismatch, every hit turned into an estimated read position, positions of one read and strand merged.
// Replaces, of the reference: SingleEndSeedingBatch::pack / packSeeds (DV-DPfunctions.cu:1058-1100), single_1_mismatch_alignment2
// (alignment.cu:1839-1894) + the hit limit of hostKernelSingle (CPUfunctions.cpp:2843-2876: a seed with more than maxHitNum hits is
// dropped), decodePositions (:1143-1219, sorted by strand, read, position) and singleMerge (:1102-1141, positions closer than
// DPS_DIVIDE_GAP = 50 collapse into the first).  Everything runs on the device next to the resident reads; the seed batch is an
// ordinary s3_batch.  PARITY UNPINNED (host code inside CUDA sources); tested against a restatement built on the pinned search oracle.
// ---------------------------------------------------------------------------------------------
struct SeedDesc { uint32_t readId, off, len; };

__global__ void seedPackKernel(uint32_t nSeeds, const SeedDesc *seeds, const uint32_t *queries, uint32_t wpq, uint32_t *seedQueries, uint32_t seedWpq,
                               uint32_t *seedLens)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (uint64_t)nSeeds * seedWpq) return;
    const uint32_t j = (uint32_t)(g / seedWpq), w = (uint32_t)(g % seedWpq);
    const SeedDesc d = seeds[j];
    const uint32_t *src = queries + (size_t)(d.readId >> 5) * 32u * wpq + (d.readId & 31u);
    uint32_t out = 0;
    for (uint32_t c = 0; c < 16u; c++) {
        const uint32_t i = w * 16u + c;
        if (i < d.len) {
            const uint32_t q = d.off + i;
            out |= ((src[(size_t)(q >> 4) * 32u] >> ((q & 15u) * 2u)) & 3u) << (2u * c);
        }
    }
    seedQueries[(size_t)(j >> 5) * 32u * seedWpq + (j & 31u) + (size_t)w * 32u] = out;
    if (w == 0) seedLens[j] = d.len;
}

// state per seed after a pass: 1 = hits kept, 0 = none / dropped; `retry` marks seeds without any hit (second pass with one mismatch)
__global__ void seedClassifyKernel(uint32_t n, const uint32_t *active, const uint32_t *totalOcc, uint32_t maxHit, uint8_t *state, uint8_t *retry, uint8_t *status)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = totalOcc[i], j = active ? active[i] : i;
    state[i] = (t > 0 && t <= maxHit) ? 1 : 0;
    retry[i] = t == 0;
    status[j] = t == 0 ? 0 : (t <= maxHit ? 1 : 4);
}

// estimated read position of every seed hit (MC_EstimatedPos, DV-DPfunctions.cu:1163): '+': pos - offset, '-': pos + seedLen + offset - readLen;
// key = strand << 63 | readID << 32 | position, the order decodePositions sorts into
__global__ void seedKeysKernel(uint32_t nSeeds, const SeedDesc *seeds, const uint32_t *readLens, const uint32_t *occStart, const uint32_t *occCount,
                               const uint32_t *outOffsets, const uint32_t *occPos, const uint32_t *occMeta, uint64_t *keys)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nSeeds) return;
    const SeedDesc d = seeds[j];
    const uint32_t cnt = occCount[j], src = occStart[j], dst = outOffsets[j], readLen = readLens[d.readId];
    for (uint32_t k = 0; k < cnt; k++) {
        const uint32_t pos = occPos[src + k], strand = occMeta[src + k] & 0xFFu;
        const uint32_t est = strand == 1u ? pos - d.off : pos + d.len + d.off - readLen;
        keys[dst + k] = (uint64_t)(strand == 2u) << 63 | (uint64_t)d.readId << 32 | est;
    }
}

// singleMerge: within one (strand, read) run the first position is kept, then every position more than 50 beyond the last kept one
__global__ void seedMergeFlagKernel(uint64_t n, const uint64_t *keys, uint8_t *keep)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t me = keys[i];
    if (i > 0 && (keys[i - 1] >> 32) == (me >> 32)) return;             // not the start of a run: the run's first thread walks it
    uint32_t prev = (uint32_t)me;
    keep[i] = 1;
    for (uint64_t k = i + 1; k < n && (keys[k] >> 32) == (me >> 32); k++) {
        const uint32_t cur = (uint32_t)keys[k];
        if ((uint64_t)prev + 50u < cur) { keep[k] = 1; prev = cur; } else keep[k] = 0;
    }
}

extern "C" int s3_seed_candidates(s3_batch *reads, s3_batch *seedBatch, const uint32_t *readLengths, int stage, uint32_t maxHitNum,
                                  s3_seed_candidate **candOut, size_t *nCandOut, uint8_t **seedStatusOut, size_t *nSeedsOut)
{
    if (!reads || !seedBatch || !readLengths || !candOut || !nCandOut) return fail(S3_EINVAL, "s3_seed_candidates: null argument");
    if (reads->ix != seedBatch->ix || reads == seedBatch) return fail(S3_EINVAL, "s3_seed_candidates: the seed batch must be a second batch on the same index");
    HIPCHK(hipSetDevice(reads->ix->dev));
    std::vector<SeedDesc> seeds;
    for (uint32_t r = 0; r < reads->n; r++) {
        int seedLength = 0, num = 0, pos[128];
        if (s3_seed_positions(stage, (int)readLengths[r], &seedLength, pos, &num) != S3_OK) return fail(S3_EINVAL, "s3_seed_candidates: bad stage");
        for (int i = 0; i < num; i++) {
            if (pos[i] < 0 || seedLength < 16 || (uint32_t)(pos[i] + seedLength) > readLengths[r]) continue;   // reads too short for the stage's seeds
            seeds.push_back(SeedDesc{r, (uint32_t)pos[i], (uint32_t)seedLength});
        }
    }
    const uint32_t nSeeds = (uint32_t)seeds.size();
    if (nSeeds > seedBatch->maxReads) return fail(S3_EINVAL, "s3_seed_candidates: seed batch too small");
    if (seedBatch->wpq * 16u < 70u) return fail(S3_EINVAL, "s3_seed_candidates: the seed batch needs wordPerQuery >= 8");
    s3_batch *b = seedBatch;
    hipStream_t st = b->stream;
    SeedDesc *dSeeds = nullptr;
    uint8_t *dRetry = nullptr, *dStatus = nullptr, *dKeep = nullptr;
    uint64_t *dKeys = nullptr, *dKeysOut = nullptr;
    char *dSortTmp = nullptr;
    std::vector<void *> tmp;
    auto talloc = [&](void **p, size_t bytes) -> int {
        hipError_t e = hipMalloc(p, bytes ? bytes : 16);
        if (e != hipSuccess) return fail(S3_ENOMEM, std::string("hipMalloc(seeding): ") + hipGetErrorString(e));
        tmp.push_back(*p);
        return S3_OK;
    };
    auto cleanup = [&]() { for (void *p : tmp) (void)hipFree(p); };
    int rc;
    if ((rc = talloc((void **)&dSeeds, (size_t)nSeeds * sizeof(SeedDesc))) || (rc = talloc((void **)&dRetry, nSeeds)) || (rc = talloc((void **)&dStatus, nSeeds))) { cleanup(); return rc; }
#define SCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(S3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    SCHK(hipMemcpyAsync(dSeeds, seeds.data(), (size_t)nSeeds * sizeof(SeedDesc), hipMemcpyHostToDevice, st));
    {
        const size_t roundUp = ((size_t)nSeeds + 31) / 32 * 32;
        SCHK(hipMemsetAsync(b->dQueries, 0, roundUp * b->wpq * 4, st));
        const uint64_t threads = (uint64_t)nSeeds * b->wpq;
        if (threads) hipLaunchKernelGGL(seedPackKernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, nSeeds, dSeeds, reads->dQueries, reads->wpq,
                                        b->dQueries, b->wpq, b->dLens);
        launchOrientReads(b->dQueries, b->dLens, b->dReadsT, nSeeds, b->wpq, b->wpq, st);
        b->n = nSeeds; b->longMode = false; b->lastKind = 0;
        b->uniformLen = nSeeds ? seeds[0].len : 0;
        for (uint32_t j = 1; j < nSeeds; j++) if (seeds[j].len != seeds[0].len) { b->uniformLen = 0; break; }
        b->fullUniformLen = b->uniformLen;
    }
    b->searchLaunches = 0; b->occUsed = 0; b->recordsTotal = 0;
    SCHK(hipMemsetAsync(b->dFlags, 0, 4, st));
    SCHK(hipMemsetAsync(b->dOccStart, 0, (size_t)(nSeeds ? nSeeds : 1) * 4, st));
    SCHK(hipMemsetAsync(b->dOccCount, 0, ((size_t)nSeeds + 1) * 4, st));
    SCHK(hipMemsetAsync(b->dNumSel, 0, 16, st));
    const LongArgs noLong{false, 0, 0, 0, 0, 0};
    const uint32_t *active = nullptr;
    uint32_t nA = nSeeds;
    for (uint32_t mm = 0; mm <= 1 && nA > 0; mm++) {                       // exact first, then one mismatch for the seeds without a hit
        if ((rc = searchPass(b, active, nA, mm, 0xFFFFFFFFu))) { cleanup(); return rc; }
        hipLaunchKernelGGL(seedClassifyKernel, dim3((nA + 255) / 256), dim3(256), 0, st, nA, active, b->dTotalOcc, maxHitNum, b->dState, dRetry, dStatus);
        if ((rc = decodeAndPublish(b, active, nA, 1, 0, 0xFFFFFFFFu, 0, noLong))) { cleanup(); return rc; }
        if (mm == 1) break;
        const uint32_t *idsIn = active;
        if (!idsIn) { hipLaunchKernelGGL(iotaKernel, dim3((nA + 255) / 256), dim3(256), 0, st, b->dIota, nA); idsIn = b->dIota; }
        size_t tb = b->tempBytes;
        SCHK(hipcub::DeviceSelect::Flagged(b->dTemp, tb, idsIn, dRetry, b->dActive[0], b->dNumSel, (int)nA, st));
        uint32_t nNext = 0;
        SCHK(hipMemcpyAsync(&nNext, b->dNumSel, 4, hipMemcpyDeviceToHost, st));
        SCHK(hipStreamSynchronize(st));
        active = b->dActive[0];
        nA = nNext;
    }
    // all kept hits -> keys -> sort -> merge
    size_t tb = b->tempBytes;
    SCHK(hipcub::DeviceScan::ExclusiveSum(b->dTemp, tb, b->dOccCount, b->dOutOffsets, (int)(nSeeds + 1), st));
    uint32_t total = 0;
    SCHK(hipMemcpyAsync(&total, b->dOutOffsets + nSeeds, 4, hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    std::vector<s3_seed_candidate> out;
    if (total) {
        size_t sortBytes = 0;
        SCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, sortBytes, dKeys, dKeysOut, (int)total, 0, 64, st));
        if ((rc = talloc((void **)&dKeys, (size_t)total * 8)) || (rc = talloc((void **)&dKeysOut, (size_t)total * 8)) || (rc = talloc((void **)&dKeep, total)) ||
            (rc = talloc((void **)&dSortTmp, sortBytes))) { cleanup(); return rc; }
        hipLaunchKernelGGL(seedKeysKernel, dim3((nSeeds + 255) / 256), dim3(256), 0, st, nSeeds, dSeeds, reads->dLens, b->dOccStart, b->dOccCount, b->dOutOffsets,
                           b->dOccPos, b->dOccMeta, dKeys);
        SCHK(hipcub::DeviceRadixSort::SortKeys(dSortTmp, sortBytes, dKeys, dKeysOut, (int)total, 0, 64, st));
        SCHK(hipMemsetAsync(dKeep, 0, total, st));
        hipLaunchKernelGGL(seedMergeFlagKernel, dim3((total + 255) / 256), dim3(256), 0, st, (uint64_t)total, dKeysOut, dKeep);
        std::vector<uint64_t> keys(total);
        std::vector<uint8_t> keep(total);
        SCHK(hipMemcpyAsync(keys.data(), dKeysOut, (size_t)total * 8, hipMemcpyDeviceToHost, st));
        SCHK(hipMemcpyAsync(keep.data(), dKeep, total, hipMemcpyDeviceToHost, st));
        SCHK(hipStreamSynchronize(st));
        for (uint32_t i = 0; i < total; i++)
            if (keep[i]) out.push_back(s3_seed_candidate{(uint32_t)((keys[i] >> 32) & 0x7FFFFFFFu), (uint32_t)keys[i], (uint8_t)((keys[i] >> 63) ? 2 : 1), {0, 0, 0}});
    }
    std::vector<uint8_t> status(nSeeds);
    if (nSeeds) SCHK(hipMemcpy(status.data(), dStatus, nSeeds, hipMemcpyDeviceToHost));
#undef SCHK
    cleanup();
    rc = checkFlags(b);
    if (rc) return rc;
    s3_seed_candidate *c = (s3_seed_candidate *)malloc(out.size() * sizeof(s3_seed_candidate) + 1);
    if (!c) return fail(S3_ENOMEM, "s3_seed_candidates: out of host memory");
    memcpy(c, out.data(), out.size() * sizeof(s3_seed_candidate));
    *candOut = c; *nCandOut = out.size();
    if (seedStatusOut) {
        uint8_t *sp = (uint8_t *)malloc(nSeeds + 1);
        if (!sp) { free(c); return fail(S3_ENOMEM, "s3_seed_candidates: out of host memory"); }
        memcpy(sp, status.data(), nSeeds);
        *seedStatusOut = sp;
    }
    if (nSeedsOut) *nSeedsOut = nSeeds;
    return S3_OK;
}

extern "C" int s3_dp_last_timing(s3_dp *h, float *ms)
{
    if (!h || !h->timed || !ms) return fail(S3_EINVAL, "no timed DP call");
    HIPCHK(hipSetDevice(h->dev));
    HIPCHK(hipEventElapsedTime(ms, h->ev[0], h->ev[1]));
    return S3_OK;
}
