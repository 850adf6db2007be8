// s3dp_api.hip -- the C-ABI of libs3dp.so (include/s3dp.h): index residency, batch objects, the
// k-mismatch search call and the single-end all-best ladder.  Host logic only; kernels are in
// s3dp_kernels.hip.  No CPU fallback exists: without a HIP device every entry point fails.
#include <hipcub/hipcub.hpp>
#include <atomic>
#include <chrono>
#include "s3dp_internal.h"

#include <stdlib.h>
#include <map>
#include <mutex>
#include <thread>
using namespace s3;
using namespace s3i;

static thread_local std::string g_err;
int s3i::fail(int code, const std::string &msg) { g_err = msg; return code; }

// The HIP runtime spreads a process's streams over FOUR hardware queues by default (GPU_MAX_HW_QUEUES); streams that share a queue run in order.  A
// streamed run keeps up to six batches in flight, each with a kernel stream and copy streams: on four queues a batch's kernels waited behind another
// batch's although nothing connected them -- 1 Mi-read batches reached 0.75 of the resident rate, 0.58 with four in flight; with sixteen queues 0.85.
// The variable is read when the runtime comes up, i.e. at the first HIP call, and it belongs to the HOST PROGRAM: the soap3-dp executable, the Python
// binding and bench.py set it (unless the user did) before their first HIP call; the library does not change its host's environment (include/s3dp.h).

// ---- scratch cache of the DP / seeding engines (s3dp_internal.h) -------------------------------------------------------------------------------------
namespace {
// A cached block remembers its last use as an EVENT recorded on the stream that used it (round 4; it used to remember the stream handle, and a taker on
// another thread could synchronise a stream its owner was destroying): whoever takes the block for another stream makes that stream wait for the event
// -- on the device, the host does not block -- and an event outlives its stream.
struct ScratchBlock { void *p; size_t bytes; hipStream_t lastStream; hipEvent_t done; int device; };
struct ScratchCache {
    std::mutex mu;
    std::vector<ScratchBlock> freeBlocks;                      // cached, by device
    std::map<void *, ScratchBlock> live;                       // handed out
    std::map<int, size_t> cachedBytes;                         // per device
} g_scratch;
size_t scratchClass(size_t bytes) { size_t c = 65536; while (c < bytes) c <<= 1; return c; }
}  // namespace

hipError_t s3i::scratchAlloc(void **p, size_t bytes, hipStream_t stream)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const size_t want = scratchClass(bytes ? bytes : 1);
    ScratchBlock got{nullptr, 0, nullptr, nullptr, dev};
    {
        std::lock_guard<std::mutex> g(g_scratch.mu);
        for (size_t i = 0; i < g_scratch.freeBlocks.size(); i++) {
            const ScratchBlock &b = g_scratch.freeBlocks[i];
            if (b.device == dev && b.bytes == want) { got = b; g_scratch.freeBlocks.erase(g_scratch.freeBlocks.begin() + (long)i); g_scratch.cachedBytes[dev] -= want; break; }
        }
    }
    if (got.p) {
        if (got.done && got.lastStream != stream && (e = hipStreamWaitEvent(stream, got.done, 0)) != hipSuccess) {
            // the block is in no list at this point: back to the cache as it was (its event still stands for its last use), not lost
            std::lock_guard<std::mutex> g(g_scratch.mu);
            g_scratch.freeBlocks.push_back(got);
            g_scratch.cachedBytes[dev] += want;
            return e;
        }
    } else {
        if ((e = hipMalloc(&got.p, want)) != hipSuccess) {
            s3i::scratchTrim(dev);                              // the cache may hold what is missing; then the flank tables give way (s3_index::flankMu)
            if ((e = s3i::mallocOrRelease(dev, &got.p, want)) != hipSuccess) return e;
        }
        got.bytes = want;
    }
    got.lastStream = stream;
    {
        std::lock_guard<std::mutex> g(g_scratch.mu);
        g_scratch.live[got.p] = got;
    }
    *p = got.p;
    return hipSuccess;
}

void s3i::scratchFree(void *p, hipStream_t stream)
{
    if (!p) return;
    ScratchBlock b{nullptr, 0, nullptr, nullptr, 0};
    bool giveBack = false, ours = false;
    {
        std::lock_guard<std::mutex> g(g_scratch.mu);
        auto it = g_scratch.live.find(p);
        if (it == g_scratch.live.end()) giveBack = true;        // not ours: plain hipMalloc memory
        else {
            b = it->second;
            g_scratch.live.erase(it);
            ours = true;
        }
    }
    if (ours) {
        // the block's last use, as an event on the stream that used it (recorded outside the lock: nobody else knows the block at this point)
        hipError_t e = b.done ? hipSuccess : hipEventCreateWithFlags(&b.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(b.done, stream);
        if (e != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(stream); if (b.done) { (void)hipEventDestroy(b.done); b.done = nullptr; } }
        b.lastStream = stream;
        std::lock_guard<std::mutex> g(g_scratch.mu);
        if (g_scratch.cachedBytes[b.device] + b.bytes > ((size_t)8 << 30)) giveBack = true;       // (per device)
        else { g_scratch.freeBlocks.push_back(b); g_scratch.cachedBytes[b.device] += b.bytes; }
    }
    if (giveBack) {
        if (ours && b.done) (void)hipEventDestroy(b.done);
        (void)hipFree(p);
    }
}

void s3i::scratchStreamGone(hipStream_t stream)
{
    // (the owner has synchronised `stream`: the events recorded on it are complete and stay valid; only the handle must not be compared again)
    std::lock_guard<std::mutex> g(g_scratch.mu);
    for (ScratchBlock &b : g_scratch.freeBlocks) if (b.lastStream == stream) b.lastStream = nullptr;
}

void s3i::scratchTrim(int device)
{
    std::vector<ScratchBlock> mine;
    {
        std::lock_guard<std::mutex> g(g_scratch.mu);
        for (size_t i = 0; i < g_scratch.freeBlocks.size();) {
            if (g_scratch.freeBlocks[i].device == device) { mine.push_back(g_scratch.freeBlocks[i]); g_scratch.cachedBytes[device] -= g_scratch.freeBlocks[i].bytes; g_scratch.freeBlocks.erase(g_scratch.freeBlocks.begin() + (long)i); }
            else i++;
        }
    }
    for (const ScratchBlock &b : mine) { if (b.done) (void)hipEventDestroy(b.done); (void)hipFree(b.p); }
}

namespace {
struct U32ToU64 { __host__ __device__ unsigned long long operator()(const uint32_t &x) const { return (unsigned long long)x; } };
}
int s3i::sum64Async(s3_batch *b, const uint32_t *counts, uint32_t n)
{
    hipcub::TransformInputIterator<unsigned long long, U32ToU64, const uint32_t *> it(counts, U32ToU64());
    size_t tb = b->tempBytes;
    HIPCHK(hipcub::DeviceReduce::Sum(b->dTemp, tb, it, b->dSum64, (int)n, b->stream));
    return S3_OK;
}

extern "C" const char *s3_last_error(void) { return g_err.c_str(); }
extern "C" const char *s3_version(void) { return "s3dp 0.1 (gfx950)"; }
extern "C" void s3_free(void *p) { free(p); }

extern "C" int s3_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The host NUMA node the device hangs off (its PCI function's numa_node in sysfs), -1 when the system does not say.  A host program keeps the threads that
// feed the device and the buffers they fill on that node (soap3_dp_main.cpp: ten million pairs took 4.1 s with the threads spread over two sockets, 3.75 s on one).
extern "C" int s3_device_host_node(int device, int *node)
{
    if (!node) return fail(S3_EINVAL, "null pointer");
    *node = -1;
    char id[64] = {0};
    if (hipDeviceGetPCIBusId(id, (int)sizeof id - 1, device) != hipSuccess) { (void)hipGetLastError(); return fail(S3_EHIP, "no such HIP device"); }
    for (char *c = id; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    const std::string path = std::string("/sys/bus/pci/devices/") + id + "/numa_node";
    if (FILE *f = fopen(path.c_str(), "r")) {
        int v = -1;
        if (fscanf(f, "%d", &v) == 1) *node = v;
        fclose(f);
    }
    return S3_OK;
}

extern "C" int s3_model_describe(int readLength, int maxError, char *buf, size_t bufLen)
{
    if (readLength < 1 || readLength > S3_MAX_READ_LENGTH || maxError < 0 || maxError > S3_MAX_MISMATCH) return fail(S3_EINVAL, "bad model args");
    return s3_model_text(readLength, maxError, buf, (unsigned long)bufLen);
}

// ---------------------------------------------------------------------------------------------

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

namespace {
struct DevTmp {                                    // a device allocation that lives as long as the scope (error returns included)
    void *p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
};
}  // namespace

static int uploadHalf(s3_index *ix, const s3_bwt_desc &h, DevBwt *out, hipStream_t s)
{
    if (!h.bwt || !h.occ) return fail(S3_EINVAL, "s3_index_upload: missing bwt/occ array");
    const size_t expectRecs = ((size_t)h.cumFreq[4] + 127) / 128 + 1;
    if (h.occRecs != expectRecs) return fail(S3_EINVAL, "s3_index_upload: occRecs does not match textLength");
    DevTmp tbHold, toHold;                         // staging of the file layout: gone when this returns, on every path
    uint4 *lines = nullptr;
    HIPCHK(hipMalloc(&tbHold.p, h.bwtWords * 4 + 16));
    HIPCHK(hipMalloc(&toHold.p, h.occRecs * 16));
    uint32_t *tb = (uint32_t *)tbHold.p, *to = (uint32_t *)toHold.p;
    int rc = devAlloc(ix, &lines, h.occRecs * 4);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(tb, h.bwt, h.bwtWords * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(to, h.occ, h.occRecs * 16, hipMemcpyHostToDevice, s));
    launchRelayout(tb, h.bwtWords, to, h.occRecs, lines, s);
    HIPCHK(hipStreamSynchronize(s));
    out->lines = lines;
    out->inverseSa0 = h.inverseSa0;
    out->n = h.cumFreq[4];
    return S3_OK;
}

static int indexUploadBody(s3_index *ix, const s3_index_desc *desc, hipStream_t s);

// ---- the flank tables give way to batches and aligners (s3_index::flankMu) ----
static std::mutex g_indexesMu;
static std::vector<s3_index *> g_indexes;                               // every resident index of the process

bool s3i::releaseFlankTable(int dev)
{
    s3_index *owner = nullptr;
    {
        std::lock_guard<std::mutex> g(g_indexesMu);
        for (s3_index *ix : g_indexes) if (ix->dev == dev && (ix->d.flankFwd || ix->d.flankBack)) { owner = ix; break; }
        if (!owner) return false;
        // (the registry's lock is held to the end: the index cannot be freed under this)
        std::unique_lock<std::shared_mutex> w(owner->flankMu);
        int cur = 0;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(dev);
        (void)hipDeviceSynchronize();                                    // every launch that read the tables was enqueued under the shared lock: done now
        const void *gone = owner->d.flankFwd ? (const void *)owner->d.flankFwd : (const void *)owner->d.flankBack;
        const bool wasFwd = owner->d.flankFwd != nullptr;
        if (wasFwd) owner->d.flankFwd = nullptr; else owner->d.flankBack = nullptr;
        for (void *&a : owner->allocs) if (a == gone) a = nullptr;
        (void)hipFree(const_cast<void *>(gone));
        const size_t bytes = ((size_t)owner->d.n + 1u) * sizeof(uint4);
        owner->bytes = owner->bytes > bytes ? owner->bytes - bytes : 0;
        (void)hipSetDevice(cur);
        fprintf(stderr, "[s3dp] device %d is out of memory: the flank table of check-and-extend in the order of the %s index (%.1f GB) is released; its tasks are verified from "
                        "the suffix array and the text from here on (same records; S3DP_FLANK, S3DP_FLANK_RESERVE_GB)\n", dev, wasFwd ? "reverse" : "forward", bytes / 1e9);
    }
    return true;
}

namespace {
struct Retired { std::mutex mu; std::vector<std::pair<int, void *>> list; size_t bytes[64] = {}; } g_retired;
}

bool s3i::freeRetired(int dev)
{
    std::vector<void *> mine;
    {
        std::lock_guard<std::mutex> g(g_retired.mu);
        auto keep = g_retired.list.begin();
        for (auto &e : g_retired.list) { if (e.first == dev) mine.push_back(e.second); else *keep++ = e; }
        g_retired.list.erase(keep, g_retired.list.end());
        if (dev >= 0 && dev < 64) g_retired.bytes[dev] = 0;
    }
    for (void *p : mine) (void)hipFree(p);
    return !mine.empty();
}

void s3i::retireBuffer(int dev, void *p)
{
    if (!p) return;
    size_t bytes = 0;
    if (hipMemPtrGetInfo(p, &bytes) != hipSuccess) { (void)hipGetLastError(); bytes = 0; }
    bool full = false;
    {
        std::lock_guard<std::mutex> g(g_retired.mu);
        g_retired.list.emplace_back(dev, p);
        if (dev >= 0 && dev < 64) full = (g_retired.bytes[dev] += bytes) > ((size_t)8 << 30);
    }
    if (full) (void)freeRetired(dev);
}

hipError_t s3i::mallocOrRelease(int dev, void **q, size_t bytes)
{
    // (tests: S3DP_TEST_OOM_EVERY=n makes every n-th allocation of the process find the device full at its first attempt)
    static const size_t oomEvery = envSize("S3DP_TEST_OOM_EVERY", 0);
    static std::atomic<size_t> calls{0};
    bool pretend = oomEvery && (++calls % oomEvery) == 0;
    if (pretend) (void)releaseFlankTable(dev);                          // (pretended: a table goes, if there is one, and the allocation is made for real)
    for (;;) {
        const hipError_t e = hipMalloc(q, bytes);
        if (e != hipErrorOutOfMemory) return e;
        (void)hipGetLastError();
        if (freeRetired(dev)) continue;                                 // (buffers that batches outgrew, kept so that nobody waited for hipFree)
        if (!releaseFlankTable(dev)) return e;
    }
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
    // S3DP_BLOCKING_SYNC=1: host threads that wait for the device sleep instead of spinning -- under a CPU quota (a container granted sixteen CPUs) every
    // spinning waiter is a CPU the parser and the formatter do not get.  Has to be said before the device's context exists: failure is ignored.
    if (const char *e = getenv("S3DP_BLOCKING_SYNC")) if (*e == '1') (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    (void)hipGetLastError();
    s3_index *ix = new s3_index();
    ix->dev = device;
    hipStream_t s = nullptr;
    hipError_t es = hipStreamCreate(&s);
    if (es != hipSuccess) { delete ix; return fail(S3_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(es)); }
    // whatever goes wrong below: the object, its device memory (up to 120 GB with the 16-mer tables) and the stream are released
    // (the code objects of the library's kernels are loaded meanwhile: s3dp_kernels.hip, preloadKernels; S3DP_PRELOAD_KERNELS=0: at their first launch, as the runtime does)
    std::thread preload;
    if (envSize("S3DP_PRELOAD_KERNELS", 1)) preload = std::thread([device] { if (hipSetDevice(device) == hipSuccess) preloadKernels(); (void)hipGetLastError(); });
    const int rc = indexUploadBody(ix, desc, s);
    if (preload.joinable()) preload.join();
    (void)hipStreamDestroy(s);
    if (rc) { s3_index_free(ix); return rc; }
    { std::lock_guard<std::mutex> g(g_indexesMu); g_indexes.push_back(ix); }
    *outp = ix;
    return S3_OK;
}

static int indexUploadBody(s3_index *ix, const s3_index_desc *desc, hipStream_t s)
{
    // S3DP_VERBOSE=1: where the upload's time goes (stderr)
    static const bool verbose = getenv("S3DP_VERBOSE") != nullptr;
    auto tUp = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!verbose) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[s3dp] index upload: %-28s %8.3f s\n", what, std::chrono::duration<double>(now - tUp).count());
        tUp = now;
    };
    int rc = uploadHalf(ix, desc->fwd, &ix->d.fwd, s);
    if (!rc) rc = uploadHalf(ix, desc->rev, &ix->d.rev, s);
    if (rc) return rc;
    lap("BWT + occurrence, both");
    const uint32_t n = desc->textLength;
    uint32_t *sa = nullptr, *pac = nullptr, *lkt = nullptr, *rlkt = nullptr;
    const size_t ltWords = (size_t)1 << (2 * desc->lktSize);
    // + zero tail: 3-word windows never leave the buffer, and the second search pass reads the text in whole 64-byte lines up to a read length past the end
    const size_t pacWords = (desc->packedWords + 15) / 16 * 16 + 16 + S3_MAX_READ_LENGTH / 16;
    if ((rc = devAlloc(ix, &sa, (size_t)n + 1)) || (rc = devAlloc(ix, &pac, pacWords)) || (rc = devAlloc(ix, &lkt, ltWords)) ||
        (rc = devAlloc(ix, &rlkt, ltWords))) return rc;
    const uint32_t saInterval = desc->saInterval > 1 ? desc->saInterval : 1;
    DevTmp samplesHold;
    uint32_t *saSamples = nullptr;
    {
        // the suffix array in pieces: as one 12 GB copy from pageable memory it takes the runtime's staging path (3.4 GB/s measured); pieces of this size
        // are copied in place at PCIe speed.  Upload of a 3.1 Gbp index: 6.1 -> 3.9 s (S3DP_VERBOSE=1 prints the laps; S3DP_H2D_PIECE_MB to change)
        const size_t entries = saInterval > 1 ? ((size_t)n + saInterval) / saInterval : (size_t)n + 1;
        uint32_t *dst = sa;
        if (saInterval > 1) {
            HIPCHK(hipMalloc(&samplesHold.p, entries * 4));
            dst = saSamples = (uint32_t *)samplesHold.p;
        }
        const size_t total = entries * 4, pieceMb = envSize("S3DP_H2D_PIECE_MB", 128), piece = pieceMb ? pieceMb << 20 : total;
        for (size_t off = 0; off < total; off += piece)
            HIPCHK(hipMemcpyAsync((char *)dst + off, (const char *)desc->sa + off, total - off < piece ? total - off : piece, hipMemcpyHostToDevice, s));
        if (saInterval > 1) {
            // a sampled array (SaValueFreq 2 / 4) is expanded here, once: every hit is then one access instead of a walk through the BWT
            launchSaExpand(ix->d.fwd, saSamples, saInterval, n, sa, s);
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipMemsetAsync(pac, 0, pacWords * 4, s));
    HIPCHK(hipMemcpyAsync(pac, desc->packedDNA, desc->packedWords * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(lkt, desc->lkt, ltWords * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(rlkt, desc->revLkt, ltWords * 4, hipMemcpyHostToDevice, s));
    lap("SA, text, 13-mer tables");
    ix->d.sa = sa; ix->d.pac = pac; ix->d.pacWords = pacWords; ix->d.lkt = lkt; ix->d.rlkt = rlkt;
    ix->d.ltSize = desc->lktSize; ix->d.n = n;
    ix->d.ltBack = nullptr; ix->d.ltBi = nullptr; ix->d.ltK = 0;
    ix->d.flankBack = nullptr; ix->d.flankFwd = nullptr;
    {
        // padded k-mers of the two tables (see DevIndex): text[n-j .. n-1] + 'A' x (K-j) and, for the reverse table, text[j-1 .. 0] + 'A' x (K-j), j = 1 .. K-1
        const int K = desc->lktSize;
        auto ch = [&](uint64_t p) -> uint32_t { return (desc->packedDNA[p >> 4] >> (2 * (15 - (p & 15)))) & 3u; };
        auto rev = [&](uint32_t key) { uint32_t r = 0; for (int j = 0; j < K; j++) r |= ((key >> (2 * j)) & 3u) << (2 * (K - 1 - j)); return r; };
        for (int j = 1; j <= 12; j++) {
            uint32_t kf = 0xFFFFFFFFu, kr = 0xFFFFFFFFu;                     // unused slots never match (keys have at most 26 bits)
            if (j < K && (uint32_t)j <= n) {
                kf = 0; kr = 0;
                for (int q = 0; q < j; q++) { kf = (kf << 2) | ch((uint64_t)n - j + q); kr = (kr << 2) | ch((uint64_t)(j - 1 - q)); }
                kf <<= 2 * (K - j); kr <<= 2 * (K - j);
            }
            ix->d.padBack[j - 1] = kf;
            ix->d.padBi[j - 1] = kr;
            ix->d.padBi[12 + j - 1] = kf == 0xFFFFFFFFu ? kf : rev(kf);
        }
    }
    {
        // longer lookup tables in HBM: S3DP_LT_EXTRA = 0 .. 3 characters on top of the file format's 13-mer table.  Default: 3 (16-mers, 103 GB) for texts
        // of two gigabases and more -- 288 GB of HBM are there to be used and every character saves two rank queries per item --, 2 (15-mers, 25.8 GB)
        // from one gigabase, 0 below, where a run is too short to pay for the allocation.  Out of memory: one character less.
        HIPCHK(hipStreamSynchronize(s));
        const size_t deflt = n >= (1u << 31) ? 3 : n >= (1u << 30) ? 2 : 0;
        int extra = (int)envSize("S3DP_LT_EXTRA", deflt);
        if (extra > 3) extra = 3;
        if (desc->lktSize + extra > 16) extra = 16 - desc->lktSize;
        for (; extra > 0; extra--) {
            uint2 *tb = nullptr;
            uint4 *ti = nullptr;
            const int e = s3::launchLongLookup(ix->d, extra, &tb, &ti, s);
            if (e == 0) {
                ix->allocs.push_back(tb); ix->allocs.push_back(ti);
                ix->d.ltBack = tb; ix->d.ltBi = ti; ix->d.ltK = desc->lktSize + extra;
                ix->bytes += ((size_t)1 << (2 * ix->d.ltK)) * 24;
                break;
            }
            (void)hipGetLastError();
            if (e != (int)hipErrorOutOfMemory) { return fail(S3_EHIP, std::string("long lookup tables: ") + hipGetErrorString((hipError_t)e)); }
        }
    }
    lap("long lookup tables");
    {
        // flank tables of check-and-extend (s3dp_flank.hip): 16 B per suffix-array entry and table -- 2 x 49.6 GB at 3.1 G bases, beside 120 GB of index with the
        // 16-mer tables.  S3DP_FLANK = 0: none, 1: the forward index's only, 2 (default): both; they are built while S3DP_FLANK_RESERVE_GB (default 24) of device
        // memory stay free for batches and aligners -- a table that does not fit is left out and its tasks are verified from the suffix array and the text.
        const int which = (int)envSize("S3DP_FLANK", 2);
        void *fa[2] = {nullptr, nullptr};
        int nfa = 0;
        size_t fbytes = 0;
        const int e = s3::launchFlankTables(ix->d, which > 2 ? 2 : which, envSize("S3DP_FLANK_RESERVE_GB", 24) << 30, fa, &nfa, &fbytes, verbose, s);
        for (int k = 0; k < nfa; k++) ix->allocs.push_back(fa[k]);
        ix->bytes += fbytes;
        if (e != (int)hipSuccess) return fail(S3_EHIP, std::string("flank tables: ") + hipGetErrorString((hipError_t)e));
    }
    lap("flank tables");
    // model tables for every read length and mismatch budget (built on the host, see s3dp_model.cpp)
    std::vector<S3Model> host(S3_MAX_READ_LENGTH + 1);
    for (int e = 0; e <= S3_MAX_MISMATCH; e++) {
        memset(host.data(), 0, host.size() * sizeof(S3Model));
        for (int L = 1; L <= S3_MAX_READ_LENGTH; L++) s3_model_build(L, e, &host[L]);
        if ((rc = devAlloc(ix, &ix->models[e], host.size()))) return rc;
        HIPCHK(hipMemcpyAsync(ix->models[e], host.data(), host.size() * sizeof(S3Model), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    HIPCHK(hipStreamSynchronize(s));
    lap("model tables");
    return S3_OK;
}

extern "C" void s3_index_free(s3_index *ix)
{
    if (!ix) return;
    { std::lock_guard<std::mutex> g(g_indexesMu); for (size_t k = 0; k < g_indexes.size(); k++) if (g_indexes[k] == ix) { g_indexes.erase(g_indexes.begin() + k); break; } }
    (void)hipSetDevice(ix->dev);
    s3i::scratchTrim(ix->dev);
    for (void *p : ix->allocs) if (p) (void)hipFree(p);
    delete ix;
}

extern "C" size_t s3_index_device_bytes(const s3_index *ix) { return ix ? ix->bytes : 0; }
extern "C" int s3_index_lookup_chars(const s3_index *ix) { return !ix ? 0 : ix->d.ltK ? ix->d.ltK : ix->d.ltSize; }
extern "C" int s3_index_flank_tables(const s3_index *ix) { return !ix ? 0 : (ix->d.flankBack ? 1 : 0) + (ix->d.flankFwd ? 1 : 0); }     // (a table can be released later: s3_index::flankMu)

extern "C" int s3_index_fetch_flank(s3_index *ix, int table, uint32_t first, uint32_t count, uint32_t *out)
{
    if (!ix || !out) return fail(S3_EINVAL, "null argument");
    std::shared_lock<std::shared_mutex> r(ix->flankMu);
    const uint4 *t = table == 0 ? ix->d.flankBack : table == 1 ? ix->d.flankFwd : nullptr;
    if (!t) return fail(S3_EINVAL, "s3_index_fetch_flank: the index holds no such table");
    if ((uint64_t)first + count > (uint64_t)ix->d.n + 1u) return fail(S3_EINVAL, "s3_index_fetch_flank: beyond the table");
    HIPCHK(hipSetDevice(ix->dev));
    HIPCHK(hipMemcpy(out, t + first, (size_t)count * sizeof(uint4), hipMemcpyDeviceToHost));
    return S3_OK;
}

// ---------------------------------------------------------------------------------------------


int s3i::growRecords(s3_batch *b)
{
    if (b->recCap >= kMaxEntries) return fail(S3_EOVERFLOW, "record buffer overflow at the 32-bit limit");
    size_t cap = (size_t)b->recCap * 2;
    if (cap > kMaxEntries) cap = kMaxEntries;
    int rc;
    if ((rc = bgrow(b, &b->dRecs, cap, 0))) return rc;
    if ((rc = bgrow(b, &b->dRecAux, cap, 0))) return rc;
    if (b->dHits) { if ((rc = bgrow(b, &b->dHits, cap, 0))) return rc; }
    b->recCap = (uint32_t)cap;
    return S3_OK;
}

int s3i::growOccurrences(s3_batch *b, uint64_t need)
{
    if (need > kMaxEntries) return fail(S3_EOVERFLOW, "occurrence buffer overflow at the 32-bit limit");
    size_t cap = (size_t)b->occCap * 2;
    if (cap < need + need / 4) cap = need + need / 4;
    if (cap > kMaxEntries) cap = kMaxEntries;
    int rc;
    if ((rc = bgrow(b, &b->dOccPos, cap, b->occUsed))) return rc;
    if ((rc = bgrow(b, &b->dOccMeta, cap, b->occUsed))) return rc;
    b->occCap = (uint32_t)cap;
    return S3_OK;
}

// the read-ordered copy of the final lists (dOutPos / dOutMeta) is sized by what the ladder KEEPS, not by what its passes decoded on the way: in long-read
// mode the seed search runs without a hit limit and a read inside a satellite array decodes a hundred times what the verification keeps of it
int s3i::ensureOutLists(s3_batch *b, uint64_t need)
{
    if (need <= b->outCap) return S3_OK;
    if (need > kMaxEntries) return fail(S3_EOVERFLOW, "occurrence lists overflow at the 32-bit limit");
    size_t cap = need + need / 4 + 65536;
    if (cap > kMaxEntries) cap = kMaxEntries;
    int rc;
    b->outCap = 0;
    if ((rc = bgrow(b, &b->dOutPos, cap, 0))) return rc;
    if ((rc = bgrow(b, &b->dOutMeta, cap, 0))) return rc;
    b->outCap = cap;
    return S3_OK;
}

void s3i::noteAlloc(const char *what, size_t bytes, std::chrono::steady_clock::time_point t0)
{
    static const bool verbose = getenv("S3DP_VERBOSE") != nullptr;
    if (!verbose || bytes < (1u << 20)) return;
    static const auto first = std::chrono::steady_clock::now();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[s3dp] %9.3f s  %s %8.1f MB  %7.2f ms\n", std::chrono::duration<double>(now - first).count(), what, bytes / 1048576.0,
            std::chrono::duration<double, std::milli>(now - t0).count());
}

bool s3i::growRefused()
{
    static std::atomic<long> left{[] { const char *e = getenv("S3DP_TEST_FAIL_GROW"); return e ? atol(e) : 0L; }()};
    if (left.load() <= 0) return false;
    return left.fetch_sub(1) == 1;
}

size_t s3i::envSize(const char *name, size_t dflt)
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
    // (32 record slots and 40 occurrences per read to begin with: a repeat-rich genome needs about that, and a batch that outgrows its buffers in the middle
    //  of a run pays a hipMalloc + hipFree -- a device-wide synchronisation -- and the repeat of its pass; 0.9 GB per 1 Mi reads is nothing of 288 GB)
    const size_t recPerRead = envSize("S3DP_REC_PER_READ", 32), occPerRead = envSize("S3DP_OCC_PER_READ", 40);
    // every wave of the persistent search grid may hold one partly filled block of record slots: that is the floor.  Both buffers
    // grow on demand (growRecords / growOccurrences); the sizes here only avoid regrowth in the common case.
    const size_t recFloor = envSize("S3DP_REC_FLOOR", 65536 + (size_t)S3_V2_REC_BLOCK * 8192);
    size_t recCap = (size_t)maxReads * recPerRead + recFloor, occCap = (size_t)maxReads * occPerRead + 65536;
    if (recCap < 4096) recCap = 4096;
    if (recCap > 0xFFFFFF00ull) recCap = 0xFFFFFF00ull;
    if (occCap > 0xFFFFFF00ull) occCap = 0xFFFFFF00ull;
    b->recCap = (uint32_t)recCap; b->occCap = (uint32_t)occCap;
    // from here on every error return goes through s3_batch_free
#define HIPB(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { s3_batch_free(b); return fail(S3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    HIPB(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    int rc;
#define BA(p, cnt) if ((rc = balloc(b, &b->p, (cnt)))) { s3_batch_free(b); return rc; }
    BA(dQueries, roundUp * wordPerQuery); BA(dLens, roundUp); BA(dSeedLens, roundUp); BA(dReadsT, roundUp * 2 * wordPerQuery + 16);
    BA(dRecs, recCap); BA(dRecAux, recCap); BA(dOrdIdx, recCap); BA(dOrdValid, recCap + 8); b->ordCap = recCap;     // (+ 8: the long merge / decode kernels read the bytes eight at a time)
    BA(dRecCount, 4); BA(dFlags, 4); BA(dCounters, S3_COUNTER_WORDS); BA(dWorkHead, 2); BA(dTaskRange, 4); BA(dSum64, 2);
    BA(dItemCounts, b->maxItems + 1); BA(dItemOffsets, b->maxItems + 1);
    BA(dWasHeavy, (size_t)maxReads + 64); BA(dHeavyFlags, (size_t)maxReads + 64); BA(dHeavyList, (size_t)maxReads + 64); BA(dHeavyCount, 4); BA(dHeavyIndex, (size_t)maxReads + 64);
    BA(dLongList, (size_t)maxReads + 64); BA(dLongCount, 4); BA(dTrunk, 16); b->trunkCap = 16; BA(dItemAcc, 16); b->accCap = 16;
    b->countCap = b->maxItems + 1;
    BA(dKept, (size_t)maxReads + 1); BA(dOccCounts, (size_t)maxReads + 1); BA(dCeCounts, (size_t)maxReads + 1); BA(dTotalOcc, (size_t)maxReads + 1);
    BA(dWithError, (size_t)maxReads * 8);
    BA(dHitOffsets, (size_t)maxReads + 1);                                 // (dHits: on the first s3_batch_search -- the ladders never need it)
    BA(dActive[0], maxReads); BA(dActive[1], maxReads); BA(dNumSel, 4); BA(dIota, maxReads);
    BA(dState, maxReads); BA(dSelFlags, maxReads);
    BA(dOccOffsets, (size_t)maxReads + 1); BA(dMasked, (size_t)maxReads + 1);
    BA(dOccPos, occCap); BA(dOccMeta, occCap);
    BA(dPhase, maxReads); BA(dFinalWithError, (size_t)maxReads * 8); BA(dOccStart, maxReads); BA(dOccCount, (size_t)maxReads + 1);
    BA(dOutOffsets, (size_t)maxReads + 1); BA(dOutPos, occCap); BA(dOutMeta, occCap); b->outCap = occCap;
#undef BA
    // scan / select scratch sized for the largest problem
    size_t t1 = 0, t2 = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t1, b->dItemCounts, b->dItemOffsets, (int)(b->maxItems + 1), b->stream);
    (void)hipcub::DeviceSelect::Flagged(nullptr, t2, b->dIota, b->dSelFlags, b->dActive[0], b->dNumSel, (int)maxReads, b->stream);
    {
        size_t t3 = 0;
        hipcub::TransformInputIterator<unsigned long long, U32ToU64, const uint32_t *> it(b->dItemCounts, U32ToU64());
        (void)hipcub::DeviceReduce::Sum(nullptr, t3, it, b->dSum64, (int)(b->maxItems + 1), b->stream);
        if (t3 > t1) t1 = t3;
    }
    b->tempBytes = (t1 > t2 ? t1 : t2) + 256;
    if ((rc = balloc(b, &b->dTemp, b->tempBytes))) { s3_batch_free(b); return rc; }
    HIPB(hipEventCreate(&b->evTotal[0])); HIPB(hipEventCreate(&b->evTotal[1]));
    b->evSearch.assign(512, nullptr);
    b->evSearchRole.assign(256, 0);
    for (auto &e : b->evSearch) HIPB(hipEventCreate(&e));
    b->evCe.assign(256, nullptr);
    for (auto &e : b->evCe) HIPB(hipEventCreate(&e));
    HIPB(hipMemsetAsync(b->dCounters, 0, S3_COUNTER_WORDS * 8, b->stream));
    HIPB(hipStreamSynchronize(b->stream));
#undef HIPB
    *outp = b;
    return S3_OK;
}

extern "C" void s3_batch_free(s3_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->ix->dev);
    if (b->stream) { (void)hipStreamSynchronize(b->stream); s3i::scratchStreamGone(b->stream); }
    for (void *p : b->allocs) (void)hipFree(p);
    (void)s3i::freeRetired(b->ix->dev);                                // (what this and other batches of the device outgrew: unused since the day it was retired)
    for (auto e : b->evSearch) if (e) (void)hipEventDestroy(e);
    for (auto e : b->evCe) if (e) (void)hipEventDestroy(e);
    for (auto e : b->evTotal) if (e) (void)hipEventDestroy(e);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

extern "C" int s3_batch_upload(s3_batch *b, const uint32_t *queries, const uint32_t *readLengths, uint32_t n)
{
    if (!b || !queries || !readLengths) return fail(S3_EINVAL, "s3_batch_upload: null argument");
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
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
    HIPCHK(hipMemsetAsync(b->dWasHeavy, 0, b->maxReads, b->stream));
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
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
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
    HIPCHK(d2h(lens.data(), b->dLens, (size_t)n * 4, b->stream));
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

extern "C" int s3_batch_set_grid(s3_batch *b, unsigned blocksPerCu)
{
    if (!b) return fail(S3_EINVAL, "null batch");
    if (blocksPerCu > 16u) return fail(S3_EINVAL, "s3_batch_set_grid: at most 16 workgroups per compute unit");
    b->blocksPerCu = blocksPerCu;
    return S3_OK;
}

extern "C" int s3_batch_enable_counters(s3_batch *b, int on)
{
    if (!b) return fail(S3_EINVAL, "null batch");
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    b->counting = on != 0;
    return S3_OK;
}

// The sort of (64-bit key, 32-bit slot) pairs.  rocPRIM's default Onesweep configuration for this shape on gfx950 (no tuned entry: 256 x 12 items, rank by match)
// indexes a private array dynamically -- 80 bytes of SCRATCH per lane, and the first launch of a kernel with scratch on a HIP stream makes the runtime
// reallocate the device's scratch buffer with every queue drained: 60 ms with the device idle, once per slot of a stream of batches (profiles/README_r06.md
// section 8).  512 x 6 items keeps the array in registers (51 VGPRs, no scratch; compile with -Rpass-analysis=kernel-resource-usage to see).
using ListedSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<512, 6>, 8,
                                                                                        rocprim::block_radix_rank_algorithm::match>>;
static hipError_t sortListed(void *temp, size_t &bytes, const unsigned long long *keysIn, unsigned long long *keysOut, const uint32_t *valsIn, uint32_t *valsOut,
                             uint32_t n, hipStream_t s)
{
    return rocprim::radix_sort_pairs<ListedSortConfig>(temp, bytes, keysIn, keysOut, valsIn, valsOut, n, 0, 64, s);
}

// slots (nullptr: 0 .. n-1) ordered by the first 32 characters of their reads; the ordered slots are left in b->dOrderVals[1] (s3dp_unit.hip, "LOCALITY")
int s3i::orderByReadPrefix(s3_batch *b, const uint32_t *slots, uint32_t n, const uint32_t *active, uint32_t stride, uint32_t words)
{
    if (n > b->orderCap) {
        const size_t cap = (size_t)n + n / 4 + 1024;
        int rc;
        b->orderCap = 0;
        for (int k = 0; k < 2; k++) {
            if (b->dOrderKeys[k]) { if ((rc = bgrow(b, &b->dOrderKeys[k], cap, 0))) return rc; } else if ((rc = balloc(b, &b->dOrderKeys[k], cap))) return rc;
            if (b->dOrderVals[k]) { if ((rc = bgrow(b, &b->dOrderVals[k], cap, 0))) return rc; } else if ((rc = balloc(b, &b->dOrderVals[k], cap))) return rc;
        }
        b->orderCap = cap;
    }
    size_t tb = 0;
    HIPCHK(sortListed(nullptr, tb, b->dOrderKeys[0], b->dOrderKeys[1], b->dOrderVals[0], b->dOrderVals[1], n, b->stream));
    if (tb + 256 > b->tempBytes) {
        int rc = bgrow(b, &b->dTemp, tb + 256, 0);
        if (rc) { b->tempBytes = 0; return rc; }
        b->tempBytes = tb + 256;
    }
    launchListedKeys(slots, n, active, b->dReadsT, stride, words, b->dOrderKeys[0], b->dOrderVals[0], b->stream);
    tb = b->tempBytes;
    HIPCHK(sortListed(b->dTemp, tb, b->dOrderKeys[0], b->dOrderKeys[1], b->dOrderVals[0], b->dOrderVals[1], n, b->stream));
    return S3_OK;
}

// search + order + merge for `nActive` slots.  Leaves per-slot results in dTotalOcc / dWithError / dKept / ...
int s3i::searchPass(s3_batch *b, const uint32_t *active, uint32_t nActive, uint32_t modelMm, uint32_t maxOut)
{
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    const int nCases = s3_num_cases((int)modelMm);
    const size_t nItems = (size_t)nActive * nCases * 2;
    SearchIn in{};
    in.queries = b->dQueries; in.readsT = b->dReadsT; in.readLengths = b->longMode ? b->dSeedLens : b->dLens; in.active = active; in.nActive = nActive;
    in.wpq = b->wpq;
    in.wpqLoad = (b->longMode && b->wpq > 8) ? 8 : b->wpq;                // seeds have at most 128 bases
    in.wLds = b->uniformLen ? (b->uniformLen + 15u) / 16u : in.wpqLoad;   // LDS per wave decides how many waves a CU holds: no padding words there
    if (in.wLds > in.wpqLoad) in.wLds = in.wpqLoad;
    in.models = b->ix->models[modelMm]; in.nCases = nCases; in.maxOut = maxOut; in.uniformLen = b->uniformLen;
    {
        // S3DP_EARLY_CE=0: ranges walk their exact lookup step to the end, as the CPU path does (same results, more rank queries); n: ranges of up to n
        // suffixes leave it for check-and-extend at once (default 1)
        const uint32_t early = (uint32_t)envSize("S3DP_EARLY_CE", 1);                           // largest range (suffixes) handed over early
        in.earlyCe = b->ix->d.ltSize <= 13 ? early : 0u;                       // the list of padded keys (DevIndex::padBack / padBi) holds twelve per table
        const uint32_t earlyMin = (uint32_t)envSize("S3DP_EARLY_MIN", 0);
        in.earlyMinI = earlyMin;
    }
    {
        S3Model m;
        s3_model_build(100, (int)modelMm, &m);
        in.nStepsPacked = 0;
        in.sStarPacked = 0;
        for (int c = 0; c < m.nCases; c++) {
            in.nStepsPacked |= (uint32_t)m.cs[c].nSteps << (3 * c);
            int sStar = m.cs[c].nSteps;                                    // the first step that allows a mismatch (depends on the budget only)
            for (int k = m.cs[c].nSteps - 1; k >= 0; k--) if (m.cs[c].st[k].maxErr > 0) sStar = k;
            in.sStarPacked |= (uint32_t)sStar << (3 * c);
        }
    }
    in.unitsPerItem = 1; in.unitBase = 0; in.unitMode = 0; in.unitGroup = 1;
    in.adoptIdle = (uint32_t)envSize("S3DP_ADOPT_IDLE", S3_UNIT_ADOPT_IDLE);
    in.takeMin = (uint32_t)envSize("S3DP_TAKE_MIN", S3_UNIT_TAKE_MIN);
    if (in.takeMin < 1u) in.takeMin = 1u;
    if (in.takeMin > 64u) in.takeMin = 64u;
    in.tailAge = (uint32_t)envSize("S3DP_TAIL_AGE", S3_UNIT_TAIL_AGE);
    in.giantAge = (uint32_t)envSize("S3DP_GIANT_AGE", S3_UNIT_GIANT_AGE);
    in.giantIdle = (uint32_t)envSize("S3DP_GIANT_IDLE", S3_UNIT_GIANT_IDLE);
    if (in.giantIdle < 1u) in.giantIdle = 1u;
    in.fastTrips = (uint32_t)envSize("S3DP_FAST_TRIPS", S3_UNIT_FAST_TRIPS);
    in.fastMin = (uint32_t)envSize("S3DP_FAST_MIN", S3_UNIT_FAST_MIN);
    if (in.fastMin < 1u) in.fastMin = 1u;
    in.firstFastTrips = (uint32_t)envSize("S3DP_FIRST_FAST_TRIPS", S3_FIRST_FAST_TRIPS);
    in.firstFastMin = (uint32_t)envSize("S3DP_FIRST_FAST_MIN", S3_FIRST_FAST_MIN);
    if (in.firstFastMin < 1u) in.firstFastMin = 1u;
    uint32_t slots = 0;                                                   // entries of the ordered array this pass needs
    unsigned long long counterSnapAll[S3_COUNTER_WORDS] = {}, counterSnap[S3_NUM_COUNTERS] = {};   // per kernel role; summed over the roles
    if (b->counting) {
        HIPCHK(hipMemcpyAsync(counterSnapAll, b->dCounters, S3_COUNTER_WORDS * 8, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        for (int r = 0; r < S3_NUM_ROLES; r++) for (int k = 0; k < S3_NUM_COUNTERS; k++) counterSnap[k] += counterSnapAll[r * S3_NUM_COUNTERS + k];
    }
    // S3DP_HEAVY=0 switches the second pass off (every read is walked by one lane: same results; reads in repeat families then hold the launch up);
    // n > 1: list a read once it has reserved more than n candidates (default 2048); S3DP_HEAVY_RANK: ... or cost more than that many rank queries
    // (default 1024: dead ends cost steps, not candidates).  Round 3, with the unit pass twice as fast: 2048 / 1024 / give-up 2 is 6 % faster than round 2's
    // 1024 / 4096 / 4 on the benchmark genome and no slower on the other repeat models (profiles/r03_threshold_sweeps.txt)
    const size_t heavyEnv = envSize("S3DP_HEAVY", 1);                       // read per pass: tests switch it
    // (defaults by the size of the batch: what a read costs before it is listed is a floor under the duration of a first pass -- 1024 rank queries are 2.3 ms --
    //  which the ladder of ten million reads does not feel and the ladder of one million does: 1 Mi-read batch 81.2 -> 75.9 ms per ladder with 1024 / 512,
    //  tools/r4_small_sweep.sh; the 10 M batch is best at 2048 / 1024, profiles/r03_threshold_sweeps.txt)
    const bool smallPass = b->n <= (2u << 20);                                // (by the BATCH: the later phases of a large batch are small passes and still best at 2048 / 1024)
    in.heavyLimit = heavyEnv == 0 ? 0xFFFFFFFFu : (heavyEnv > 1 ? (uint32_t)heavyEnv : (smallPass ? 1024u : 2048u));
    in.heavyRank = heavyEnv == 0 ? 0xFFFFFFFFu : (uint32_t)envSize("S3DP_HEAVY_RANK", heavyEnv > 1 && heavyEnv < 64 ? 64 : (smallPass ? 512 : 1024));
    in.heavyLevel = modelMm + 1u;
    const bool heavyMemo = envSize("S3DP_HEAVY_MEMO", 1) != 0;              // a read listed in one phase of a ladder is listed at once in the next (s3dp_first.hip, F_LOAD)
    in.slotList = nullptr; in.slotCount = nullptr;
    // units of the second pass: first-mismatch positions of an item's first mismatch step, + 1 for "none" -- the longest such step over the cases
    // and over the read lengths this batch can hold (S3DP_UNITS=0: whole items)
    uint32_t unitsPerItem = 1;
    if (in.heavyLimit != 0xFFFFFFFFu && envSize("S3DP_UNITS", 1) != 0) {
        const uint32_t lo = b->uniformLen ? b->uniformLen : 16u, hi = b->uniformLen ? b->uniformLen : std::min<uint32_t>(in.wpqLoad * 16u, S3_MAX_READ_LENGTH);
        bool ok = true;
        int longest = 0;
        for (uint32_t L = lo; L <= hi && ok; L++) {
            S3Model m;
            s3_model_build((int)L, (int)modelMm, &m);
            for (int c = 0; c < m.nCases && ok; c++) {
                const int sStar = (int)((in.sStarPacked >> (3 * c)) & 7u);
                if (sStar >= m.cs[c].nSteps) continue;
                if (sStar == 0) { ok = false; break; }                      // mismatches inside the lookup step: positions the lookup swallows cannot be units
                const S3Step &st = m.cs[c].st[sStar];
                const int lenEff = st.minErr > 1 ? st.len - (st.minErr - 1) : st.len;
                if (lenEff > longest) longest = lenEff;
            }
        }
        // (round 5: a unit is a GROUP of consecutive positions -- a unit per position is seven steps long with three and four mismatches, and starting it costs
        //  as much as walking it; s3dp_unit.hip, "GROUPS")
        // Two for the single-end ladders (10 M x 100 bp: unit passes 111.4 -> 93.6 ms), one for the paired ladder and the seeds of the DP stages: the same
        // pairs on the same box took 2.93 s with groups of two and 2.70 s with one position per unit (search kernels 1.36 against 1.19 s; README_r05 9.7)
        // (round 6: with three and four mismatches a read rarely reaches MaxOutputPerRead -- the cut's coarseness costs nothing there -- and its units are many
        //  and short: S3DP_UNIT_GROUP_HI positions per unit from three mismatches on)
        in.unitGroup = (uint32_t)envSize("S3DP_UNIT_GROUP", b->unitGroup ? b->unitGroup : 1u);
        if (modelMm >= 3u && b->unitGroup > 1u) in.unitGroup = (uint32_t)envSize("S3DP_UNIT_GROUP_HI", envSize("S3DP_UNIT_GROUP", S3_UNIT_GROUP_HI));
        if (in.unitGroup < 1u) in.unitGroup = 1u;
        if (ok && longest > 0) unitsPerItem = ((uint32_t)longest + in.unitGroup - 1u) / in.unitGroup + 1u;
    }
    uint32_t nHeavy = 0, nCount = (uint32_t)nItems;                         // entries of itemCounts in use: the items, then the units of the listed reads
    for (;;) {
        HIPCHK(hipMemsetAsync(b->dRecCount, 0, 4, b->stream));
        HIPCHK(hipMemsetAsync(b->dHeavyCount, 0, 4, b->stream));
        HIPCHK(hipMemsetAsync(b->dHeavyFlags, 0, nActive, b->stream));
        SearchOut out{};
        out.recs = b->dRecs; out.recAux = b->dRecAux; out.recCap = b->recCap; out.recCount = b->dRecCount; out.itemCounts = b->dItemCounts; out.flags = b->dFlags;
        out.counters = b->counting ? b->dCounters : nullptr;
        out.workHead = b->dWorkHead;
        out.cutDummy = b->dFlags + 2;
        out.heavyFlags = b->dHeavyFlags; out.heavyList = b->dHeavyList; out.heavyCount = b->dHeavyCount; out.heavyIndex = b->dHeavyIndex;
        out.wasHeavy = in.heavyLimit != 0xFFFFFFFFu && modelMm > 0 && heavyMemo ? b->dWasHeavy : nullptr;
        nHeavy = 0; nCount = (uint32_t)nItems;
        b->scatterDyn = ScatterDyn{nullptr, nullptr, 0, 0};
        HIPCHK(hipMemsetAsync(b->dWorkHead, 0, 8, b->stream));
        // timing: one event pair per group of stepping launches / per check-and-extend launch (they alternate when the second pass runs in rounds)
        // (role: which kernel the pair brackets -- 0 first pass, 1 trunk pass, 2 unit pass, 3 check-and-extend: s3_batch_kernel_stats)
        auto mark = [&](std::vector<hipEvent_t> &ev, uint32_t &used, bool begin, uint8_t role) -> hipError_t {
            if ((size_t)used * 2 + 1 >= ev.size()) return hipSuccess;
            const hipError_t e = hipEventRecord(ev[used * 2 + (begin ? 0 : 1)], b->stream);
            if (&ev == &b->evSearch) b->evSearchRole[used] = role;
            if (!begin) used++;
            return e;
        };
        auto countersOf = [&](int role) { return b->counting ? b->dCounters + role * S3_NUM_COUNTERS : nullptr; };
        out.counters = countersOf(0);
        HIPCHK(mark(b->evSearch, b->evSearchUsed, true, 0));
        {
            // persistent grid: what is resident at once (4 blocks per CU), or less when several batches share the device (the streamed form: a
            // narrower grid per batch lets the kernels of two batches run side by side and fill each other's tails)
            const unsigned blocksPerCuEnv = (unsigned)envSize("S3DP_BLOCKS_PER_CU", 0);
            const unsigned blocksPerCu = blocksPerCuEnv ? blocksPerCuEnv : (b->blocksPerCu ? b->blocksPerCu : S3_FIRST_WAVES_PER_SIMD);
            // S3DP_VERBOSE=1 with counters enabled: reads, listed reads, rank queries and milliseconds of every launch of this pass (stderr)
            static const bool verbose = getenv("S3DP_VERBOSE") != nullptr;
            unsigned long long rankSeen = counterSnap[0], tripsSeen = counterSnap[6], wavesSeen = counterSnap[9], adoptSeen = counterSnap[8], saSeen = counterSnap[2], ceSeen = counterSnap[3], accSeen = counterSnap[10];
            unsigned long long waveTicksSeen = counterSnap[19], wavesCountSeen = counterSnap[20];
            unsigned long long dSeen[8] = {counterSnap[11], counterSnap[12], counterSnap[13], counterSnap[14], counterSnap[15], counterSnap[16], counterSnap[17], counterSnap[18]};
            auto lapT = std::chrono::steady_clock::now();
            auto lap = [&](const char *what, uint64_t units) {
                if (!verbose || !b->counting) return;
                unsigned long long call[S3_COUNTER_WORDS] = {}, c[S3_NUM_COUNTERS] = {};
                (void)d2h(call, b->dCounters, S3_COUNTER_WORDS * 8, b->stream);
                for (int r = 0; r < S3_NUM_ROLES; r++) for (int k = 0; k < S3_NUM_COUNTERS; k++) c[k] += call[r * S3_NUM_COUNTERS + k];
                const auto now = std::chrono::steady_clock::now();
                fprintf(stderr, "[s3dp] search mm=%u %-12s %10llu units %8.2f ms %12llu rank queries (second pass so far: cut %llu + %llu units, %llu lane trips, longest unit %llu trips; "
                                "unit kernel: %llu wave trips, %llu frames adopted; %llu suffix-array reads, %llu compare windows, %llu candidates accepted by ceKernel)\n",
                        modelMm, what, (unsigned long long)units, std::chrono::duration<double, std::milli>(now - lapT).count(), c[0] - rankSeen, c[4], c[5], c[6] - tripsSeen, c[7],
                        c[9] - wavesSeen, c[8] - adoptSeen, c[2] - saSeen, c[3] - ceSeen, c[10] - accSeen);
                if (c[9] - wavesSeen)
                    fprintf(stderr, "[s3dp]   unit kernel paths: %llu wave trips with a lane in check-and-extend (%llu lane trips), %llu stepping lane trips, %llu lane pops, %llu wave trips that book\n",
                            c[11] - dSeen[0], c[12] - dSeen[1], c[13] - dSeen[2], c[15] - dSeen[4], c[14] - dSeen[3]);
                if (c[20] - wavesCountSeen)
                    fprintf(stderr, "[s3dp]   %llu waves, %.2f ms on average from their first instruction to their last (of the launch's time above: the rest is its tail and its launch)\n",
                            c[20] - wavesCountSeen, (double)(c[19] - waveTicksSeen) * 1e-5 / (double)(c[20] - wavesCountSeen));
                waveTicksSeen = c[19]; wavesCountSeen = c[20];
                if (c[16] - dSeen[5])
                    fprintf(stderr, "[s3dp]   unit kernel fast trips: %llu of the wave trips (%llu stepping lane trips in them); %llu lane trips parked at a full trip\n",
                            c[16] - dSeen[5], c[17] - dSeen[6], c[18] - dSeen[7]);
                dSeen[0] = c[11]; dSeen[1] = c[12]; dSeen[2] = c[13]; dSeen[3] = c[14]; dSeen[4] = c[15]; dSeen[5] = c[16]; dSeen[6] = c[17]; dSeen[7] = c[18];
                rankSeen = c[0]; tripsSeen = c[6]; wavesSeen = c[9]; adoptSeen = c[8]; saSeen = c[2]; ceSeen = c[3]; accSeen = c[10]; lapT = now;
                for (int r = 0; r < S3_NUM_ROLES; r++) (void)hipMemsetAsync(b->dCounters + r * S3_NUM_COUNTERS + 7, 0, 8, b->stream);
            };
            if (verbose && b->counting) { (void)hipStreamSynchronize(b->stream); lapT = std::chrono::steady_clock::now(); }
            // the depth-first stacks of the first-pass and unit-pass kernels: a ring of frames per lane of the widest grid
            const unsigned blocksFirst = std::min<unsigned>(blocksPerCu, S3_FIRST_WAVES_PER_SIMD), blocksUnit = std::min<unsigned>(blocksPerCu, S3_UNIT_WAVES_PER_SIMD);
            {
                const size_t waves = unitPassMaxWaves(256u * std::max(blocksFirst, blocksUnit));
                if (waves > b->unitStackWaves) {
                    int rc = bgrow(b, &b->dUnitStack, waves * S3_UNIT_STACK * 64u, 0);
                    if (rc) { b->unitStackWaves = 0; return rc; }
                    b->unitStackWaves = waves;
                }
                out.unitStack = b->dUnitStack;
            }
            hipError_t le = (hipError_t)launchFirstPass(b->ix->d, in, out, 256u * blocksFirst, b->stream);
            if (le != hipSuccess) return fail(S3_EHIP, std::string("search kernel: cannot raise the dynamic LDS limit: ") + hipGetErrorString(le));
            lap("first pass", nActive);
            HIPCHK(mark(b->evSearch, b->evSearchUsed, false, 0));
            // check-and-extend takes the task records written since its last launch: dTaskRange = {first, end}; what it accepts is appended behind them
            auto checkExtend = [&]() -> int {
                HIPCHK(hipMemcpyAsync(b->dTaskRange + 1, b->dRecCount, 4, hipMemcpyDeviceToDevice, b->stream));
                HIPCHK(mark(b->evCe, b->evCeUsed, true, 3));
                out.counters = countersOf(3);
                // (S3DP_FLANK is read here too: 0 / 1 make ceKernel ignore tables the index holds -- A/B runs and the tests' matrix on one resident index)
                {
                    std::shared_lock<std::shared_mutex> r(b->ix->flankMu);   // (the tables cannot be released between this copy and the launch; s3_index::flankMu)
                    DevIndex dce = b->ix->d;
                    const size_t useFlank = envSize("S3DP_FLANK", 2);
                    if (useFlank < 2) dce.flankFwd = nullptr;
                    if (useFlank < 1) dce.flankBack = nullptr;
                    launchCheckExtend(dce, in, out, b->dTaskRange, b->stream);
                }
                HIPCHK(mark(b->evCe, b->evCeUsed, false, 3));
                HIPCHK(hipMemcpyAsync(b->dTaskRange, b->dRecCount, 4, hipMemcpyDeviceToDevice, b->stream));
                return S3_OK;
            };
            HIPCHK(hipMemsetAsync(b->dTaskRange, 0, 8, b->stream));
            if (in.heavyLimit != 0xFFFFFFFFu) {
                // second pass: the listed reads taken apart into units.  Their number sizes the count / offset arrays: one host round trip (the first
                // pass is short: it drops a heavy read after a few hundred candidates)
                HIPCHK(hipMemcpyAsync(&nHeavy, b->dHeavyCount, 4, hipMemcpyDeviceToHost, b->stream));
                HIPCHK(hipStreamSynchronize(b->stream));
                if (nHeavy) {
                    uint32_t U = unitsPerItem;
                    const uint64_t groups = (uint64_t)nCases * 2u;
                    if (nItems + (uint64_t)nHeavy * groups * U >= 0x7FFFFF00ull) U = 1;          // (unit indices share a word with a flag bit)
                    if (nItems + (uint64_t)nHeavy * groups * U >= 0x7FFFFF00ull) return fail(S3_EOVERFLOW, "too many heavy reads in one batch: use smaller batches");
                    // the unit pass (s3dp_unit.hip) lets idle lanes adopt frames of busy ones once the queue is empty: every adopted frame counts its
                    // records in a counter of its own behind the units' (S3DP_UNIT_NODES of them, 0: none)
                    const bool newUnitPass = U > 1;
                    size_t dynCap = newUnitPass ? envSize("S3DP_UNIT_NODES", 1u << 24) : 0;
                    const size_t nStatic = nItems + (size_t)nHeavy * groups * U + 1;
                    if (nStatic + dynCap >= 0x7FFFFF00ull) dynCap = 0;
                    const size_t need = nStatic + dynCap;
                    if (need > b->countCap) {
                        size_t cap = b->countCap;
                        while (cap < need) cap += cap / 2 + 1;
                        int rc;
                        if ((rc = bgrow(b, &b->dItemCounts, cap, nItems)) || (rc = bgrow(b, &b->dItemOffsets, cap, 0))) return rc;
                        b->countCap = cap;
                        size_t t1 = 0, t3 = 0;
                        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t1, b->dItemCounts, b->dItemOffsets, (int)cap, b->stream);
                        hipcub::TransformInputIterator<unsigned long long, U32ToU64, const uint32_t *> it(b->dItemCounts, U32ToU64());
                        (void)hipcub::DeviceReduce::Sum(nullptr, t3, it, b->dSum64, (int)cap, b->stream);
                        if (t3 > t1) t1 = t3;
                        if (t1 + 256 > b->tempBytes) { if ((rc = bgrow(b, &b->dTemp, t1 + 256, 0))) return rc; b->tempBytes = t1 + 256; }
                        out.itemCounts = b->dItemCounts;
                    }
                    nCount = (uint32_t)(nStatic - 1);
                    // family-mates next to each other in the list (s3dp_unit.hip, "LOCALITY"; S3DP_SORT_LISTED=0: the order the first pass listed them in)
                    if (nHeavy > 1 && envSize("S3DP_SORT_LISTED", 1) != 0) {
                        int rc = orderByReadPrefix(b, b->dHeavyList, nHeavy, active, 2u * in.wpqLoad, in.wLds);
                        if (rc) return rc;
                        launchListedPlace(b->dOrderVals[1], nHeavy, b->dHeavyList, b->dHeavyIndex, b->stream);
                    }
                    HIPCHK(hipMemsetAsync(b->dItemCounts + nItems, 0, (nStatic - nItems) * 4, b->stream));
                    // the listed reads count per unit from here on
                    launchZeroListedCounts(b->dHeavyList, b->dHeavyCount, nHeavy, (uint32_t)groups, b->dItemCounts, b->stream);
                    // the trunk states: 16 B per (listed read, item, position)
                    const size_t nStates = (size_t)nHeavy * groups * U;
                    if (nStates > b->trunkCap) {
                        int rc = bgrow(b, &b->dTrunk, nStates + nStates / 4, 0);
                        if (rc) return rc;
                        b->trunkCap = nStates + nStates / 4;
                    }
                    HIPCHK(hipMemsetAsync(b->dTrunk, 0, nStates * sizeof(uint4), b->stream));
                    out.trunkStates = b->dTrunk;
                    out.readCut = b->dLongList;                                // (free until the merge kernels list their long reads)
                    HIPCHK(hipMemsetAsync(b->dLongList, 0xFF, (size_t)nHeavy * 4, b->stream));
                    if ((size_t)nHeavy * groups > b->accCap) {
                        int rc = bgrow(b, &b->dItemAcc, (size_t)nHeavy * groups * 5 / 4, 0);
                        if (rc) return rc;
                        b->accCap = (size_t)nHeavy * groups * 5 / 4;
                    }
                    HIPCHK(hipMemsetAsync(b->dItemAcc, 0, (size_t)nHeavy * groups * 4, b->stream));
                    out.itemAcc = b->dItemAcc;
                    SearchIn in2 = in;
                    in2.slotList = b->dHeavyList; in2.slotCount = b->dHeavyCount; in2.heavyLimit = 0xFFFFFFFFu; in2.heavyRank = 0xFFFFFFFFu;
                    in2.unitsPerItem = U; in2.unitBase = (uint32_t)nItems;
                    in.unitsPerItem = U; in.unitBase = (uint32_t)nItems;       // (the scatter decodes unit indices with them)
                    {
                        // the listed reads' numbers in readsT (trunk pass and unit pass)
                        int rc;
                        if (nHeavy > b->listedCap) {
                            if ((rc = bgrow(b, &b->dListedRead, (size_t)nHeavy + nHeavy / 4 + 64, 0))) { b->listedCap = 0; return rc; }
                            b->listedCap = (size_t)nHeavy + nHeavy / 4 + 64;
                        }
                        out.listedRead = b->dListedRead;
                        launchListedReads(b->dHeavyList, b->dHeavyCount, active, b->dListedRead, nHeavy, b->stream);
                    }
                    if (newUnitPass) {
                        // what the unit-pass kernel needs besides: the chains of adopted frames
                        int rc;
                        const size_t nChain = (size_t)nHeavy * groups * U + 1 + dynCap;
                        if (nChain > b->chainCap) {
                            if ((rc = bgrow(b, &b->dChainNext, nChain + nChain / 4, 0))) { b->chainCap = 0; return rc; }
                            b->chainCap = nChain + nChain / 4;
                        }
                        if (dynCap > b->dynCap) {
                            b->dynCap = 0;
                            if ((rc = bgrow(b, &b->dDynOwner, dynCap, 0)) || (rc = bgrow(b, &b->dDynLocal, dynCap, 0))) return rc;
                            b->dynCap = dynCap;
                        }
                        if (!b->dDynUsed && (rc = balloc(b, &b->dDynUsed, 4))) return rc;
                        HIPCHK(hipMemsetAsync(b->dChainNext, 0, ((size_t)nHeavy * groups * U + 1) * 4, b->stream));
                        HIPCHK(hipMemsetAsync(b->dDynUsed, 0, 4, b->stream));
                        out.chainNext = b->dChainNext;
                        out.dynOwner = b->dDynOwner; out.dynLocal = b->dDynLocal; out.dynUsed = b->dDynUsed;
                        out.dynBase = (uint32_t)nStatic; out.dynCap = (uint32_t)dynCap;
                    }
                    for (uint32_t mode = 1; mode <= (U > 1 ? 2u : 1u); mode++) {   // trunk pass, then unit pass
                        in2.unitMode = mode;
                        HIPCHK(hipMemsetAsync(b->dWorkHead, 0, 8, b->stream));
                        out.counters = countersOf((int)mode);
                        HIPCHK(mark(b->evSearch, b->evSearchUsed, true, (uint8_t)mode));
                        if (mode == 2u) le = (hipError_t)launchUnitPass(b->ix->d, in2, out, 256u * blocksUnit, b->stream);
                        else le = (hipError_t)launchTrunkPass(b->ix->d, in2, out, 256u * blocksFirst, b->stream);
                        if (le != hipSuccess) return fail(S3_EHIP, std::string("search kernel (second pass): ") + hipGetErrorString(le));
                        HIPCHK(mark(b->evSearch, b->evSearchUsed, false, (uint8_t)mode));
                        lap(mode == 1 ? "trunk pass" : "unit pass", (uint64_t)nHeavy * groups * (mode == 1 ? 1 : U));
                    }
                    // the counters of adopted frames become part of their units' totals before anything is scanned (check-and-extend fills reserved
                    // sequence numbers: it changes no count)
                    if (newUnitPass && dynCap) launchUnitChains(out, (uint32_t)nItems, (uint32_t)((size_t)nHeavy * groups * U), b->stream);
                    b->scatterDyn = ScatterDyn{b->dDynOwner, b->dDynLocal, out.dynBase, out.dynCap};
                }
            }
            {
                int rc = checkExtend();
                if (rc) return rc;
                lap("check-extend", 0);
            }
            if (verbose && b->counting)
                fprintf(stderr, "[s3dp] search mm=%u listed reads %u of %u, units per item %u\n", modelMm, nHeavy, nActive, in.unitsPerItem);
        }
        // offsets of the items in the ordered array (the scan runs before the host looks at the flags: one synchronisation per pass for both answers)
        HIPCHK(hipMemsetAsync(b->dItemCounts + nCount, 0, 4, b->stream));
        {
            size_t tb = b->tempBytes;
            HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->dTemp, tb, b->dItemCounts, b->dItemOffsets, (int)(nCount + 1), b->stream));
            int rc = sum64Async(b, b->dItemCounts, nCount);
            if (rc) return rc;
        }

        // a record that did not fit was dropped but counted: the ordering below must not run on such a pass.  Grow and search again.
        uint32_t flags = 0;
        HIPCHK(hipMemcpyAsync(&flags, b->dFlags, 4, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipMemcpyAsync(&slots, b->dItemOffsets + nCount, 4, hipMemcpyDeviceToHost, b->stream));
        unsigned long long slots64 = 0;
        HIPCHK(hipMemcpyAsync(&slots64, b->dSum64, 8, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        if (slots64 != slots)
            return fail(S3_EOVERFLOW, "more than 2^32 hit records and check-and-extend candidates in one search pass (" + std::to_string(slots64) +
                                          "): use smaller batches");
        if (!(flags & S3_FLAG_REC_OVERFLOW)) break;
        int rc = growRecords(b);
        if (rc) return rc;
        flags &= ~S3_FLAG_REC_OVERFLOW;
        HIPCHK(hipMemcpyAsync(b->dFlags, &flags, 4, hipMemcpyHostToDevice, b->stream));
        if (b->counting) HIPCHK(hipMemcpyAsync(b->dCounters, counterSnapAll, S3_COUNTER_WORDS * 8, hipMemcpyHostToDevice, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    b->searchLaunches++;
    {
        // sequence numbers include the rejected check-and-extend candidates: the ordered array can need more slots than there are records
        if (slots > b->ordCap) {
            size_t cap = (size_t)slots + slots / 8 + 65536;               // (not doubled: with its validity bytes, 5 B per sequence number of the pass -- 16 B copies of the records until round 6)
            if (cap > kMaxEntries) cap = kMaxEntries;
            if (slots > cap) return fail(S3_EOVERFLOW, "ordering buffer overflow at the 32-bit limit");
            int rc = bgrow(b, &b->dOrdIdx, cap, 0);
            if (rc) return rc;
            if ((rc = bgrow(b, &b->dOrdValid, cap + 8, 0))) return rc;
            b->ordCap = (uint32_t)cap;
        }
    }
    // nine in ten sequence numbers belong to rejected check-and-extend candidates: the holes are a zero BYTE each (they used to be 16-byte entries written
    // by a fill pass and read back by merge and decode: 31 GB per ladder of the benchmark, three times)
    if (slots) HIPCHK(hipMemsetAsync(b->dOrdValid, 0, slots, b->stream));
    launchScatter(b->dRecs, b->dRecCount, b->recCap, b->dItemOffsets, b->dOrdIdx, b->dOrdValid, b->dHeavyFlags, b->dHeavyList, (uint32_t)nCases * 2u, in.unitBase, in.unitsPerItem,
                  b->scatterDyn, b->stream);
    // what the merge / decode kernels need to find a read's slice of the ordered array (kept for the decode calls that follow this pass)
    b->mergeUnitBase = in.unitBase; b->mergeUnitsPerRead = nHeavy ? (uint32_t)nCases * 2u * in.unitsPerItem : 0u;
    // the long-read list of the merge kernels shares the buffer of the heavy list, which the scatter above still reads: a second list
    
    MergeArgs m{};
    m.nActive = nActive; m.nCases = (uint32_t)nCases; m.maxOut = maxOut; m.itemOffsets = b->dItemOffsets; m.ordIdx = b->dOrdIdx; m.recs = b->dRecs; m.valid = b->dOrdValid;
    m.keptRecs = b->dKept; m.occCounts = b->dOccCounts; m.ceCounts = b->dCeCounts; m.totalOcc = b->dTotalOcc; m.withError = b->dWithError;
    m.longList = b->dLongList; m.longCount = b->dLongCount;
    m.heavyFlags = b->mergeUnitsPerRead ? b->dHeavyFlags : nullptr; m.heavyIndex = b->dHeavyIndex; m.unitBase = b->mergeUnitBase; m.unitsPerRead = b->mergeUnitsPerRead;
    launchMergeCount(m, b->stream);
    b->lastNumCases = (uint32_t)nCases;
    if (b->counting) b->recordsTotal += slots;
    HIPCHK(hipGetLastError());
    return S3_OK;
}

int s3i::checkFlags(s3_batch *b)
{
    uint32_t flags = 0;
    HIPCHK(hipMemcpyAsync(&flags, b->dFlags, 4, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (flags & S3_FLAG_STACK_OVERFLOW) return fail(S3_EOVERFLOW, "search stack overflow (internal bound exceeded)");
    if (flags & S3_FLAG_REC_OVERFLOW) return fail(S3_EOVERFLOW, "record buffer overflow: raise S3DP_REC_PER_READ");
    if (flags & S3_FLAG_OCC_OVERFLOW) return fail(S3_EOVERFLOW, "occurrence buffer overflow: raise S3DP_OCC_PER_READ");
    if (flags & S3_FLAG_FLANK) return fail(S3_EHIP, "check-and-extend: the flank table of the reverse index disagrees with the index (run with S3DP_FLANK=1 and report this)");
    return S3_OK;
}

extern "C" int s3_batch_search(s3_batch *b, uint32_t numMismatch, uint32_t maxOutputPerRead)
{
    if (!b) return fail(S3_EINVAL, "null batch");
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    if (numMismatch > S3_MAX_MISMATCH) return fail(S3_EINVAL, "numMismatch must be 0..4");
    if (maxOutputPerRead == 0 || maxOutputPerRead > 0xFFFFFFu) return fail(S3_EINVAL, "maxOutputPerRead must be 1..16777215");
    HIPCHK(hipSetDevice(b->ix->dev));
    b->searchLaunches = 0; b->evSearchUsed = 0; b->evCeUsed = 0;
    b->recordsTotal = 0;
    HIPCHK(hipMemsetAsync(b->dFlags, 0, 4, b->stream));
    if (b->counting) HIPCHK(hipMemsetAsync(b->dCounters, 0, S3_COUNTER_WORDS * 8, b->stream));
    HIPCHK(hipEventRecord(b->evTotal[0], b->stream));
    b->unitGroup = S3_UNIT_GROUP;
    HIPCHK(hipMemsetAsync(b->dWasHeavy, 0, b->maxReads, b->stream));
    int rc = searchPass(b, nullptr, b->n, numMismatch, maxOutputPerRead);
    if (rc) return rc;
    // hits in the ABI format
    HIPCHK(hipMemsetAsync(b->dKept + b->n, 0, 4, b->stream));
    size_t tb = b->tempBytes;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->dTemp, tb, b->dKept, b->dHitOffsets, (int)(b->n + 1), b->stream));
    MergeArgs m{};
    m.nActive = b->n; m.nCases = b->lastNumCases; m.maxOut = maxOutputPerRead; m.itemOffsets = b->dItemOffsets; m.ordIdx = b->dOrdIdx; m.recs = b->dRecs; m.valid = b->dOrdValid;
    m.keptRecs = b->dKept; m.occCounts = b->dOccCounts; m.ceCounts = b->dCeCounts; m.totalOcc = b->dTotalOcc; m.withError = b->dWithError;
    m.heavyFlags = b->mergeUnitsPerRead ? b->dHeavyFlags : nullptr; m.heavyIndex = b->dHeavyIndex; m.unitBase = b->mergeUnitBase; m.unitsPerRead = b->mergeUnitsPerRead;
    if (!b->dHits) { int rc2 = balloc(b, &b->dHits, b->recCap); if (rc2) return rc2; }
    launchMergeFill(m, b->dHitOffsets, b->dHits, b->stream);
    HIPCHK(hipEventRecord(b->evTotal[1], b->stream));
    b->lastKind = 1;
    return S3_OK;
}

extern "C" int s3_batch_sync(s3_batch *b)
{
    if (!b) return fail(S3_EINVAL, "null batch");
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    HIPCHK(hipSetDevice(b->ix->dev));
    HIPCHK(hipStreamSynchronize(b->stream));
    return checkFlags(b);
}

extern "C" int s3_batch_fetch_hits(s3_batch *b, uint64_t *hitOffsets, s3_hit *hits, size_t hitCap, uint32_t *totalOcc, uint32_t *withError,
                                   uint64_t *nHitsOut)
{
    if (!b || b->lastKind != 1) return fail(S3_EINVAL, "s3_batch_fetch_hits: no search result pending");
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    int rc = s3_batch_sync(b);
    if (rc) return rc;
    const uint32_t n = b->n;
    std::vector<uint32_t> off((size_t)n + 1);
    HIPCHK(d2h(off.data(), b->dHitOffsets, ((size_t)n + 1) * 4, b->stream));
    const uint64_t total = off[n];
    if (nHitsOut) *nHitsOut = total;
    if (hitOffsets) for (size_t i = 0; i <= n; i++) hitOffsets[i] = off[i];
    if (hits) {
        if (total > hitCap) return fail(S3_EOVERFLOW, "s3_batch_fetch_hits: hit buffer too small");
        static_assert(sizeof(s3_hit) == sizeof(DevHit), "hit layout");
        HIPCHK(d2h(hits, b->dHits, total * sizeof(DevHit), b->stream));
    }
    if (totalOcc) HIPCHK(d2h(totalOcc, b->dTotalOcc, (size_t)n * 4, b->stream));
    if (withError) HIPCHK(d2h(withError, b->dWithError, (size_t)n * 32, b->stream));
    return S3_OK;
}

extern "C" int s3_batch_last_timing(s3_batch *b, float *totalMs, float *searchMs, uint32_t *searchLaunches)
{
    if (!b || b->lastKind == 0) return fail(S3_EINVAL, "no timed call");
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    HIPCHK(hipSetDevice(b->ix->dev));
    HIPCHK(hipStreamSynchronize(b->stream));
    float t = 0, s = 0;
    HIPCHK(hipEventElapsedTime(&t, b->evTotal[0], b->evTotal[1]));
    uint32_t timed = b->evSearchUsed;
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

extern "C" int s3_batch_last_ce_timing(s3_batch *b, float *ceMs, uint32_t *ceLaunches)
{
    if (!b || b->lastKind == 0) return fail(S3_EINVAL, "no timed call");
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    HIPCHK(hipSetDevice(b->ix->dev));
    HIPCHK(hipStreamSynchronize(b->stream));
    float s = 0;
    uint32_t timed = b->evCeUsed;
    if (timed * 2 > b->evCe.size()) timed = (uint32_t)(b->evCe.size() / 2);
    for (uint32_t i = 0; i < timed; i++) {
        float x = 0;
        HIPCHK(hipEventElapsedTime(&x, b->evCe[2 * i], b->evCe[2 * i + 1]));
        s += x;
    }
    if (ceMs) *ceMs = s;
    if (ceLaunches) *ceLaunches = timed;
    return S3_OK;
}

// per kernel of the search passes of the last call: its device time (HIP events around its launches) and, with counters enabled, its own tallies
extern "C" int s3_batch_kernel_stats(s3_batch *b, s3_kernel_stats *out)
{
    if (!b || !out || b->lastKind == 0) return fail(S3_EINVAL, "s3_batch_kernel_stats: no timed call");
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    HIPCHK(hipSetDevice(b->ix->dev));
    HIPCHK(hipStreamSynchronize(b->stream));
    for (int r = 0; r < S3_NUM_ROLES; r++) out[r] = s3_kernel_stats{};
    uint32_t timed = b->evSearchUsed;
    if (timed * 2 > b->evSearch.size()) timed = (uint32_t)(b->evSearch.size() / 2);
    for (uint32_t i = 0; i < timed; i++) {
        float x = 0;
        HIPCHK(hipEventElapsedTime(&x, b->evSearch[2 * i], b->evSearch[2 * i + 1]));
        const int r = b->evSearchRole[i] < 3 ? b->evSearchRole[i] : 0;
        out[r].ms += x; out[r].launches++;
    }
    timed = b->evCeUsed;
    if (timed * 2 > b->evCe.size()) timed = (uint32_t)(b->evCe.size() / 2);
    for (uint32_t i = 0; i < timed; i++) {
        float x = 0;
        HIPCHK(hipEventElapsedTime(&x, b->evCe[2 * i], b->evCe[2 * i + 1]));
        out[3].ms += x; out[3].launches++;
    }
    unsigned long long call[S3_COUNTER_WORDS] = {0};
    HIPCHK(d2h(call, b->dCounters, S3_COUNTER_WORDS * 8, b->stream));
    for (int r = 0; r < S3_NUM_ROLES; r++) {
        const unsigned long long *c = call + r * S3_NUM_COUNTERS;
        out[r].rankQueries = c[0]; out[r].ltLookups = c[1]; out[r].saReads = c[2]; out[r].ceWindows = c[3];
        out[r].listedRankQueries = c[21]; out[r].accepted = c[10];
        for (int k = 0; k < S3_CE_HIST_SLOTS; k++) out[r].ceRejected[k] = c[S3_CE_HIST_BASE + k];
    }
    return S3_OK;
}

extern "C" int s3_batch_counters(s3_batch *b, s3_counters *out)
{
    if (!b || !out) return fail(S3_EINVAL, "null argument");
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    HIPCHK(hipSetDevice(b->ix->dev));
    unsigned long long call[S3_COUNTER_WORDS] = {0}, c[8] = {0};
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(d2h(call, b->dCounters, S3_COUNTER_WORDS * 8, b->stream));
    for (int r = 0; r < S3_NUM_ROLES; r++) for (int k = 0; k < 8; k++) c[k] += call[r * S3_NUM_COUNTERS + k];
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

// measurement aid (bench.py's roofline denominator): streaming copy bandwidth of the device with an own 16-byte copy kernel
extern "C" int s3_probe_copy(int device, size_t bytes, int repeats, float *msOut)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(S3_ENODEV, "no HIP device available");
    HIPCHK(hipSetDevice(device));
    const size_t n16 = bytes / 16;
    if (n16 == 0 || repeats < 1) return fail(S3_EINVAL, "bad probe arguments");
    uint4 *src = nullptr, *dst = nullptr;
    HIPCHK(hipMalloc((void **)&src, n16 * 16));
    if (hipMalloc((void **)&dst, n16 * 16) != hipSuccess) { (void)hipFree(src); return fail(S3_ENOMEM, "s3_probe_copy: hipMalloc"); }
    HIPCHK(hipMemset(src, 1, n16 * 16));
    HIPCHK(hipMemset(dst, 2, n16 * 16));
    hipStream_t s;
    hipEvent_t e0, e1;
    HIPCHK(hipStreamCreate(&s));
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    // the fastest of the copy shapes of launchCopyProbe (S3DP_PROBE_COPY_VERBOSE=1 lists them all)
    const char *pv = getenv("S3DP_PROBE_COPY_VERBOSE");
    float best = 0;
    for (int v = 0; v < copyProbeVariants(); v++) {
        launchCopyProbe(src, dst, n16, v, s);                                                      // warm-up
        HIPCHK(hipEventRecord(e0, s));
        for (int r = 0; r < repeats; r++) launchCopyProbe(src, dst, n16, v, s);
        HIPCHK(hipEventRecord(e1, s));
        HIPCHK(hipStreamSynchronize(s));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        ms /= repeats;
        if (pv && atoi(pv) > 0) fprintf(stderr, "[s3dp] copy probe shape %d: %.3f ms, %.0f GB/s\n", v, ms, 2.0 * n16 * 16 / ms / 1e6);
        if (best == 0 || ms < best) best = ms;
    }
    if (msOut) *msOut = best;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    (void)hipFree(src); (void)hipFree(dst);
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

// this file's code object holds the rocPRIM / hipCUB kernels of the search pass (sums, scans, the sort of the listed reads): an empty kernel to ask for
// (preloadKernels, s3dp_kernels.hip)
__global__ void apiPreloadKernel() {}
void s3::preloadApi() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)apiPreloadKernel); (void)hipGetLastError(); }
