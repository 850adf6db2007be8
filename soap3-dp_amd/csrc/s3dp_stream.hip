// s3dp_stream.hip -- streamed single-end alignment behind the C-ABI (include/s3dp.h: s3_stream_*).
//
// The reference overlaps the device work of batch i+1 with the CPU threads of batch i (alignment.cu:555-561, 1005-1031) and loops over batches
// of reads (SOAP3-DP.cu:607-800).  Here a stream object owns `depth` slots; a slot = one s3_batch (its own non-blocking HIP stream and work
// buffers) + pinned host staging for the packed reads going in and the occurrence lists coming out + one worker thread that runs
//     upload (H2D from pinned memory) -> ladder (s3_batch_align_single) -> download (D2H into pinned memory)
// for the batch it was handed.  With three slots the upload of batch i+1, the kernels of batch i and the download of batch i-1 are in flight at
// the same time on three HIP streams, and the caller (who parses reads into the next slot's input buffers and formats the previous slot's
// results) never waits for a copy.  Results come back in submission order.
#include <condition_variable>
#include <mutex>
#include <thread>
#include <unordered_map>
#include "s3dp_internal.h"

using namespace s3i;

namespace {
struct Slot {
    s3_batch *b = nullptr;
    // pinned input
    uint32_t *hQueries = nullptr, *hLens = nullptr, *hSeedLens = nullptr;
    // pinned output
    uint8_t *hPhase = nullptr;
    uint32_t *hWithError = nullptr, *hOffsets = nullptr, *hPos = nullptr, *hMeta = nullptr;
    size_t occCap = 0;
    // job
    uint32_t n = 0, mm = 0, maxOut = 0;
    int reportType = 2;
    bool useSeeds = false;
    uint64_t ticket = 0;
    float deviceMs = 0;
    int rc = S3_OK;
    std::string err;
    enum { FREE, ACQUIRED, QUEUED, DONE, UNBORN, FAILED } state = UNBORN;     // UNBORN: its buffers are still being made (s3_stream_create's helper thread)
    std::thread worker;
};
}  // namespace

struct s3_stream {
    s3_index *ix = nullptr;
    uint32_t maxReads = 0, wpq = 0;
    std::vector<Slot> slots;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t nextTicket = 0, nextResult = 0;      // tickets are handed out in submission order; results are returned in that order
    int acquired = -1;                            // slot handed to the caller by s3_stream_acquire and not yet submitted
    // batches whose device work is under way, and how many may be (S3DP_STREAM_DEVICE_JOBS, default: every slot): a caller who formats a batch's results
    // while it holds the slot can ask for more slots than ladders on the device at once; batches start in submission order
    int running = 0, deviceJobs = 0;
    uint64_t nextStart = 0;
    bool quit = false;
    std::thread maker;                            // makes the slots behind the first one while the first batches are under way
    std::string makerErr;
};

// Pinned host memory comes from a cache of the process: hipHostMalloc pins page by page (0.2-0.4 s per GB) and NO kernel of any stream starts while it does
// (three slots growing their staging to 256 MB in the first batches of a run: 0.2 s of a 1.2 s run with the device idle, profiles/README_r06.md section 8);
// hipHostFree is no better.  s3_host_staging_reserve fills the cache ahead of time (the command line: while the index files are read), blocks a stream gives
// back stay in it (up to 4 GB) for the next stream of the process, s3_host_staging_release empties it.
namespace {
struct PinnedBlock { void *p; size_t bytes; int kind; };                 // kind 1: a slot's occurrence lists (taken whole, whatever their size); 0: its other arrays
struct PinnedCache {
    std::mutex mu;
    std::vector<PinnedBlock> idle;
    std::unordered_map<void *, PinnedBlock> known;
    size_t idleBytes = 0;
} g_pinned;

// a block of at least `atLeast` bytes: the smallest idle one of its kind in the cache (kind 0: not above twice the size), or a fresh one of `fresh` bytes
hipError_t pinnedAlloc(int kind, void **p, size_t atLeast, size_t fresh, size_t *got)
{
    {
        std::lock_guard<std::mutex> g(g_pinned.mu);
        int best = -1;
        for (int k = 0; k < (int)g_pinned.idle.size(); k++) {
            const PinnedBlock &c = g_pinned.idle[(size_t)k];
            if (c.kind == kind && c.bytes >= atLeast && (kind == 1 || c.bytes <= 2 * fresh) && (best < 0 || c.bytes < g_pinned.idle[(size_t)best].bytes)) best = k;
        }
        if (best >= 0) {
            *p = g_pinned.idle[(size_t)best].p;
            if (got) *got = g_pinned.idle[(size_t)best].bytes;
            g_pinned.idleBytes -= g_pinned.idle[(size_t)best].bytes;
            g_pinned.idle.erase(g_pinned.idle.begin() + best);
            return hipSuccess;
        }
    }
    const hipError_t e = hipHostMalloc(p, fresh, hipHostMallocDefault);
    if (e == hipSuccess) { std::lock_guard<std::mutex> g(g_pinned.mu); g_pinned.known[*p] = PinnedBlock{*p, fresh, kind}; }
    if (got) *got = fresh;
    return e;
}

void pinnedFree(void *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_pinned.mu);
        auto it = g_pinned.known.find(p);
        if (it != g_pinned.known.end() && g_pinned.idleBytes + it->second.bytes <= ((size_t)4 << 30)) {
            g_pinned.idle.push_back(it->second);
            g_pinned.idleBytes += it->second.bytes;
            return;
        }
        if (it != g_pinned.known.end()) g_pinned.known.erase(it);
    }
    (void)hipHostFree(p);
}
}  // namespace

// the pinned arrays of one slot apart from the occurrence lists (makeSlot below)
static void slotStagingSizes(uint32_t maxReads, uint32_t wordPerQuery, size_t out[6])
{
    const size_t roundUp = ((size_t)maxReads + 31) / 32 * 32;
    out[0] = roundUp * wordPerQuery * 4; out[1] = roundUp * 4; out[2] = roundUp * 4; out[3] = roundUp; out[4] = roundUp * 32; out[5] = (roundUp + 1) * 4;
}

extern "C" int s3_host_staging_reserve(uint32_t maxReads, uint32_t wordPerQuery, int depth, uint32_t occPerRead)
{
    if (maxReads == 0 || wordPerQuery == 0 || depth < 1 || depth > 8) return fail(S3_EINVAL, "s3_host_staging_reserve: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(S3_ENODEV, "no HIP device available");
    size_t sizes[8];
    slotStagingSizes(maxReads, wordPerQuery, sizes);
    sizes[6] = sizes[7] = (size_t)maxReads * occPerRead * 4;
    for (int k = 0; k < depth; k++)
        for (int k2 = 0; k2 < 8; k2++) {
            const size_t bytes = sizes[k2];
            if (!bytes) continue;
            void *p = nullptr;
            const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
            if (e != hipSuccess) { (void)hipGetLastError(); return fail(S3_ENOMEM, std::string("s3_host_staging_reserve: hipHostMalloc: ") + hipGetErrorString(e)); }
            std::lock_guard<std::mutex> g(g_pinned.mu);
            const PinnedBlock blk{p, bytes, k2 >= 6 ? 1 : 0};
            g_pinned.known[p] = blk;
            g_pinned.idle.push_back(blk);
            g_pinned.idleBytes += bytes;
        }
    return S3_OK;
}

extern "C" void s3_host_staging_release(void)
{
    std::vector<void *> mine;
    {
        std::lock_guard<std::mutex> g(g_pinned.mu);
        for (auto &e : g_pinned.idle) { mine.push_back(e.p); g_pinned.known.erase(e.p); }
        g_pinned.idle.clear();
        g_pinned.idleBytes = 0;
    }
    for (void *p : mine) (void)hipHostFree(p);
}

static int growOut(Slot &s, size_t need)
{
    if (need <= s.occCap) return S3_OK;
    size_t cap = s.occCap ? s.occCap : 1 << 20;
    while (cap < need) cap *= 2;
    const auto t0 = std::chrono::steady_clock::now();
    pinnedFree(s.hPos);
    pinnedFree(s.hMeta);
    s.hPos = s.hMeta = nullptr; s.occCap = 0;
    size_t gotPos = 0, gotMeta = 0;
    HIPCHK(pinnedAlloc(1, (void **)&s.hPos, need * 4, cap * 4, &gotPos));
    HIPCHK(pinnedAlloc(1, (void **)&s.hMeta, need * 4, cap * 4, &gotMeta));
    cap = std::min(gotPos, gotMeta) / 4;
    noteAlloc("pinned staging of the occurrence lists", cap * 8, t0);
    s.occCap = cap;
    return S3_OK;
}

// the device work of a slot's batch: upload, ladder (what S3DP_STREAM_DEVICE_JOBS counts) ...
static int runLadder(s3_stream *st, Slot &s)
{
    s3_batch *b = s.b;
    HIPCHK(hipSetDevice(st->ix->dev));
    int rc = s3_batch_upload(b, s.hQueries, s.hLens, s.n);
    if (rc) return rc;
    if (s.useSeeds && (rc = s3_batch_set_seed_lengths(b, s.hSeedLens))) return rc;
    if ((rc = s3_batch_align_single(b, s.mm, s.maxOut, s.reportType))) return rc;
    if ((rc = s3_batch_sync(b))) return rc;
    float tot = 0, srch = 0;
    uint32_t k = 0;
    if (s3_batch_last_timing(b, &tot, &srch, &k) == S3_OK) s.deviceMs = tot;
    return S3_OK;
}

// ... and the download of its results into the slot's pinned staging
static int runDownload(s3_stream *st, Slot &s)
{
    s3_batch *b = s.b;
    int rc = S3_OK;
    const uint32_t n = s.n;
    HIPCHK(hipMemcpyAsync(s.hOffsets, b->dOutOffsets, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(s.hPhase, b->dPhase, n, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(s.hWithError, b->dFinalWithError, (size_t)n * 32, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    const size_t total = s.hOffsets[n];
    if ((rc = growOut(s, total ? total : 1))) return rc;
    if (total) {
        HIPCHK(hipMemcpyAsync(s.hPos, b->dOutPos, total * 4, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipMemcpyAsync(s.hMeta, b->dOutMeta, total * 4, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    return S3_OK;
}

static void workerLoop(s3_stream *st, int k)
{
    Slot &s = st->slots[(size_t)k];
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(st->mu);
            st->cv.wait(lk, [&] { return st->quit || (s.state == Slot::QUEUED && s.ticket == st->nextStart && st->running < st->deviceJobs); });
            if (st->quit) return;
            st->running++;
            st->nextStart++;
        }
        st->cv.notify_all();
        int rc = runLadder(st, s);
        {
            std::lock_guard<std::mutex> g(st->mu);
            st->running--;
        }
        st->cv.notify_all();
        if (!rc) rc = runDownload(st, s);
        {
            std::lock_guard<std::mutex> g(st->mu);
            s.rc = rc;
            s.err = rc ? s3_last_error() : "";
            s.state = Slot::DONE;
        }
        st->cv.notify_all();
    }
}

extern "C" void s3_stream_free(s3_stream *st)
{
    if (!st) return;
    if (st->maker.joinable()) st->maker.join();
    {
        std::unique_lock<std::mutex> lk(st->mu);
        // let queued batches finish: their workers hold device work
        st->cv.wait(lk, [&] { for (const Slot &s : st->slots) if (s.state == Slot::QUEUED) return false; return true; });
        st->quit = true;
    }
    st->cv.notify_all();
    for (Slot &s : st->slots) if (s.worker.joinable()) s.worker.join();
    (void)hipSetDevice(st->ix->dev);
    for (Slot &s : st->slots) {
        if (s.b) s3_batch_free(s.b);
        for (void *p : {(void *)s.hQueries, (void *)s.hLens, (void *)s.hSeedLens, (void *)s.hPhase, (void *)s.hWithError, (void *)s.hOffsets, (void *)s.hPos, (void *)s.hMeta})
            pinnedFree(p);
    }
    delete st;
}

extern "C" int s3_stream_create(s3_index *ix, uint32_t maxReads, uint32_t wordPerQuery, int depth, s3_stream **outp)
{
    if (!ix || !outp || maxReads == 0 || wordPerQuery == 0 || depth < 1 || depth > 8) return fail(S3_EINVAL, "s3_stream_create: bad argument");
    HIPCHK(hipSetDevice(ix->dev));
    s3_stream *st = new s3_stream();
    st->ix = ix; st->maxReads = maxReads; st->wpq = wordPerQuery;
    st->deviceJobs = (int)envSize("S3DP_STREAM_DEVICE_JOBS", (size_t)depth);
    if (st->deviceJobs < 1 || st->deviceJobs > depth) st->deviceJobs = depth;
    st->slots.resize((size_t)depth);
    // a slot = a batch on the device + pinned staging on the host: 0.04 to 0.2 s each (pinning touches every page).  The first slot is made here; the
    // others by a helper thread while the first batches are under way -- s3_stream_acquire waits for a slot that is not there yet as it waits for a busy one.
    auto makeSlot = [ix, maxReads, wordPerQuery, depth](Slot &s) -> int {
        HIPCHK(hipSetDevice(ix->dev));
        int rc = s3_batch_create(ix, maxReads, wordPerQuery, &s.b);
        if (rc) return rc;
        // the slots share the device: half-width search grids, so that the kernels of two batches are resident together (S3DP_STREAM_BLOCKS_PER_CU)
        if (depth > 1) s.b->blocksPerCu = (unsigned)envSize("S3DP_STREAM_BLOCKS_PER_CU", 2);
#define PIN(p, bytes) do { hipError_t e_ = pinnedAlloc(0, (void **)&(p), (bytes), (bytes), nullptr); \
        if (e_ != hipSuccess) return fail(S3_ENOMEM, std::string("hipHostMalloc(stream staging): ") + hipGetErrorString(e_)); } while (0)
        size_t sz[6];
        slotStagingSizes(maxReads, wordPerQuery, sz);
        PIN(s.hQueries, sz[0]); PIN(s.hLens, sz[1]); PIN(s.hSeedLens, sz[2]);
        PIN(s.hPhase, sz[3]); PIN(s.hWithError, sz[4]); PIN(s.hOffsets, sz[5]);
#undef PIN
        // pinned space for the occurrence lists: S3DP_STREAM_OCC_PER_READ entries per read to begin with (default 4; pinning more up front was measured slower than growing); a list that outgrows it costs a
        // hipHostFree + hipHostMalloc of twice the size in the middle of a run (pinning touches every page: tens of milliseconds per 100 MB)
        return growOut(s, (size_t)maxReads * envSize("S3DP_STREAM_OCC_PER_READ", 4) + 65536);
    };
    {
        const int rc = makeSlot(st->slots[0]);
        if (rc) { const std::string keep = s3_last_error(); s3_stream_free(st); return fail(rc, keep); }
        st->slots[0].state = Slot::FREE;
    }
    for (int k = 0; k < depth; k++) st->slots[(size_t)k].worker = std::thread(workerLoop, st, k);
    if (depth > 1)
        st->maker = std::thread([st, makeSlot, depth] {
            for (int k = 1; k < depth; k++) {
                Slot &s = st->slots[(size_t)k];
                const int rc = makeSlot(s);
                {
                    std::lock_guard<std::mutex> g(st->mu);
                    if (rc) { s.state = Slot::FAILED; s.rc = rc; st->makerErr = s3_last_error(); }
                    else s.state = Slot::FREE;
                }
                st->cv.notify_all();
                if (rc) {
                    std::lock_guard<std::mutex> g(st->mu);
                    for (int j = k + 1; j < depth; j++) { st->slots[(size_t)j].state = Slot::FAILED; st->slots[(size_t)j].rc = rc; }
                    st->cv.notify_all();
                    return;
                }
            }
        });
    *outp = st;
    return S3_OK;
}

extern "C" int s3_stream_acquire(s3_stream *st, uint32_t **queries, uint32_t **readLengths, uint32_t **seedLengths)
{
    if (!st || !queries || !readLengths) return fail(S3_EINVAL, "s3_stream_acquire: null argument");
    std::unique_lock<std::mutex> lk(st->mu);
    if (st->acquired >= 0) return fail(S3_EINVAL, "s3_stream_acquire: the previous slot was not submitted");
    // slots are used round-robin, so that results can be handed back in submission order without bookkeeping
    const int k = (int)(st->nextTicket % st->slots.size());
    Slot &s = st->slots[(size_t)k];
    st->cv.wait(lk, [&] { return s.state == Slot::FREE || s.state == Slot::FAILED; });
    if (s.state == Slot::FAILED) return fail(s.rc ? s.rc : S3_ENOMEM, "s3_stream_acquire: a slot of the stream could not be made: " + st->makerErr);
    s.state = Slot::ACQUIRED;
    st->acquired = k;
    *queries = s.hQueries; *readLengths = s.hLens;
    if (seedLengths) *seedLengths = s.hSeedLens;
    return S3_OK;
}

extern "C" int s3_stream_submit(s3_stream *st, uint32_t n, int useSeedLengths, uint32_t numMismatch, uint32_t maxOutputPerRead, int reportType, uint64_t *ticket)
{
    if (!st) return fail(S3_EINVAL, "s3_stream_submit: null stream");
    if (n == 0 || n > st->maxReads) return fail(S3_EINVAL, "s3_stream_submit: bad read count");
    {
        std::lock_guard<std::mutex> g(st->mu);
        if (st->acquired < 0) return fail(S3_EINVAL, "s3_stream_submit: no slot acquired");
        Slot &s = st->slots[(size_t)st->acquired];
        s.n = n; s.mm = numMismatch; s.maxOut = maxOutputPerRead; s.reportType = reportType; s.useSeeds = useSeedLengths != 0;
        s.ticket = st->nextTicket++;
        if (ticket) *ticket = s.ticket;
        s.state = Slot::QUEUED;
        st->acquired = -1;
    }
    st->cv.notify_all();
    return S3_OK;
}

extern "C" int s3_stream_next(s3_stream *st, s3_stream_result *res)
{
    if (!st || !res) return fail(S3_EINVAL, "s3_stream_next: null argument");
    std::unique_lock<std::mutex> lk(st->mu);
    if (st->nextResult >= st->nextTicket) return fail(S3_EINVAL, "s3_stream_next: nothing submitted");
    Slot &s = st->slots[(size_t)(st->nextResult % st->slots.size())];
    st->cv.wait(lk, [&] { return s.state == Slot::DONE; });
    if (s.rc) return fail(s.rc, s.err);
    res->ticket = s.ticket; res->n = s.n;
    res->phase = s.hPhase; res->withError = s.hWithError; res->occOffsets = s.hOffsets; res->occPos = s.hPos; res->occMeta = s.hMeta;
    res->deviceMs = s.deviceMs;
    res->batch = s.b;
    return S3_OK;
}

extern "C" int s3_stream_release(s3_stream *st)
{
    if (!st) return fail(S3_EINVAL, "s3_stream_release: null stream");
    {
        std::lock_guard<std::mutex> g(st->mu);
        if (st->nextResult >= st->nextTicket) return fail(S3_EINVAL, "s3_stream_release: nothing to release");
        Slot &s = st->slots[(size_t)(st->nextResult % st->slots.size())];
        if (s.state != Slot::DONE) return fail(S3_EINVAL, "s3_stream_release: the oldest batch has not been fetched with s3_stream_next");
        s.state = Slot::FREE;
        st->nextResult++;
    }
    st->cv.notify_all();
    return S3_OK;
}
