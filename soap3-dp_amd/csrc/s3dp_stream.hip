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
    bool quit = false;
    std::thread maker;                            // makes the slots behind the first one while the first batches are under way
    std::string makerErr;
};

static int growOut(Slot &s, size_t need)
{
    if (need <= s.occCap) return S3_OK;
    size_t cap = s.occCap ? s.occCap : 1 << 20;
    while (cap < need) cap *= 2;
    if (s.hPos) (void)hipHostFree(s.hPos);
    if (s.hMeta) (void)hipHostFree(s.hMeta);
    s.hPos = s.hMeta = nullptr; s.occCap = 0;
    HIPCHK(hipHostMalloc((void **)&s.hPos, cap * 4, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&s.hMeta, cap * 4, hipHostMallocDefault));
    s.occCap = cap;
    return S3_OK;
}

static int runJob(s3_stream *st, Slot &s)
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
            st->cv.wait(lk, [&] { return st->quit || s.state == Slot::QUEUED; });
            if (st->quit) return;
        }
        const int rc = runJob(st, s);
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
            if (p) (void)hipHostFree(p);
    }
    delete st;
}

extern "C" int s3_stream_create(s3_index *ix, uint32_t maxReads, uint32_t wordPerQuery, int depth, s3_stream **outp)
{
    if (!ix || !outp || maxReads == 0 || wordPerQuery == 0 || depth < 1 || depth > 8) return fail(S3_EINVAL, "s3_stream_create: bad argument");
    HIPCHK(hipSetDevice(ix->dev));
    s3_stream *st = new s3_stream();
    st->ix = ix; st->maxReads = maxReads; st->wpq = wordPerQuery;
    st->slots.resize((size_t)depth);
    const size_t roundUp = ((size_t)maxReads + 31) / 32 * 32;
    // a slot = a batch on the device + pinned staging on the host: 0.04 to 0.2 s each (pinning touches every page).  The first slot is made here; the
    // others by a helper thread while the first batches are under way -- s3_stream_acquire waits for a slot that is not there yet as it waits for a busy one.
    auto makeSlot = [ix, maxReads, wordPerQuery, depth, roundUp](Slot &s) -> int {
        HIPCHK(hipSetDevice(ix->dev));
        int rc = s3_batch_create(ix, maxReads, wordPerQuery, &s.b);
        if (rc) return rc;
        // the slots share the device: half-width search grids, so that the kernels of two batches are resident together (S3DP_STREAM_BLOCKS_PER_CU)
        if (depth > 1) s.b->blocksPerCu = (unsigned)envSize("S3DP_STREAM_BLOCKS_PER_CU", 2);
#define PIN(p, bytes) do { hipError_t e_ = hipHostMalloc((void **)&(p), (bytes), hipHostMallocDefault); \
        if (e_ != hipSuccess) return fail(S3_ENOMEM, std::string("hipHostMalloc(stream staging): ") + hipGetErrorString(e_)); } while (0)
        PIN(s.hQueries, roundUp * wordPerQuery * 4); PIN(s.hLens, roundUp * 4); PIN(s.hSeedLens, roundUp * 4);
        PIN(s.hPhase, roundUp); PIN(s.hWithError, roundUp * 32); PIN(s.hOffsets, (roundUp + 1) * 4);
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
