// s3dp_group.hip -- several devices behind the C-ABI (include/s3dp.h: s3_group_*).
//
// The path shards by reads: every device holds the whole index (replicated in its own HBM) and aligns its own batches; nothing is exchanged
// while reads are aligned.  The one exchange step is the sum of the alignment counters at the end (the `[Main] Overall ...` numbers of
// SOAP3-DP.cu:816-830): one ncclAllReduce over the group's devices (RCCL, xGMI between the GPUs of a node).
#include <rccl/rccl.h>
#include <future>
#include <thread>
#include "s3dp_internal.h"

using namespace s3i;

struct s3_group {
    std::vector<int> devices;
    std::vector<s3_index *> indexes;
    bool distinct = true;                    // RCCL needs one communicator rank per DISTINCT device
    // the communicators of a group of several distinct devices are made while the index is uploaded (ncclCommInitAll takes seconds: at the end of a
    // run it would be the longest single step of a short job); a group of one device has nothing to exchange and never makes one
    std::vector<ncclComm_t> comm;
    std::future<ncclResult_t> commReady;
    bool commMade = false;
};

extern "C" void s3_group_free(s3_group *g)
{
    if (!g) return;
    if (g->commReady.valid()) g->commMade = g->commReady.get() == ncclSuccess;
    if (g->commMade) for (ncclComm_t c : g->comm) (void)ncclCommDestroy(c);
    for (s3_index *ix : g->indexes) if (ix) s3_index_free(ix);
    delete g;
}

extern "C" int s3_group_create(const int *devices, int nDevices, const s3_index_desc *desc, s3_group **outp)
{
    if (!devices || nDevices < 1 || nDevices > 64 || !desc || !outp) return fail(S3_EINVAL, "s3_group_create: bad argument");
    s3_group *g = new s3_group();
    g->devices.assign(devices, devices + nDevices);
    g->indexes.assign((size_t)nDevices, nullptr);
    for (int a = 0; a < nDevices; a++) for (int b = a + 1; b < nDevices; b++) if (devices[a] == devices[b]) g->distinct = false;
    if (g->distinct && nDevices > 1) {
        g->comm.assign((size_t)nDevices, nullptr);
        g->commReady = std::async(std::launch::async, [g] { return ncclCommInitAll(g->comm.data(), (int)g->devices.size(), g->devices.data()); });
    }
    // the index is replicated: one upload per member, all at once (each is bound by its own PCIe link and its own table kernels)
    std::vector<int> rc((size_t)nDevices, S3_OK);
    std::vector<std::string> err((size_t)nDevices);
    std::vector<std::thread> th;
    for (int k = 0; k < nDevices; k++) th.emplace_back([&, k] {
        rc[(size_t)k] = s3_index_upload(devices[k], desc, &g->indexes[(size_t)k]);
        if (rc[(size_t)k]) err[(size_t)k] = s3_last_error();
    });
    for (auto &t : th) t.join();
    for (int k = 0; k < nDevices; k++) if (rc[(size_t)k]) { const int r = rc[(size_t)k]; const std::string e = err[(size_t)k]; s3_group_free(g); return fail(r, e); }
    *outp = g;
    return S3_OK;
}

extern "C" int s3_group_size(const s3_group *g) { return g ? (int)g->devices.size() : 0; }
extern "C" s3_index *s3_group_index(s3_group *g, int member) { return g && member >= 0 && member < (int)g->indexes.size() ? g->indexes[(size_t)member] : nullptr; }
extern "C" int s3_group_device(const s3_group *g, int member) { return g && member >= 0 && member < (int)g->devices.size() ? g->devices[(size_t)member] : -1; }

// batches are dealt to the members in turn; a batch holds whole pairs (reads 2p / 2p+1 stay together, CPUfunctions.cpp:1982,2029)
extern "C" int s3_group_member_of_batch(const s3_group *g, uint64_t batchNumber) { return g && !g->devices.empty() ? (int)(batchNumber % g->devices.size()) : -1; }

// counters: nMembers rows of nCounters values (row k = what member k counted); on return every row holds the column sums
extern "C" int s3_group_allreduce(s3_group *g, uint64_t *counters, int nCounters)
{
    if (!g || !counters || nCounters < 1 || nCounters > 1024) return fail(S3_EINVAL, "s3_group_allreduce: bad argument");
    const int n = (int)g->devices.size();
    if (n == 1) return S3_OK;                                                 // one member: its counters are the sums
    if (!g->distinct) {
        // members that share a device (a test on one GPU) cannot be ranks of one communicator: their sum is taken here
        std::vector<uint64_t> sum((size_t)nCounters, 0);
        for (int k = 0; k < n; k++) for (int c = 0; c < nCounters; c++) sum[(size_t)c] += counters[(size_t)k * nCounters + c];
        for (int k = 0; k < n; k++) for (int c = 0; c < nCounters; c++) counters[(size_t)k * nCounters + c] = sum[(size_t)c];
        return S3_OK;
    }
    if (g->commReady.valid()) {
        const ncclResult_t ci = g->commReady.get();
        if (ci != ncclSuccess) return fail(S3_EHIP, std::string("ncclCommInitAll: ") + ncclGetErrorString(ci));
        g->commMade = true;
    }
    if (!g->commMade) return fail(S3_EHIP, "s3_group_allreduce: the group has no communicators");
    std::vector<ncclComm_t> &comm = g->comm;
    std::vector<hipStream_t> st((size_t)n, nullptr);
    std::vector<uint64_t *> buf((size_t)n, nullptr);
    int rc = S3_OK;
    std::string msg;
    auto cleanup = [&]() {
        for (int k = 0; k < n; k++) {
            (void)hipSetDevice(g->devices[(size_t)k]);
            if (buf[(size_t)k]) (void)hipFree(buf[(size_t)k]);
            if (st[(size_t)k]) (void)hipStreamDestroy(st[(size_t)k]);
        }
    };
#define GCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { msg = std::string(#x) + ": " + hipGetErrorString(e_); rc = S3_EHIP; } } while (0)
    for (int k = 0; k < n && rc == S3_OK; k++) {
        GCHK(hipSetDevice(g->devices[(size_t)k]));
        if (rc == S3_OK) GCHK(hipStreamCreateWithFlags(&st[(size_t)k], hipStreamNonBlocking));
        if (rc == S3_OK) GCHK(hipMalloc((void **)&buf[(size_t)k], (size_t)nCounters * 8));
        if (rc == S3_OK) GCHK(hipMemcpyAsync(buf[(size_t)k], counters + (size_t)k * nCounters, (size_t)nCounters * 8, hipMemcpyHostToDevice, st[(size_t)k]));
    }
    if (rc == S3_OK) {
        ncclResult_t r = ncclGroupStart();
        for (int k = 0; k < n && r == ncclSuccess; k++) r = ncclAllReduce(buf[(size_t)k], buf[(size_t)k], (size_t)nCounters, ncclUint64, ncclSum, comm[(size_t)k], st[(size_t)k]);
        const ncclResult_t r2 = ncclGroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) { msg = std::string("ncclAllReduce: ") + ncclGetErrorString(r); rc = S3_EHIP; }
    }
    for (int k = 0; k < n && rc == S3_OK; k++) {
        GCHK(hipSetDevice(g->devices[(size_t)k]));
        if (rc == S3_OK) GCHK(hipMemcpyAsync(counters + (size_t)k * nCounters, buf[(size_t)k], (size_t)nCounters * 8, hipMemcpyDeviceToHost, st[(size_t)k]));
        if (rc == S3_OK) GCHK(hipStreamSynchronize(st[(size_t)k]));
    }
#undef GCHK
    cleanup();
    return rc == S3_OK ? S3_OK : fail(rc, msg);
}
