// s3dp_internal.h -- private to libs3dp.so: the objects behind the C-ABI handles and the helpers the source files share.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <chrono>
#include <shared_mutex>
#include "../../include/s3dp.h"
#include "s3dp_device.h"
#include "s3dp_model.h"

namespace s3i {
// blocking device-to-host copy on the object's own stream (the streams are non-blocking: nothing here may lean on the legacy default stream,
// or two batches driven by two host threads would serialise)
inline hipError_t d2h(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}
int fail(int code, const std::string &msg);               // records the message for s3_last_error and returns `code`

// Scratch memory of the DP and seeding engines.  Every engine call needs a dozen device buffers whose sizes depend on its input; hipMalloc costs tens of
// microseconds, hipFree a DEVICE-WIDE synchronisation -- forty of each per chunk of pairs, on a thread that runs beside another chunk's search.  scratchAlloc
// hands out blocks of a per-device cache (sizes rounded up to powers of two from 64 KB), scratchFree puts them back; nothing is given to the driver until the
// cache holds more than 8 GB or the index of the device is freed (scratchTrim).  A block comes back with the stream it was last used on: whoever takes it for
// another stream waits for that stream first -- what hipFree's synchronisation used to guarantee.
hipError_t scratchAlloc(void **p, size_t bytes, hipStream_t stream);
void scratchFree(void *p, hipStream_t stream);
void scratchTrim(int device);
void scratchStreamGone(hipStream_t stream);               // the owner of `stream` has synchronised it and is about to destroy it: cached blocks forget it
}
#define HIPCHK(x)                                                                                          \
    do {                                                                                                   \
        hipError_t e_ = (x);                                                                               \
        if (e_ != hipSuccess) return s3i::fail(S3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_));  \
    } while (0)

struct s3_index {
    int       dev = 0;
    s3::DevIndex  d{};
    // The flank tables of check-and-extend (d.flankBack / d.flankFwd, 2 x 16 B per suffix-array entry) are a convenience the device may not have room for once the
    // batches grow: an allocation of this library that finds the device full releases them -- the reverse index's first -- and tries again (mallocOrRelease);
    // their tasks then take the suffix array and the text, as all did until round 6.  Launches that read the tables hold this lock shared while they are enqueued;
    // the release takes it exclusively and waits for the device before it frees.
    std::shared_mutex flankMu;
    std::vector<void *> allocs;
    S3Model  *models[S3_MAX_MISMATCH + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // device, indexed by read length
    size_t    bytes = 0;
};

struct s3_batch {
    s3_index *ix = nullptr;
    hipStream_t stream = nullptr;
    uint32_t maxReads = 0, wpq = 0, n = 0;
    size_t   maxItems = 0;
    uint32_t recCap = 0, occCap = 0, ordCap = 0;
    // inputs
    uint32_t *dQueries = nullptr, *dLens = nullptr, *dReadsT = nullptr, *dSeedLens = nullptr;
    bool longMode = false;          // seed lengths set: the reference's long-read mode
    uint32_t fullUniformLen = 0;
    // search / ordering work buffers
    uint4    *dRecs = nullptr;
    uint32_t *dOrdIdx = nullptr;                     // the ordered array: per sequence number of a pass the number of the record scattered there (MergeArgs::ordIdx)
    uint4    *dRecAux = nullptr;                      // SearchOut::recAux: 16 B per slot of dRecs
    uint8_t  *dOrdValid = nullptr;                    // per slot of dOrdIdx: holds a record (round 4: the holes are no longer written and read as 16-byte entries)
    uint32_t *dRecCount = nullptr, *dFlags = nullptr, *dItemCounts = nullptr, *dItemOffsets = nullptr, *dTaskRange = nullptr;
    unsigned long long *dCounters = nullptr, *dWorkHead = nullptr;
    unsigned long long *dSum64 = nullptr;            // 64-bit total next to a 32-bit scan (sum64Async): a wrapped total must not pass as a small one
    uint8_t  *dWasHeavy = nullptr;                   // per read: the mismatch budget + 1 at which a first pass listed it (reset when the reads change and when a ladder starts)
    uint8_t  *dHeavyFlags = nullptr;                 // reads the first search pass listed for the exact pass (s3dp_first.hip)
    uint32_t *dHeavyList = nullptr, *dHeavyCount = nullptr, *dHeavyIndex = nullptr;
    uint32_t *dLongList = nullptr, *dLongCount = nullptr;      // reads whose slice of the ordered array is walked by a whole wave (merge / decode)
    uint32_t *dItemAcc = nullptr; size_t accCap = 0;            // per listed read and item: certain occurrences (second search pass)
    uint4 *dTrunk = nullptr; size_t trunkCap = 0;              // trunk states of the second search pass
    size_t countCap = 0;                                       // entries of dItemCounts / dItemOffsets (items + units of the listed reads + 1 + the counters of adopted frames)
    // unit pass (s3dp_unit.hip)
    uint32_t *dListedRead = nullptr; size_t listedCap = 0;     // per listed read: its index in readsT
    uint4 *dUnitStack = nullptr; size_t unitStackWaves = 0;    // depth-first frames of the unit-pass grid
    uint32_t *dChainNext = nullptr; size_t chainCap = 0;       // per unit and per adopted frame: the counter that follows
    uint32_t *dDynOwner = nullptr, *dDynLocal = nullptr, *dDynUsed = nullptr; size_t dynCap = 0;
    unsigned long long *dOrderKeys[2] = {nullptr, nullptr}; uint32_t *dOrderVals[2] = {nullptr, nullptr}; size_t orderCap = 0;   // sort buffers: order of the listed reads / of the batch
    s3::ScatterDyn scatterDyn{nullptr, nullptr, 0, 0};         // of the last search pass
    uint32_t mergeUnitBase = 0, mergeUnitsPerRead = 0;         // of the last search pass (0: no read was taken apart)
    uint32_t uniformLen = 0;
    unsigned unitGroup = 1;                           // first-mismatch positions per unit of the second pass (SearchIn::unitGroup): set by the entry point -- single-end ladders 2, everything else 1
    unsigned blocksPerCu = 0;                         // width of the persistent search grid (0: all that is resident at once)
    uint32_t *dKept = nullptr, *dOccCounts = nullptr, *dCeCounts = nullptr, *dTotalOcc = nullptr, *dWithError = nullptr;
    uint32_t *dHitOffsets = nullptr;
    s3::DevHit   *dHits = nullptr;
    char     *dTemp = nullptr; size_t tempBytes = 0;
    // ladder
    uint32_t *dActive[2] = {nullptr, nullptr}, *dNumSel = nullptr, *dIota = nullptr;
    uint8_t  *dState = nullptr, *dSelFlags = nullptr;
    uint32_t *dOccOffsets = nullptr, *dMasked = nullptr;
    uint32_t *dOccPos = nullptr, *dOccMeta = nullptr;
    uint8_t  *dPhase = nullptr;
    uint32_t *dFinalWithError = nullptr, *dOccStart = nullptr, *dOccCount = nullptr;
    uint32_t *dOutOffsets = nullptr, *dOutPos = nullptr, *dOutMeta = nullptr; size_t outCap = 0;   // read-ordered final lists (ensureOutLists)
    // paired-end ladder (allocated by the first s3_batch_align_pair)
    bool pairReady = false;
    s3::DevPair  *dPairs = nullptr;
    uint8_t  *dPairMin = nullptr, *dPairNext = nullptr;
    uint32_t *dReadList = nullptr, *dPairActive[2] = {nullptr, nullptr}, *dBestIdx = nullptr, *dXaCount = nullptr, *dXaOffsets = nullptr;
    uint32_t *dXaStart = nullptr, *dXaCountOut = nullptr, *dXaOutOffsets = nullptr, *dOccCountIn = nullptr, *dSegBegin = nullptr, *dSegEnd = nullptr;
    s3::DevPairXa *dXa = nullptr, *dXaOut = nullptr;
    uint32_t *dPairHeavyList = nullptr, *dPairHeavyCount = nullptr, *dBestRank = nullptr, *dBestLocal = nullptr;   // pairs with long lists (s3dp_pair.hip)
    uint32_t xaCap = 0, xaUsed = 0;
    uint64_t *dSortKeys[2] = {nullptr, nullptr};
    uint32_t *dSortMeta = nullptr;
    char     *dSortTemp = nullptr;
    size_t    sortItems = 0, sortTempBytes = 0;
    // routing of the unpaired pairs (s3dp_route.hip; allocated by the first s3_batch_route_pairs)
    uint32_t *dRouteKept = nullptr, *dRouteCounts = nullptr, *dRouteOffs = nullptr, *dRouteOcc = nullptr;
    uint8_t  *dRouteThr = nullptr, *dRoute = nullptr, *dPairRoute = nullptr;
    char     *dRouteTemp = nullptr; size_t routeTempBytes = 0, routeCap = 0;
    bool routed = false;
    s3_pair_routing routing{};
    std::vector<uint8_t> hPairRoute;               // host copies of the last routing: per pair the S3_ROUTE_* bits ...
    std::vector<uint32_t> hRouteOffs;              // ... and where every read's entries start in each of the three lists ([3][n] + 1)
    std::vector<void *> allocs;
    // state of the last call
    bool failed = false;           // a buffer was lost while growing (bgrow): every later call on this batch returns S3_ENOMEM
    int lastKind = 0;              // 1 = search, 2 = align (single), 3 = align (pairs)
    uint32_t lastNumCases = 0;
    bool counting = false;
    hipEvent_t evTotal[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> evSearch, evCe;          // per launch of the stepping kernel / of the check-and-extend kernel
    std::vector<uint8_t> evSearchRole;               // per pair of evSearch: 0 first pass, 1 trunk pass, 2 unit pass (3 = check-and-extend marks the pairs of evCe)
    uint32_t evSearchUsed = 0, evCeUsed = 0;         // event pairs in use
    uint32_t searchLaunches = 0;                     // search passes of the last call
    uint32_t occUsed = 0;
    uint64_t recordsTotal = 0;      // records of all search passes of the last call (read back when counters are enabled)
};

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
    uint32_t *dPatMax = nullptr;           // longest pattern of the last launch (bytes with the terminator)
    size_t tableWords = 0;
    bool timed = false, folded = true;
    double totalMs = 0;                    // device time of every launch since the aligner was made (s3_dp_totals) ...
    uint64_t totalProblems = 0;            // ... the problems of those launches ...
    unsigned long long *dCells = nullptr;  // ... and their cells (device counter)
    std::vector<uint32_t> dumpClipRt;      // S3DP_DUMP_DP: the right clips as the packer set them (the kernel overwrites them)
};

namespace s3i {
// hipMalloc; on hipErrorOutOfMemory the flank tables of an index on this device are released one at a time and the allocation is tried again (s3dp_api.hip)
hipError_t mallocOrRelease(int dev, void **q, size_t bytes);
bool releaseFlankTable(int dev);
// a buffer that was replaced by a larger one is not freed on the spot -- hipFree waits for EVERY stream of the device, and a batch that grows beside two
// others in flight stalls all three (the first batches of a streamed run: a third of the device's time idle, profiles/README_r06.md section 8) -- but kept
// on a list that is freed when a batch is freed, when it passes 8 GB, or when an allocation finds the device full (before a flank table gives way)
void retireBuffer(int dev, void *p);
bool freeRetired(int dev);
// S3DP_VERBOSE: one line per allocation and growth of a batch's buffers -- when (since the first one), how large, how long it took (stderr)
void noteAlloc(const char *what, size_t bytes, std::chrono::steady_clock::time_point t0);

template <class T>
inline int balloc(s3_batch *b, T **p, size_t count)
{
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    if (!bytes) bytes = 16;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = mallocOrRelease(b->ix->dev, &q, bytes);
    if (e != hipSuccess) return fail(S3_ENOMEM, std::string("hipMalloc(batch): ") + hipGetErrorString(e));
    noteAlloc("alloc", bytes, t0);
    b->allocs.push_back(q);
    *p = (T *)q;
    return S3_OK;
}

bool growRefused();                                           // s3dp_api.hip

// replace a batch buffer by a larger one, keeping the first `keep` elements.  On failure the batch is left in a state every later call refuses
// (s3_batch::failed): the callers keep their capacities next to the pointers, and a freed buffer under an old capacity would be a wild launch.
template <class T>
inline int bgrow(s3_batch *b, T **p, size_t newCount, size_t keep)
{
    void *q = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (!keep) {
        // nothing to carry over.  Try to allocate next to the old buffer; when that does not fit (the ordering buffer of a batch of reads from repeat
        // families is tens of gigabytes) the old buffer goes first.
        hipError_t e0 = hipStreamSynchronize(b->stream);
        if (e0 != hipSuccess) return fail(S3_EHIP, std::string("grow (sync): ") + hipGetErrorString(e0));
        const bool refuse = growRefused();                        // (tests: S3DP_TEST_FAIL_GROW=n makes the n-th growth of the process find the device full)
        e0 = refuse ? hipErrorOutOfMemory : hipMalloc(&q, newCount * sizeof(T));
        if (e0 == hipErrorOutOfMemory && !refuse && freeRetired(b->ix->dev)) { (void)hipGetLastError(); e0 = hipMalloc(&q, newCount * sizeof(T)); }
        if (e0 != hipSuccess) {
            (void)hipGetLastError();
            q = nullptr;
            e0 = hipFree(*p);
            if (e0 != hipSuccess) { b->failed = true; return fail(S3_EHIP, std::string("grow (free): ") + hipGetErrorString(e0)); }
            for (auto &a : b->allocs) if (a == (void *)*p) a = nullptr;
            *p = nullptr;
            e0 = refuse ? hipErrorOutOfMemory : mallocOrRelease(b->ix->dev, &q, newCount * sizeof(T));
            if (e0 != hipSuccess) {
                (void)hipGetLastError();
                b->failed = true;                                     // the old buffer is gone and there is no new one
                return fail(S3_ENOMEM, std::string("hipMalloc(batch grow, ") + std::to_string(newCount * sizeof(T) >> 20) + " MB): " + hipGetErrorString(e0) +
                                           "; the batch is unusable from here on: free it and create a smaller one");
            }
            b->allocs.push_back(q);
            *p = (T *)q;
            return S3_OK;
        }
        bool placed = false;
        for (auto &a : b->allocs) if (*p && a == (void *)*p) { a = q; placed = true; }
        if (!placed) b->allocs.push_back(q);
        retireBuffer(b->ix->dev, *p);
        *p = (T *)q;
        noteAlloc("grow", newCount * sizeof(T), t0);
        return S3_OK;
    }
    hipError_t e = mallocOrRelease(b->ix->dev, &q, newCount * sizeof(T));
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(S3_ENOMEM, std::string("hipMalloc(batch grow, ") + std::to_string(newCount * sizeof(T) >> 20) + " MB): " + hipGetErrorString(e)); }
    e = hipMemcpyAsync(q, *p, keep * sizeof(T), hipMemcpyDeviceToDevice, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) { (void)hipFree(q); return fail(S3_EHIP, std::string("grow copy: ") + hipGetErrorString(e)); }
    for (auto &a : b->allocs) if (a == (void *)*p) a = q;
    retireBuffer(b->ix->dev, *p);
    *p = (T *)q;
    noteAlloc("grow (kept part copied)", newCount * sizeof(T), t0);
    return S3_OK;
}


static const size_t kMaxEntries = 0xFFFFFF00ull;         // record / occurrence indices are 32 bit
int growRecords(s3_batch *b);
int growOccurrences(s3_batch *b, uint64_t need);
int ensureOutLists(s3_batch *b, uint64_t need);
size_t envSize(const char *name, size_t dflt);
// search + order + merge for `nActive` slots (s3dp_api.hip); per-slot results land in dTotalOcc / dWithError / dKept / ...
int searchPass(s3_batch *b, const uint32_t *active, uint32_t nActive, uint32_t modelMm, uint32_t maxOut);
int checkFlags(s3_batch *b);
// slots[0, n) (nullptr: 0 .. n-1) in the order of the first 32 characters of the reads behind them, into b->dOrderVals[1] (s3dp_api.hip)
int orderByReadPrefix(s3_batch *b, const uint32_t *slots, uint32_t n, const uint32_t *active, uint32_t stride, uint32_t words);
// ladders (s3dp_ladder.hip)
struct LongArgs { bool on; int onlyBest, minSeedMis, needMapq; uint32_t maxHit; int carryOn; };
int decodeAndPublish(s3_batch *b, const uint32_t *active, uint32_t nActive, uint8_t want, uint8_t phase, uint32_t maxOut, uint8_t selectWhen, const LongArgs &lg);
void launchIota(uint32_t *p, uint32_t n, hipStream_t s);
void launchStateAll(uint32_t n, const uint32_t *totalOcc, uint8_t *state, hipStream_t s);
// seeding (s3dp_seed.hip)
struct SeedDesc { uint32_t readId, off, len, keyHi; };   // seed of read `readId` at `off`; keyHi = the upper key bits its hits sort by
int seedSearchKeys(s3_batch *reads, s3_batch *seedBatch, const std::vector<SeedDesc> &seeds, uint32_t maxHitNum, uint32_t strandBit,
                   std::vector<uint64_t> &keysSorted, std::vector<uint8_t> &seedStatus, uint64_t **devKeysOut = nullptr, uint32_t *nKeysOut = nullptr);
// pairEndMerge on the device (s3dp_seed.hip): the sorted position keys of both ends (device; plus `given` occurrences, new default DP) -> the pairs of
// (left, right) positions inside the insert window, ordered by the left read; perPair[k] = candidates of pairFirstReads[k] (both orientations)
struct PairCandHost { uint32_t pos[2]; uint32_t readIDLeft; };
// (givenOnDevice: `given` is a device pointer -- the new default DP's list where s3_batch_route_pairs left it)
int pairMergeOnDevice(s3_batch *b, uint64_t *dKeys, uint32_t nKeys, const s3_occ *given, size_t nGiven, bool givenOnDevice, const uint32_t *pairFirstReads, size_t nPairs,
                      int insertLow, int insertHigh, std::vector<PairCandHost> &cands, std::vector<uint32_t> &perPair);
// enqueue the 64-bit sum of counts[0, n) into b->dSum64 (uses b->dTemp: call it after the scan that shares the scratch was enqueued, same stream)
int sum64Async(s3_batch *b, const uint32_t *counts, uint32_t n);
// DP (s3dp_dpapi.hip): launch the DP kernels on what is in the aligner's device buffers
int dpRun(s3_dp *h, uint32_t n, bool hasClipLt, bool hasClipRt, bool hasAL, bool hasAR);
// the patterns of the last dpRun, one row of `pitch` bytes per problem (pitch = the longest pattern with its terminator, not the aligner's patternLen:
// a pattern is a few dozen bytes, the row it is written into eight hundred)
int dpFetchPatterns(s3_dp *h, uint32_t n, std::vector<uint8_t> &pat, uint32_t &pitch);
// the same decoded on the device: per problem [int32 insertions, deletions, soft clips, gap penalty][CIGAR text, 0-terminated] (dpDecodeKernel, s3dp_dpapi.hip)
int dpFetchDecoded(s3_dp *h, uint32_t n, std::vector<uint8_t> &rows, uint32_t &pitch);
// S3DP_DUMP_DP=<file> (tests): the problems of the last dpRun with their results, and stage separators
int dpDump(s3_dp *h, uint32_t n, bool hasClipLt, bool hasClipRt, bool hasAL, bool hasAR);
void dpDumpNote(const char *line);
}  // namespace s3i
