// s3dp_seed.hip -- seeding for the DP stages (s3_seed_candidates): seeds are packed from the resident reads into a second batch, searched with
// the ordinary search passes, and their hits sorted and merged into candidate positions on the device.
#include <hipcub/hipcub.hpp>
#include "s3dp_internal.h"

using namespace s3;
using namespace s3i;

// ---------------------------------------------------------------------------------------------
// Seeding for the DP stages of single reads: seeds cut from the reads at the tabulated offsets, searched exactly, the seeds
// without a hit once more with one mismatch, every hit turned into an estimated read position, positions of one read and strand merged.
// Replaces, of the reference: SingleEndSeedingBatch::pack / packSeeds (DV-DPfunctions.cu:1058-1100), single_1_mismatch_alignment2
// (alignment.cu:1839-1894) + the hit limit of hostKernelSingle (CPUfunctions.cpp:2843-2876: a seed with more than maxHitNum hits is
// dropped), decodePositions (:1143-1219, sorted by strand, read, position) and singleMerge (:1102-1141, positions closer than
// DPS_DIVIDE_GAP = 50 collapse into the first).  Everything runs on the device next to the resident reads; the seed batch is an
// ordinary s3_batch.  Pinned through the reference's own seeding engines in the oracle harness (tests/test_gpu_dp_pipeline.py); also tested against a restatement built on the pinned search oracle.
// ---------------------------------------------------------------------------------------------

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
                               const uint32_t *outOffsets, const uint32_t *occPos, const uint32_t *occMeta, uint32_t strandBit, uint64_t *keys)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nSeeds) return;
    const SeedDesc d = seeds[j];
    const uint32_t cnt = occCount[j], src = occStart[j], dst = outOffsets[j], readLen = readLens[d.readId];
    for (uint32_t k = 0; k < cnt; k++) {
        const uint32_t pos = occPos[src + k], strand = occMeta[src + k] & 0xFFu;
        const uint32_t est = strand == 1u ? pos - d.off : pos + d.len + d.off - readLen;
        keys[dst + k] = (uint64_t)(d.keyHi | (strand == 2u ? 1u << strandBit : 0u)) << 32 | est;
    }
}

// Search the seeds and return their hits as sorted 64-bit keys: (keyHi | strand bit) << 32 | estimated read position.
int s3i::seedSearchKeys(s3_batch *reads, s3_batch *seedBatch, const std::vector<SeedDesc> &seeds, uint32_t maxHitNum, uint32_t strandBit,
                        std::vector<uint64_t> &keysOut, std::vector<uint8_t> &statusOut)
{
    HIPCHK(hipSetDevice(reads->ix->dev));
    const uint32_t nSeeds = (uint32_t)seeds.size();
    if (nSeeds > seedBatch->maxReads) return fail(S3_EINVAL, "s3_seed_candidates: seed batch too small");
    if (seedBatch->wpq * 16u < 70u) return fail(S3_EINVAL, "s3_seed_candidates: the seed batch needs wordPerQuery >= 8");
    s3_batch *b = seedBatch;
    hipStream_t st = b->stream;
    SeedDesc *dSeeds = nullptr;
    uint8_t *dRetry = nullptr, *dStatus = nullptr;
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
    b->searchLaunches = 0; b->evSearchUsed = 0; b->evCeUsed = 0; b->occUsed = 0; b->recordsTotal = 0;
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
        if (!idsIn) { launchIota(b->dIota, nA, st); idsIn = b->dIota; }
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
    keysOut.clear();
    if (total) {
        size_t sortBytes = 0;
        SCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, sortBytes, dKeys, dKeysOut, (int)total, 0, 64, st));
        if ((rc = talloc((void **)&dKeys, (size_t)total * 8)) || (rc = talloc((void **)&dKeysOut, (size_t)total * 8))||
            (rc = talloc((void **)&dSortTmp, sortBytes))) { cleanup(); return rc; }
        hipLaunchKernelGGL(seedKeysKernel, dim3((nSeeds + 255) / 256), dim3(256), 0, st, nSeeds, dSeeds, reads->dLens, b->dOccStart, b->dOccCount, b->dOutOffsets,
                           b->dOccPos, b->dOccMeta, strandBit, dKeys);
        SCHK(hipcub::DeviceRadixSort::SortKeys(dSortTmp, sortBytes, dKeys, dKeysOut, (int)total, 0, 64, st));
        keysOut.resize(total);
        SCHK(hipMemcpyAsync(keysOut.data(), dKeysOut, (size_t)total * 8, hipMemcpyDeviceToHost, st));
        SCHK(hipStreamSynchronize(st));
    }
    statusOut.resize(nSeeds);
    if (nSeeds) SCHK(d2h(statusOut.data(), dStatus, nSeeds, st));
#undef SCHK
    cleanup();
    return checkFlags(b);
}

extern "C" int s3_seed_candidates(s3_batch *reads, s3_batch *seedBatch, const uint32_t *readLengths, const uint32_t *readIDs, size_t nReadIDs, int stage,
                                  uint32_t maxHitNum, s3_seed_candidate **candOut, size_t *nCandOut, uint8_t **seedStatusOut, size_t *nSeedsOut)
{
    if (!reads || !seedBatch || !readLengths || !candOut || !nCandOut) return fail(S3_EINVAL, "s3_seed_candidates: null argument");
    if (reads->ix != seedBatch->ix || reads == seedBatch) return fail(S3_EINVAL, "s3_seed_candidates: the seed batch must be a second batch on the same index");
    std::vector<SeedDesc> seeds;
    const size_t nSel = readIDs ? nReadIDs : reads->n;
    for (size_t k = 0; k < nSel; k++) {
        const uint32_t r = readIDs ? readIDs[k] : (uint32_t)k;
        if (r >= reads->n) return fail(S3_EINVAL, "s3_seed_candidates: read index out of range");
        int seedLength = 0, num = 0, pos[128];
        if (s3_seed_positions(stage, (int)readLengths[r], &seedLength, pos, &num) != S3_OK) return fail(S3_EINVAL, "s3_seed_candidates: bad stage");
        for (int i = 0; i < num; i++) {
            if (pos[i] < 0 || seedLength < 16 || (uint32_t)(pos[i] + seedLength) > readLengths[r]) continue;   // reads too short for the stage's seeds
            seeds.push_back(SeedDesc{r, (uint32_t)pos[i], (uint32_t)seedLength, r});
        }
    }
    std::vector<uint64_t> keys;
    std::vector<uint8_t> status;
    int rc = seedSearchKeys(reads, seedBatch, seeds, maxHitNum, 31, keys, status);
    if (rc) return rc;
    // singleMerge (DV-DPfunctions.cu:1102-1141): within one (strand, read) run the first position is kept, then every position more than
    // DPS_DIVIDE_GAP = 50 beyond the last kept one
    std::vector<s3_seed_candidate> out;
    for (size_t i = 0; i < keys.size();) {
        const uint64_t run = keys[i] >> 32;
        uint32_t prev = (uint32_t)keys[i];
        out.push_back(s3_seed_candidate{(uint32_t)(run & 0x7FFFFFFFu), prev, (uint8_t)((run >> 31) ? 2 : 1), {0, 0, 0}});
        size_t k = i + 1;
        for (; k < keys.size() && (keys[k] >> 32) == run; k++) {
            const uint32_t cur = (uint32_t)keys[k];
            if ((uint64_t)prev + 50u < cur) { out.push_back(s3_seed_candidate{(uint32_t)(run & 0x7FFFFFFFu), cur, (uint8_t)((run >> 31) ? 2 : 1), {0, 0, 0}}); prev = cur; }
        }
        i = k;
    }
    const size_t nSeeds = status.size();
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
