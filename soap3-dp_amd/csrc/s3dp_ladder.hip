// s3dp_ladder.hip -- the phase ladders of the C-ABI: single-end report types (all valid / all best / unique best / random best), long-read
// verification, and the paired-end ladder with pairing.  Host orchestration + small bookkeeping kernels; the search passes themselves are
// s3i::searchPass (s3dp_api.hip).
#include <hipcub/hipcub.hpp>
#include "s3dp_internal.h"

using namespace s3;
using namespace s3i;

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

// read-ordered copy of the occurrence lists.  A wave takes 64 reads; lists of up to 8 entries are copied by their own lane, longer ones by the whole
// wave, entry k by lane k % 64 (a read inside a repeat family holds a thousand entries: one lane per read left the launch waiting for those lanes).
__global__ void gatherOccKernel(uint32_t n, const uint32_t *occStart, const uint32_t *outOffsets, const uint32_t *pos, const uint32_t *meta,
                                uint32_t *outPos, uint32_t *outMeta)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    uint32_t cnt = 0, src = 0, dst = 0;
    if (r < n) { cnt = outOffsets[r + 1] - outOffsets[r]; src = occStart[r]; dst = outOffsets[r]; }
    if (cnt <= 8u) for (uint32_t k = 0; k < cnt; k++) { outPos[dst + k] = pos[src + k]; outMeta[dst + k] = meta[src + k]; }
    unsigned long long big = __ballot(cnt > 8u);
    while (big) {
        const int l = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t c = (uint32_t)__shfl((int)cnt, l), s0 = (uint32_t)__shfl((int)src, l), d0 = (uint32_t)__shfl((int)dst, l);
        for (uint32_t k = lane; k < c; k += 64u) { outPos[d0 + k] = pos[s0 + k]; outMeta[d0 + k] = meta[s0 + k]; }
    }
}

void s3i::launchIota(uint32_t *p, uint32_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(iotaKernel, dim3((n + 255) / 256), dim3(256), 0, s, p, n);
}
void s3i::launchStateAll(uint32_t n, const uint32_t *totalOcc, uint8_t *state, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(stateAllKernel, dim3((n + 255) / 256), dim3(256), 0, s, n, totalOcc, state);
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
    uint32_t *bigList, *bigCount; // slots whose list is longer than S3_VALIDATE_OWN: left to validateBigKernel, one wave per list
};

// The filter restated so that a list can be shared by a wave.  The reference's loop keeps, in list order, the hits with seed mismatches >= minSeedMis and
// seed + extension mismatches <= maxMis -- in the best modes only those with the FEWEST total (a smaller total empties what was kept so far), and it
// stops once the smallest possible total (minSeedMis) has maxHit entries; the list is cut at maxHit afterwards.  Whatever the order of events, what
// remains is: the qualifying hits (best modes: those whose total equals the list's minimum), in list order, the first maxHit of them.  So: one sweep
// that counts every hit's mismatches (kept in bits 16..23 of its meta word) and takes the minimum, one sweep that compacts in order.
// The extension is compared sixteen characters at a time: the read's tail from its packed words (LSB-first characters), the text window from two packed
// words (MSB-first characters) -- read as a number that window IS the reversed sequence, which is what the reverse strand compares against (complemented);
// for the forward strand its character order is turned around (bit reversal, then the two bits of each character swapped back).
struct TailCompare {
    const uint32_t *rd;           // the read's first packed word (32-interleaved layout)
    const uint32_t *text;
    uint32_t wpq, textWords, textLength, readLen, seedLen, ext;
    int maxMis;
    __device__ __forceinline__ uint32_t readWord(uint32_t q0) const       // characters q0 .. q0+15 of the read, LSB-first
    {
        const uint32_t w = q0 >> 4, sh = (q0 & 15u) * 2u;
        const uint32_t lo = w < wpq ? rd[(size_t)w * 32u] : 0u, hi = w + 1u < wpq ? rd[(size_t)(w + 1u) * 32u] : 0u;
        return sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
    }
    __device__ __forceinline__ uint32_t textWord(long long s) const       // characters s .. s+15 of the text, MSB-first (positions outside the text read as 0)
    {
        const long long w = s >> 4;                                        // (arithmetic shift: s may be negative for the last, partial window of a reverse hit)
        const uint32_t sh = (uint32_t)(s & 15) * 2u;
        const uint32_t hi = (w >= 0 && w < (long long)textWords) ? text[w] : 0u, lo = (w + 1 >= 0 && w + 1 < (long long)textWords) ? text[w + 1] : 0u;
        return sh ? (hi << sh) | (lo >> (32u - sh)) : hi;
    }
    // seed + extension mismatches of one seed hit, 0xFF when it does not qualify
    __device__ __forceinline__ uint32_t total(uint32_t pos, uint32_t meta, int minSeedMis) const
    {
        const uint32_t strand = meta & 0xFFu;
        const int seedMis = (int)((meta >> 8) & 0xFFu);
        if (seedMis < minSeedMis) return 0xFFu;
        const bool fwd = strand == 1u && (uint64_t)pos + readLen <= textLength, rev = strand == 2u && pos >= ext;
        if (!fwd && !rev) return 0xFFu;
        int mis = seedMis;
        for (uint32_t k = 0; k < ext && mis <= maxMis; k += 16u) {
            const uint32_t left = ext - k, mask = left >= 16u ? 0x55555555u : ((1u << (2u * left)) - 1u) & 0x55555555u;
            const uint32_t r = readWord(seedLen + k);
            uint32_t x;
            if (fwd) {
                uint32_t t = __brev(textWord((long long)pos + seedLen + k));
                t = ((t >> 1) & 0x55555555u) | ((t & 0x55555555u) << 1);
                x = r ^ t;
            } else {
                x = r ^ ~textWord((long long)pos - 16 - (long long)k);
            }
            mis += __popc((x | (x >> 1)) & mask);
        }
        return mis <= maxMis ? (uint32_t)mis : 0xFFu;
    }
};

#define S3_VALIDATE_OWN 8u          // lists up to this long stay with their read's lane; longer ones are taken by a whole wave each (validateBigKernel)

__global__ void validateKernel(ValidateArgs a)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = slot < a.nActive && a.state[slot] == a.want;
    uint32_t r = 0, readLen = 0, seedLen = 0, base = 0, cnt = 0;
    if (mine) {
        r = a.active ? a.active[slot] : slot;
        readLen = a.lens[r]; seedLen = a.seedLens[r];
        base = a.occBase + a.occOffsets[slot];
        cnt = a.occOffsets[slot + 1] - a.occOffsets[slot];
    }
    const bool extend = mine && readLen > seedLen;
    uint32_t out = cnt;
    auto compareOf = [&](uint32_t rr, uint32_t rl, uint32_t sl) {
        TailCompare c;
        c.rd = a.queries + (size_t)(rr >> 5) * 32u * a.wpq + (rr & 31u);
        c.text = a.text; c.wpq = a.wpq; c.textLength = a.textLength; c.textWords = (a.textLength + 15u) / 16u;
        c.readLen = rl; c.seedLen = sl; c.ext = rl - sl;
        int m = (int)ceil(0.02 * (double)rl);                            // MISMATCH_RATIO_FOR_LONG, CPUfunctions.cpp:1819
        if (a.needMapq) m *= 2;
        c.maxMis = m;
        return c;
    };
    if (extend && cnt <= S3_VALIDATE_OWN) {
        const TailCompare c = compareOf(r, readLen, seedLen);
        uint32_t tot[S3_VALIDATE_OWN], best = 0xFFu;
        #pragma unroll
        for (uint32_t i = 0; i < S3_VALIDATE_OWN; i++) {
            tot[i] = 0xFFu;
            if (i < cnt) { tot[i] = c.total(a.occPos[base + i], a.occMeta[base + i], a.minSeedMis); best = tot[i] < best ? tot[i] : best; }
        }
        out = 0;
        #pragma unroll
        for (uint32_t i = 0; i < S3_VALIDATE_OWN; i++) {
            if (i >= cnt || tot[i] == 0xFFu || (a.onlyBest && tot[i] != best) || out >= a.maxHit) continue;
            const uint32_t pos = a.occPos[base + i], strand = a.occMeta[base + i] & 0xFFu;
            a.occPos[base + out] = strand == 1u ? pos : pos - c.ext;       // (out <= i: entry i has been read by now)
            a.occMeta[base + out] = strand | tot[i] << 8;
            out++;
        }
    }
    if (extend && cnt > S3_VALIDATE_OWN) { a.bigList[atomicAdd(a.bigCount, 1u)] = slot; return; }
    if (!mine) return;
    if (out > a.maxHit) out = a.maxHit;                                  // origHitLimit, CPUfunctions.cpp:1838
    a.outOccCount[r] = out;
    if (out == 0) {
        a.outPhase[r] = 0;
        if (a.carryOn) a.selFlags[slot] = 1;
    }
}

// the long lists, one wave per list (a wave that kept the long lists of its own 64 reads left the launch waiting for the waves with several
// thousand-hit reads among theirs: 27 ms per million reads on the repeat-rich genome)
__global__ void validateBigKernel(ValidateArgs a)
{
    const uint32_t lane = threadIdx.x & 63u, nBig = *a.bigCount;
    const uint32_t nWaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nBig; w += nWaves) {
        const uint32_t slot = a.bigList[w];
        const uint32_t r = a.active ? a.active[slot] : slot;
        const uint32_t b0 = a.occBase + a.occOffsets[slot], c0 = a.occOffsets[slot + 1] - a.occOffsets[slot];
        TailCompare c;
        c.rd = a.queries + (size_t)(r >> 5) * 32u * a.wpq + (r & 31u);
        c.text = a.text; c.wpq = a.wpq; c.textLength = a.textLength; c.textWords = (a.textLength + 15u) / 16u;
        c.readLen = a.lens[r]; c.seedLen = a.seedLens[r]; c.ext = c.readLen - c.seedLen;
        c.maxMis = (int)ceil(0.02 * (double)c.readLen);                  // MISMATCH_RATIO_FOR_LONG, CPUfunctions.cpp:1819
        if (a.needMapq) c.maxMis *= 2;
        // All-valid / all-best reports keep every qualifying hit: one sweep, left as soon as the hit limit is reached -- a read inside a satellite array
        // comes with millions of seed hits (the search runs without a limit in this mode) of which the first thousand that qualify are the answer.
        // Best modes need the list's minimum first; that sweep ends early too once the smallest possible total (minSeedMis) has filled the limit.
        uint32_t best = 0xFFu, kept = 0;
        if (a.onlyBest) {
            uint32_t atFloor = 0;
            for (uint32_t k0 = 0; k0 < c0 && atFloor < a.maxHit; k0 += 64u) {
                const uint32_t k = k0 + lane;
                uint32_t t = 0xFFu;
                if (k < c0) {
                    const uint32_t meta = a.occMeta[b0 + k];
                    t = c.total(a.occPos[b0 + k], meta, a.minSeedMis);
                    a.occMeta[b0 + k] = (meta & 0xFFFFu) | t << 16;
                }
                best = t < best ? t : best;
                atFloor += (uint32_t)__popcll(__ballot((int)t == a.minSeedMis));
            }
            for (int d = 32; d; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)best, d); best = o < best ? o : best; }
        }
        // (best modes: entries behind the point where the first sweep stopped carry no total -- by then `kept` has reached the limit, see above)
        for (uint32_t k0 = 0; k0 < c0 && kept < a.maxHit; k0 += 64u) {
            const uint32_t k = k0 + lane;
            uint32_t pos = 0, meta = 0xFF0000u, t = 0xFFu;
            if (k < c0) {
                pos = a.occPos[b0 + k]; meta = a.occMeta[b0 + k];
                t = a.onlyBest ? meta >> 16 : c.total(pos, meta, a.minSeedMis);
            }
            const bool keep = t != 0xFFu && (!a.onlyBest || t == best);
            const unsigned long long m = __ballot(keep);
            const uint32_t at = kept + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            // (a kept entry lands at or before its own place, and every lane of the wave has read this round's entries before any of them is written)
            if (keep && at < a.maxHit) {
                const uint32_t strand = meta & 0xFFu;
                a.occPos[b0 + at] = strand == 1u ? pos : pos - c.ext;
                a.occMeta[b0 + at] = strand | t << 8;
            }
            kept += (uint32_t)__popcll(m);
        }
        if (lane == 0) {
            if (kept > a.maxHit) kept = a.maxHit;                        // origHitLimit, CPUfunctions.cpp:1838
            a.outOccCount[r] = kept;
            if (kept == 0) {
                a.outPhase[r] = 0;
                if (a.carryOn) a.selFlags[slot] = 1;
            }
        }
    }
}

// ---- long-read mode inside repeats: the decode in slices (round 5) ----------------------------------------------------------------------------------
// The seed search of long-read mode runs without a hit limit (CPUfunctions.cpp:1655-1656): a read inside a satellite array decodes hundreds of thousands of
// seed hits of which the verification keeps MaxOutputPerRead.  Decoding a whole pass at once needed 16 bytes per DECODED hit -- 1 Mi such reads beside a
// 120 GB index ran out of HBM.  Above S3DP_LONG_BUDGET decoded hits (default 2^28) a pass is worked off in slices of slots: a slice's hits are decoded into
// scratch lists, verified there, and only what is kept moves behind the batch's final lists.
__global__ void sliceMaskKernel(uint32_t n, uint32_t s0, uint32_t s1, const uint8_t *state, uint8_t want, const uint32_t *occCounts, uint32_t *masked, uint8_t *sliceState)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { masked[i] = 0; return; }
    const bool in = i >= s0 && i < s1 && state[i] == want;
    masked[i] = in ? occCounts[i] : 0u;
    sliceState[i] = in ? want : (uint8_t)0xFF;
}

__global__ void sliceKeptKernel(uint32_t n, const uint32_t *active, const uint8_t *sliceState, uint8_t want, const uint32_t *outOccCount, uint32_t *kept)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    kept[i] = (i < n && sliceState[i] == want) ? outOccCount[active ? active[i] : i] : 0u;
}

__global__ void sliceStartsKernel(uint32_t n, const uint32_t *active, const uint8_t *sliceState, uint8_t want, const uint32_t *keptOffsets, uint32_t finalBase, uint32_t *outOccStart)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && sliceState[i] == want) outOccStart[active ? active[i] : i] = finalBase + keptOffsets[i];
}

// decode + publish the slots whose state == want; the occurrence region is appended at *dOccBase

// (called by decodeAndPublish behind its mask kernel -- which has set the selection flags -- and its count of the pass's hits)
static int decodeSliced(s3_batch *b, const uint32_t *active, uint32_t nActive, uint8_t want, uint8_t phase, uint32_t maxOut, const LongArgs &lg)
{
    const size_t budget = envSize("S3DP_LONG_BUDGET", (size_t)1 << 28);
    std::vector<uint32_t> counts(nActive);
    std::vector<uint8_t> state(nActive);
    HIPCHK(d2h(counts.data(), b->dOccCounts, (size_t)nActive * 4, b->stream));
    HIPCHK(d2h(state.data(), b->dState, nActive, b->stream));
    // slices of slots whose decoded hits fit the budget (a read with more than that by itself is a slice of its own)
    std::vector<std::pair<uint32_t, uint32_t>> slices;
    uint64_t biggest = 0;
    for (uint32_t s0 = 0; s0 < nActive;) {
        uint64_t sum = 0;
        uint32_t s1 = s0;
        while (s1 < nActive) {
            const uint64_t c = state[s1] == want ? counts[s1] : 0u;
            if (s1 > s0 && sum + c > budget) break;
            sum += c; s1++;
        }
        if (sum >= kMaxEntries) return fail(S3_EOVERFLOW, "a single read with more than 2^32 seed hits (long-read mode)");
        if (sum) slices.push_back({s0, s1});
        biggest = std::max(biggest, sum);
        s0 = s1;
    }
    hipStream_t st = b->stream;
    uint32_t *sPos = nullptr, *sMeta = nullptr, *dKeptCnt = nullptr, *dKeptOff = nullptr, *dBig = nullptr;
    uint8_t *dSlice = nullptr;
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *p : tmp) scratchFree(p, st); };
    auto talloc = [&](void **p, size_t bytes) -> int {
        const hipError_t e = scratchAlloc(p, bytes ? bytes : 16, st);
        if (e != hipSuccess) { cleanup(); return fail(S3_ENOMEM, std::string("hipMalloc(long-read slices): ") + hipGetErrorString(e)); }
        tmp.push_back(*p);
        return S3_OK;
    };
    int rc;
    if ((rc = talloc((void **)&sPos, (size_t)biggest * 4)) || (rc = talloc((void **)&sMeta, (size_t)biggest * 4)) || (rc = talloc((void **)&dKeptCnt, ((size_t)nActive + 1) * 4)) ||
        (rc = talloc((void **)&dKeptOff, ((size_t)nActive + 1) * 4)) || (rc = talloc((void **)&dBig, ((size_t)nActive + 4) * 4)) || (rc = talloc((void **)&dSlice, (size_t)nActive + 16))) return rc;
#define LCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(S3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    const unsigned g = (nActive + 1 + 255) / 256;
    // phase and mismatch tallies of the resolved reads (the list starts and counts it writes are replaced slice by slice below)
    hipLaunchKernelGGL(publishKernel, dim3((nActive + 255) / 256), dim3(256), 0, st, nActive, active, b->dState, want, phase, b->dWithError, b->dOccOffsets, 0u, b->dPhase,
                       b->dFinalWithError, b->dOccStart, b->dOccCount);
    for (const auto &sl : slices) {
        hipLaunchKernelGGL(sliceMaskKernel, dim3(g), dim3(256), 0, st, nActive, sl.first, sl.second, b->dState, want, b->dOccCounts, b->dMasked, dSlice);
        size_t tb = b->tempBytes;
        LCHK(hipcub::DeviceScan::ExclusiveSum(b->dTemp, tb, b->dMasked, b->dOccOffsets, (int)(nActive + 1), st));
        MergeArgs m{};
        m.nActive = nActive; m.nCases = b->lastNumCases; m.maxOut = maxOut; m.itemOffsets = b->dItemOffsets; m.ordIdx = b->dOrdIdx; m.recs = b->dRecs; m.valid = b->dOrdValid;
        m.keptRecs = b->dKept; m.occCounts = b->dOccCounts; m.ceCounts = b->dCeCounts; m.totalOcc = b->dTotalOcc; m.withError = b->dWithError;
        m.decodeMask = dSlice; m.decodeWant = want;
        m.longList = b->dLongList; m.longCount = b->dLongCount;
        m.heavyFlags = b->mergeUnitsPerRead ? b->dHeavyFlags : nullptr; m.heavyIndex = b->dHeavyIndex; m.unitBase = b->mergeUnitBase; m.unitsPerRead = b->mergeUnitsPerRead;
        launchDecode(m, b->ix->d, b->dOccOffsets, 0u, (uint32_t)biggest, sPos, sMeta, b->dFlags, st);
        ValidateArgs v{};
        v.nActive = nActive; v.active = active; v.state = dSlice; v.want = want; v.occOffsets = b->dOccOffsets; v.occBase = 0;
        v.occPos = sPos; v.occMeta = sMeta; v.queries = b->dQueries; v.wpq = b->wpq; v.lens = b->dLens; v.seedLens = b->dSeedLens;
        v.text = b->ix->d.pac; v.textLength = b->ix->d.n;
        v.onlyBest = lg.onlyBest; v.minSeedMis = lg.minSeedMis; v.needMapq = lg.needMapq; v.maxHit = lg.maxHit; v.carryOn = lg.carryOn;
        v.outPhase = b->dPhase; v.outOccCount = b->dOccCount; v.selFlags = b->dSelFlags;
        v.bigList = dBig + 4; v.bigCount = dBig;                              // (a list of its own: the merge kernels' list of long reads is read again by the next slice's decode)
        LCHK(hipMemsetAsync(dBig, 0, 4, st));
        hipLaunchKernelGGL(validateKernel, dim3((nActive + 255) / 256), dim3(256), 0, st, v);
        const unsigned bigBlocks = nActive / 4 + 1 < 4096u ? nActive / 4 + 1 : 4096u;
        hipLaunchKernelGGL(validateBigKernel, dim3(bigBlocks), dim3(256), 0, st, v);
        // what the verification kept moves behind the batch's final lists
        hipLaunchKernelGGL(sliceKeptKernel, dim3(g), dim3(256), 0, st, nActive, active, dSlice, want, b->dOccCount, dKeptCnt);
        tb = b->tempBytes;
        LCHK(hipcub::DeviceScan::ExclusiveSum(b->dTemp, tb, dKeptCnt, dKeptOff, (int)(nActive + 1), st));
        uint32_t kept = 0;
        LCHK(hipMemcpyAsync(&kept, dKeptOff + nActive, 4, hipMemcpyDeviceToHost, st));
        LCHK(hipStreamSynchronize(st));
        if ((uint64_t)b->occUsed + kept >= kMaxEntries) { cleanup(); return fail(S3_EOVERFLOW, "more than 2^32 occurrences in one batch: use smaller batches or a smaller hit limit"); }
        if ((uint64_t)b->occUsed + kept > b->occCap && (rc = growOccurrences(b, (uint64_t)b->occUsed + kept))) { cleanup(); return rc; }
        if (kept)
            hipLaunchKernelGGL(gatherOccKernel, dim3((nActive + 255) / 256), dim3(256), 0, st, nActive, b->dOccOffsets, dKeptOff, sPos, sMeta, b->dOccPos + b->occUsed,
                               b->dOccMeta + b->occUsed);
        hipLaunchKernelGGL(sliceStartsKernel, dim3((nActive + 255) / 256), dim3(256), 0, st, nActive, active, dSlice, want, dKeptOff, b->occUsed, b->dOccStart);
        b->occUsed += kept;
    }
#undef LCHK
    const hipError_t e = hipStreamSynchronize(st);
    cleanup();
    if (e != hipSuccess) return fail(S3_EHIP, std::string("long-read slices: ") + hipGetErrorString(e));
    return S3_OK;
}

int s3i::decodeAndPublish(s3_batch *b, const uint32_t *active, uint32_t nActive, uint8_t want, uint8_t phase, uint32_t maxOut, uint8_t selectWhen,
                            const LongArgs &lg)
{
    if (!nActive) return S3_OK;
    const unsigned g = (nActive + 1 + 255) / 256;
    hipLaunchKernelGGL(maskKernel, dim3(g), dim3(256), 0, b->stream, nActive, b->dState, want, b->dOccCounts, b->dMasked, b->dSelFlags, selectWhen);
    size_t tb = b->tempBytes;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(b->dTemp, tb, b->dMasked, b->dOccOffsets, (int)(nActive + 1), b->stream));
    {
        int rc = sum64Async(b, b->dMasked, nActive);
        if (rc) return rc;
    }
    uint32_t added = 0;
    unsigned long long added64 = 0;
    HIPCHK(hipMemcpyAsync(&added, b->dOccOffsets + nActive, 4, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(&added64, b->dSum64, 8, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (lg.on && added64 > envSize("S3DP_LONG_BUDGET", (size_t)1 << 28)) return decodeSliced(b, active, nActive, want, phase, maxOut, lg);
    if (added64 != added || added64 + b->occUsed >= 0xFFFFFF00ull)
        return fail(S3_EOVERFLOW, "more than 2^32 occurrences in one batch (" + std::to_string(added64 + b->occUsed) + "): use smaller batches or a smaller hit limit");
    if ((uint64_t)b->occUsed + added > b->occCap) {
        int rc = growOccurrences(b, (uint64_t)b->occUsed + added);
        if (rc) return rc;
    }
    const uint32_t base = b->occUsed;
    hipLaunchKernelGGL(publishKernel, dim3((nActive + 255) / 256), dim3(256), 0, b->stream, nActive, active, b->dState, want, phase, b->dWithError,
                       b->dOccOffsets, base, b->dPhase, b->dFinalWithError, b->dOccStart, b->dOccCount);
    MergeArgs m{};
    m.nActive = nActive; m.nCases = b->lastNumCases; m.maxOut = maxOut; m.itemOffsets = b->dItemOffsets; m.ordIdx = b->dOrdIdx; m.recs = b->dRecs; m.valid = b->dOrdValid;
    m.keptRecs = b->dKept; m.occCounts = b->dOccCounts; m.ceCounts = b->dCeCounts; m.totalOcc = b->dTotalOcc; m.withError = b->dWithError;
    m.decodeMask = b->dState; m.decodeWant = want;
    m.longList = b->dLongList; m.longCount = b->dLongCount;                // the list mergeCountKernel of this search pass wrote
    m.heavyFlags = b->mergeUnitsPerRead ? b->dHeavyFlags : nullptr; m.heavyIndex = b->dHeavyIndex; m.unitBase = b->mergeUnitBase; m.unitsPerRead = b->mergeUnitsPerRead;
    launchDecode(m, b->ix->d, b->dOccOffsets, base, b->occCap, b->dOccPos, b->dOccMeta, b->dFlags, b->stream);
    if (lg.on) {
        ValidateArgs v{};
        v.nActive = nActive; v.active = active; v.state = b->dState; v.want = want; v.occOffsets = b->dOccOffsets; v.occBase = base;
        v.occPos = b->dOccPos; v.occMeta = b->dOccMeta; v.queries = b->dQueries; v.wpq = b->wpq; v.lens = b->dLens; v.seedLens = b->dSeedLens;
        v.text = b->ix->d.pac; v.textLength = b->ix->d.n;
        v.onlyBest = lg.onlyBest; v.minSeedMis = lg.minSeedMis; v.needMapq = lg.needMapq; v.maxHit = lg.maxHit; v.carryOn = lg.carryOn;
        v.outPhase = b->dPhase; v.outOccCount = b->dOccCount; v.selFlags = b->dSelFlags;
        v.bigList = b->dLongList; v.bigCount = b->dLongCount;                // (the merge kernels' list of long reads: read for the last time by the decode above)
        HIPCHK(hipMemsetAsync(b->dLongCount, 0, 4, b->stream));
        hipLaunchKernelGGL(validateKernel, dim3((nActive + 255) / 256), dim3(256), 0, b->stream, v);
        const unsigned bigBlocks = nActive / 4 + 1 < 4096u ? nActive / 4 + 1 : 4096u;                    // four lists per block at a time
        hipLaunchKernelGGL(validateBigKernel, dim3(bigBlocks), dim3(256), 0, b->stream, v);
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
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    if (numMismatch > S3_MAX_MISMATCH) return fail(S3_EINVAL, "numMismatch must be 0..4");
    if (maxOutputPerRead == 0 || maxOutputPerRead > 0xFFFFFFu) return fail(S3_EINVAL, "maxOutputPerRead must be 1..16777215");
    HIPCHK(hipSetDevice(b->ix->dev));
    const uint32_t n = b->n;
    b->searchLaunches = 0; b->evSearchUsed = 0; b->evCeUsed = 0;
    b->occUsed = 0;
    b->recordsTotal = 0;
    HIPCHK(hipMemsetAsync(b->dFlags, 0, 4, b->stream));
    HIPCHK(hipMemsetAsync(b->dWasHeavy, 0, b->maxReads, b->stream));                  // the ladder's memory of the reads its first passes listed
    b->unitGroup = S3_UNIT_GROUP;                                                       // units of the second pass: groups of two positions (searchPass)
    if (b->counting) HIPCHK(hipMemsetAsync(b->dCounters, 0, S3_COUNTER_WORDS * 8, b->stream));
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
    // (Starting the ladder from the reads in the order of their first 32 characters -- family-mates of a repeat as neighbours in the first pass's waves -- was
    //  measured in round 5: a fifth fewer bytes fetched by the first passes, the same time.  Taken out in round 6; the LISTED reads are still ordered that way.)
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
            launchIota(b->dIota, nActive, b->stream);
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
                launchStateAll(nNext, b->dTotalOcc, b->dState, b->stream);
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
    {
        uint32_t kept = 0;
        HIPCHK(hipMemcpyAsync(&kept, b->dOutOffsets + n, 4, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        int rc = ensureOutLists(b, kept);
        if (rc) return rc;
    }
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
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    int rc = s3_batch_sync(b);
    if (rc) return rc;
    const uint32_t n = b->n;
    std::vector<uint32_t> off((size_t)n + 1);
    HIPCHK(d2h(off.data(), b->dOutOffsets, ((size_t)n + 1) * 4, b->stream));
    const uint64_t total = off[n];
    if (nOccOut) *nOccOut = total;
    if (phase) HIPCHK(d2h(phase, b->dPhase, n, b->stream));
    if (withError) HIPCHK(d2h(withError, b->dFinalWithError, (size_t)n * 32, b->stream));
    if (occOffsets) for (size_t i = 0; i <= n; i++) occOffsets[i] = off[i];
    if (occPos || occStrand || occMis) {
        if (total > occCap) return fail(S3_EOVERFLOW, "s3_batch_fetch_alignments: output arrays too small");
        if (occPos) HIPCHK(d2h(occPos, b->dOutPos, total * 4, b->stream));
        if (occStrand || occMis) {
            std::vector<uint32_t> meta(total);
            HIPCHK(d2h(meta.data(), b->dOutMeta, total * 4, b->stream));
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
    BA(dPairHeavyList, maxPairs); BA(dPairHeavyCount, 4); BA(dBestRank, maxPairs); BA(dBestLocal, maxPairs);
    b->xaCap = (uint32_t)(maxPairs + 65536);
    BA(dXa, b->xaCap); BA(dXaOut, b->xaCap);
#undef BA
    b->pairReady = true;
    return S3_OK;
}

// The segmented sort of the pairs' occurrence keys (64-bit key, 32-bit value).  rocPRIM's default configuration for gfx950 (no tuned entry) spills: 248 VGPRs
// and 148 bytes of SCRATCH per lane, and the first kernel with scratch on a HIP stream costs a reallocation of the device's scratch buffer with every queue
// drained (s3dp_api.hip, ListedSortConfig).  256 threads x 8 items: 194 VGPRs, no scratch.
using PairSortConfig = rocprim::segmented_radix_sort_config<8, rocprim::kernel_config<256, 8>, rocprim::WarpSortConfig<32, 4, 256, 3000, 32, 4, 256>, true>;
static hipError_t sortSegments(void *temp, size_t &bytes, const uint64_t *keysIn, uint64_t *keysOut, const uint32_t *valsIn, uint32_t *valsOut, uint32_t items,
                               uint32_t segments, const uint32_t *segBegin, const uint32_t *segEnd, unsigned beginBit, unsigned endBit, hipStream_t s)
{
    return rocprim::segmented_radix_sort_pairs<PairSortConfig>(temp, bytes, keysIn, keysOut, valsIn, valsOut, items, segments, segBegin, segEnd, beginBit, endBit, s);
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
    HIPCHK(sortSegments(nullptr, tb, b->dSortKeys[0], b->dSortKeys[1], b->dOccMeta, b->dSortMeta, (uint32_t)items, (uint32_t)segments, b->dSegBegin, b->dSegEnd, 0, 64, b->stream));
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
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
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
    b->searchLaunches = 0; b->evSearchUsed = 0; b->evCeUsed = 0;
    b->occUsed = 0;
    b->xaUsed = 0;
    b->routed = false;
    b->recordsTotal = 0;
    HIPCHK(hipMemsetAsync(b->dFlags, 0, 4, b->stream));
    HIPCHK(hipMemsetAsync(b->dWasHeavy, 0, b->maxReads, b->stream));                  // the ladder's memory of the reads its first passes listed
    b->unitGroup = 1;                                                                   // ... one position per unit for pairs (measured: searchPass)
    if (b->counting) HIPCHK(hipMemsetAsync(b->dCounters, 0, S3_COUNTER_WORDS * 8, b->stream));
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
                launchStateAll(nR, b->dTotalOcc, b->dState, b->stream);
                rc = decodeAndPublish(b, b->dActive[0], nR, 1, 0, maxHitsPerEnd, 0, noLong);
                if (rc) return rc;
            }
        }
        // position-sort both lists of every pair with hits on both ends (PERadixSort), then pair them up
        const size_t items = (size_t)b->occUsed - regionBase;
        PairSortArgs sa{};
        sa.nPairs = nA; sa.activePairs = active; sa.pairNext = b->dPairNext; sa.occStart = b->dOccStart; sa.occCount = b->dOccCount;
        sa.regionBase = regionBase; sa.occPos = b->dOccPos; sa.occMeta = b->dOccMeta; sa.segBegin = b->dSegBegin; sa.segEnd = b->dSegEnd;
        sa.idxBits = 1;
        while ((1ull << sa.idxBits) <= (unsigned long long)maxHitsPerEnd) sa.idxBits++;          // (a list holds at most maxHitsPerEnd <= 2^24 - 1 hits)
        if (items) {
            rc = ensureSortBuffers(b, items, (int)(2 * nA));
            if (rc) return rc;
            sa.keys = b->dSortKeys[0];
            launchPairSortKeys(sa, b->stream);
            size_t tb = b->sortTempBytes;
            HIPCHK(sortSegments(b->dSortTemp, tb, b->dSortKeys[0], b->dSortKeys[1], b->dOccMeta + regionBase, b->dSortMeta, (uint32_t)items, 2u * nA, b->dSegBegin,
                                b->dSegEnd, 0, 32u + sa.idxBits, b->stream));
            launchPairSortBack(sa, b->dSortKeys[1], b->dSortMeta, b->stream);
        }
        launchPairSnapshotCounts(nA, active, b->dOccCount, b->dOccCountIn, b->stream);
        PairSweepArgs sw{};
        sw.nPairs = nA; sw.activePairs = active; sw.pairNext = b->dPairNext; sw.pairMin = b->dPairMin; sw.finalWithError = b->dFinalWithError;
        sw.occStart = b->dOccStart; sw.occCount = b->dOccCount; sw.occCountIn = b->dOccCountIn; sw.occPos = b->dOccPos; sw.occMeta = b->dOccMeta;
        sw.readLengths = b->dLens; sw.insertLow = insertLow; sw.insertHigh = insertHigh; sw.peMaxOutputPerPair = peMaxOutputPerPair; sw.mm = mm;
        sw.terminal = terminal; sw.pairs = b->dPairs; sw.bestIdx = b->dBestIdx; sw.xaCount = b->dXaCount; sw.xaStart = b->dXaStart; sw.xaCountOut = b->dXaCountOut;
        // pairs with more than S3DP_PAIR_HEAVY entries on both ends together: one wave per pair (s3dp_pair.hip); 0xFFFFFFFF = every pair by its own lane
        const uint32_t heavyMin = (uint32_t)envSize("S3DP_PAIR_HEAVY", 48);      // (read per call: tests/test_gpu_parity.py switches it)
        sw.heavyMin = items ? heavyMin : 0xFFFFFFFFu; sw.regionBase = regionBase; sw.heavyList = b->dPairHeavyList; sw.heavyCount = b->dPairHeavyCount;
        sw.bestRank = b->dBestRank; sw.bestLocal = b->dBestLocal; sw.scratchW = (uint32_t *)b->dSortKeys[0]; sw.scratchR = (uint32_t *)b->dSortKeys[1];
        HIPCHK(hipMemsetAsync(b->dPairHeavyCount, 0, 4, b->stream));
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
            launchIota(b->dIota, nA, b->stream);
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
    {
        uint32_t kept = 0;
        HIPCHK(hipMemcpyAsync(&kept, b->dOutOffsets + n, 4, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        if ((rc = ensureOutLists(b, kept))) return rc;
    }
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

// The paired-end result of a ladder that was not run: every pair without a proper pairing, no occurrence on either end.  (Pairs of reads above 150 bases skip
// the BWT stage, SOAP3-DP.cu:658-662, alignment.cu:1928-1944: the DP stages then find every pair routed to the deep DP.)
extern "C" int s3_batch_no_pair_result(s3_batch *b)
{
    if (!b) return fail(S3_EINVAL, "null batch");
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    if (b->n & 1u) return fail(S3_EINVAL, "s3_batch_no_pair_result: the batch must hold an even number of reads");
    HIPCHK(hipSetDevice(b->ix->dev));
    int rc = ensurePairBuffers(b);
    if (rc) return rc;
    const uint32_t n = b->n, nPairs = n / 2;
    b->searchLaunches = 0; b->evSearchUsed = 0; b->evCeUsed = 0; b->occUsed = 0; b->xaUsed = 0; b->recordsTotal = 0; b->routed = false;
    HIPCHK(hipEventRecord(b->evTotal[0], b->stream));
    HIPCHK(hipMemsetAsync(b->dOutOffsets, 0, ((size_t)n + 1) * 4, b->stream));
    HIPCHK(hipMemsetAsync(b->dXaOutOffsets, 0, ((size_t)nPairs + 1) * 4, b->stream));
    if (nPairs) hipLaunchKernelGGL(pairInitKernel, dim3((nPairs + 255) / 256), dim3(256), 0, b->stream, nPairs, b->dPairs);
    HIPCHK(hipEventRecord(b->evTotal[1], b->stream));
    HIPCHK(hipGetLastError());
    b->lastKind = 3;
    return S3_OK;
}

extern "C" int s3_batch_fetch_pairs(s3_batch *b, s3_pair_result *pairs, uint64_t *xaOffsets, s3_pair_xa *xa, size_t xaCap, uint64_t *nXaOut,
                                    uint64_t *occOffsets, uint32_t *occPos, uint8_t *occStrand, uint8_t *occMis, size_t occCap, uint64_t *nOccOut)
{
    if (!b || b->lastKind != 3) return fail(S3_EINVAL, "s3_batch_fetch_pairs: no paired-end result pending");
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    int rc = s3_batch_sync(b);
    if (rc) return rc;
    static_assert(sizeof(s3_pair_result) == sizeof(DevPair) && sizeof(s3_pair_xa) == sizeof(DevPairXa), "pair layouts");
    const uint32_t n = b->n, nPairs = n / 2;
    if (pairs && nPairs) HIPCHK(d2h(pairs, b->dPairs, (size_t)nPairs * sizeof(DevPair), b->stream));
    std::vector<uint32_t> off((size_t)nPairs + 1);
    HIPCHK(d2h(off.data(), b->dXaOutOffsets, ((size_t)nPairs + 1) * 4, b->stream));
    const uint64_t nXa = off[nPairs];
    if (nXaOut) *nXaOut = nXa;
    if (xaOffsets) for (size_t i = 0; i <= nPairs; i++) xaOffsets[i] = off[i];
    if (xa) {
        if (nXa > xaCap) return fail(S3_EOVERFLOW, "s3_batch_fetch_pairs: pair list array too small");
        if (nXa) HIPCHK(d2h(xa, b->dXaOut, nXa * sizeof(DevPairXa), b->stream));
    }
    if (!occOffsets && !occPos && !occStrand && !occMis && !nOccOut) return S3_OK;   // (a caller that routes the unpaired pairs on the device takes no lists)
    std::vector<uint32_t> ooff((size_t)n + 1);
    HIPCHK(d2h(ooff.data(), b->dOutOffsets, ((size_t)n + 1) * 4, b->stream));
    const uint64_t total = ooff[n];
    if (nOccOut) *nOccOut = total;
    if (occOffsets) for (size_t i = 0; i <= n; i++) occOffsets[i] = ooff[i];
    if (occPos || occStrand || occMis) {
        if (total > occCap) return fail(S3_EOVERFLOW, "s3_batch_fetch_pairs: occurrence arrays too small");
        if (occPos && total) HIPCHK(d2h(occPos, b->dOutPos, total * 4, b->stream));
        if ((occStrand || occMis) && total) {
            std::vector<uint32_t> meta(total);
            HIPCHK(d2h(meta.data(), b->dOutMeta, total * 4, b->stream));
            for (uint64_t i = 0; i < total; i++) {
                if (occStrand) occStrand[i] = (uint8_t)(meta[i] & 0xFF);
                if (occMis) occMis[i] = (uint8_t)((meta[i] >> 8) & 0xFF);
            }
        }
    }
    return S3_OK;
}

extern "C" int s3_batch_fetch_pair_origins(s3_batch *b, uint8_t *fromSaRange, size_t cap)
{
    if (!b || b->lastKind != 3 || !fromSaRange) return fail(S3_EINVAL, "s3_batch_fetch_pair_origins: no paired-end result pending");
    if (b->failed) return fail(S3_ENOMEM, "this batch lost a buffer while growing (out of device memory earlier): free it and create a smaller one");
    int rc = s3_batch_sync(b);
    if (rc) return rc;
    uint32_t total = 0;
    HIPCHK(d2h(&total, b->dOutOffsets + b->n, 4, b->stream));
    if (total > cap) return fail(S3_EOVERFLOW, "s3_batch_fetch_pair_origins: array too small");
    std::vector<uint32_t> meta(total ? total : 1);
    if (total) HIPCHK(d2h(meta.data(), b->dOutMeta, (size_t)total * 4, b->stream));
    for (uint32_t i = 0; i < total; i++) fromSaRange[i] = (uint8_t)((meta[i] >> 16) & 1u);
    return S3_OK;
}

// one kernel of this file's code object, asked for its attributes: the runtime loads a code object at the first use of one of its kernels (preloadKernels, s3dp_kernels.hip)
void s3::preloadLadder() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, (const void *)iotaKernel); (void)hipGetLastError(); }
