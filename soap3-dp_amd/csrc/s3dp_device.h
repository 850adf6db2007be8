// s3dp_device.h -- structures shared between the kernels (s3dp_kernels.hip) and the C-ABI layer (s3dp_api.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "s3dp_model.h"

#define S3_SEARCH_BLOCK 256
#define S3_V2_COUNT_BATCH 32u             // second pass: accepted candidates a unit books at a time
#define S3_V2_REC_BLOCK 256u              // record slots a wave reserves with one atomic
#define S3_V2_MODEL_BYTES 1104u           // sizeof(S3Model) rounded up to 16
#define S3_FIRST_WAVES_PER_SIMD 4         // register budget of the first-pass kernel (s3dp_first.hip)
#define S3_FIRST_RDY_BYTES 512u            // LDS per wave of the first pass for the reads waiting for a lane (2 words x 64)
#define S3_UNIT_WAVES_PER_SIMD 4          // register budget of the unit-pass kernel (s3dp_unit.hip)
#define S3_UNIT_CHUNK 64u                 // units a wave of the unit pass takes from the work queue at a time (one per lane: tested before they start)
#define S3_UNIT_STACK 32u                 // depth-first frames per lane of the unit pass (a ring in global memory; a power of two)
#define S3_UNIT_RDY_BYTES 1536u            // LDS per wave of the unit pass for its tested units (6 words x 64)
#define S3_UNIT_ADOPT_IDLE 32u             // idle lanes of a wave before frames are adopted
#define S3_UNIT_TAKE_MIN 8u                // idle lanes of a wave before they take their next units (S3DP_TAKE_MIN)
#define S3_UNIT_FAST_TRIPS 7u              // step-only trips of a wave of the unit pass between two full trips, at most (S3DP_FAST_TRIPS; 0: none)
#define S3_UNIT_FAST_MIN 32u               // ... made while this many of its lanes are stepping (S3DP_FAST_MIN)
#define S3_FIRST_FAST_TRIPS 3u             // the same for the first pass and the trunk pass (S3DP_FIRST_FAST_TRIPS / S3DP_FIRST_FAST_MIN)
#define S3_FIRST_FAST_MIN 40u
#define S3_UNIT_GIANT_AGE 48u              // full trips on one unit that make a lane a giant whose frames its wave adopts before the queue is empty (S3DP_GIANT_AGE; 0: never)
#define S3_UNIT_GIANT_IDLE 8u
#define S3_UNIT_TAIL_AGE 0u
#define S3_UNIT_GROUP_HI 8u                // ... from three mismatches on (S3DP_UNIT_GROUP_HI)
#define S3_UNIT_GROUP 2u                   // first-mismatch positions per unit of the second pass (S3DP_UNIT_GROUP)
#define S3_UNIT_NODE_BLOCK 256u           // sub-unit counters a wave of the unit pass reserves with one atomic
#define S3_NUM_COUNTERS 40                 // 64-bit tallies of a kernel (SearchOut::counters); [21]: rank queries the first pass spent on reads it then listed; [22..39]: ceKernel's rejection histogram (s3dp_ce.hip)
#define S3_CE_HIST_BASE 22
#define S3_CE_HIST_SLOTS 18
#define S3_NUM_ROLES 4                     // ... kept per kernel of a search pass: first pass, trunk pass, unit pass, check-and-extend (s3_batch_kernel_stats)
#define S3_COUNTER_WORDS (S3_NUM_COUNTERS * S3_NUM_ROLES)
#define S3_FLAG_STACK_OVERFLOW 1u
#define S3_FLAG_REC_OVERFLOW   2u
#define S3_FLAG_OCC_OVERFLOW   4u
#define S3_FLAG_FLANK          8u     // ceKernel: a survivor of a forward task is not in its task's forward range (the flank table of the reverse index and the index disagree)

namespace s3 {

struct DevBwt {
    const uint4 *lines;          // 64-byte lines, 4 x uint4 per 128 BWT characters (see s3dp_kernels.hip)
    uint32_t     inverseSa0;
    uint32_t     n;
};

struct DevIndex {
    DevBwt          fwd, rev;
    const uint32_t *sa;
    const uint32_t *pac;
    size_t          pacWords;
    const uint32_t *lkt, *rlkt;
    int             ltSize;
    uint32_t        n;
    // long lookup tables held in HBM (built at upload, s3dp_kernels.hip::launchLongLookup): the state of an item after its lookup step's first
    // ltK = ltSize + 1 or 2 characters -- what the ltSize-mer table plus one / two exact steps give, stored so that the steps are not walked again
    // FLANK TABLES (round 6; s3dp_flank.hip): per suffix-array entry its text position and the 48 characters NEXT to it, the adjacent one first, 16 per word MSB
    // first -- flankBack[k]: {SA[k], T[SA[k]-1], T[SA[k]-2], ...} in the order of the forward index, flankFwd[k']: {E, T[E], T[E+1], ...} in the order of the REVERSE
    // index (E = the text position behind the suffix's first character as the reverse index reads it, i.e. behind the matched part of a forward step).
    // ceKernel rejects nine candidates in ten from this one 16-byte entry (instead of a suffix-array line and then, dependently, a text line).  nullptr: none.
    const uint4    *flankBack, *flankFwd;
    const uint2    *ltBack;      // backward lookup steps: {l, r}, indexed by the ltK characters in read order
    const uint4    *ltBi;        // bidirectional lookup steps: {l, r, rev l, rev r}, indexed by the ltK characters REVERSED (order of the reverse index)
    int             ltK;         // 0: none
    // The 13-mers that the file format's tables count for text positions closer than 13 characters to an end (padded with 'A', LTConstruct.c:45-101):
    // a lookup range for one of them may hold an element that is no occurrence.  padBack: as a backward item builds its key (read order);
    // padBi: as a bidirectional item builds its reversed key -- the reverse table's own ones and the forward table's, pairs reversed.
    uint32_t        padBack[12], padBi[24];
};

struct SearchIn {
    const uint32_t *queries;     // packed reads, reference layout
    const uint32_t *readsT;      // both orientations, MSB-first words: readsT[r * 2 * wpqLoad + strand * wpqLoad + w]  (v2 kernel)
    const uint32_t *readLengths;
    const uint32_t *active;      // optional indirection: slot -> read index
    uint32_t        nActive;
    uint32_t        wpq;         // words per read in `queries` / `readsT` (stride)
    uint32_t        wpqLoad;     // words per read the search needs (covers the longest searched length; power of two, <= wpq)
    uint32_t        wLds;        // of those, the words per orientation the v2 kernel keeps in LDS (ceil(longest searched length / 16), <= wpqLoad)
    const S3Model  *models;      // indexed by read length, for this phase's mismatch budget
    int             nCases;
    uint32_t        maxOut;
    uint32_t        nStepsPacked;  // 3 bits per case: number of steps (depends on the mismatch budget only)
    uint32_t        uniformLen;  // != 0: every active read has this length (model kept in LDS by the v2 kernel)
    uint32_t        earlyMinI;   // ... but only once this many characters are matched: a chance match of the lookup characters costs a step (one line) to drop,
                                 // a candidate three lines to verify
    uint32_t        heavyRank;   // first pass: ... or that has cost more rank queries than this
    uint32_t        heavyLevel;  // first pass: mismatch budget of this pass + 1 (what SearchOut::wasHeavy remembers; a read listed at a level <= this one is listed at once)
    uint32_t        heavyLimit;  // first pass: a read that reserves more check-and-extend candidates than this is listed for the second pass
    const uint32_t *slotList;    // second pass (item mode: one work unit per item of a listed read): the listed slots ...
    const uint32_t *slotCount;   // ... and how many (device memory, written by the first pass)
    uint32_t        unitMode;      // 0: first pass (a unit = a read); 1: trunk pass over the listed reads' items; 2: unit pass (s3dp_first.hip, s3dp_unit.hip)
    uint32_t        unitsPerItem;  // second pass: units an item is cut into -- GROUPS of unitGroup consecutive first-mismatch positions, + 1 for "none" (1: whole items)
    uint32_t        unitGroup;     // second pass: consecutive positions per unit (S3DP_UNIT_GROUP; 1: one unit per position, as until round 5)
    uint32_t        unitBase;      // second pass: unit u counts its records in itemCounts[unitBase + u]
    uint32_t        sStarPacked;   // 3 bits per case: the first step that allows a mismatch (= nSteps: none)
    uint32_t        adoptIdle;   // unit pass: idle lanes a wave needs before its busy lanes give frames away (S3_UNIT_ADOPT_IDLE)
    uint32_t        takeMin;     // unit pass: idle lanes a wave lets gather before they start their next units together (S3_UNIT_TAKE_MIN; 1: at once)
    uint32_t        tailAge;             // unit pass, queue empty: full trips a lane must have spent on its unit before it gives frames away (S3DP_TAIL_AGE)
    uint32_t        giantAge, giantIdle; // unit pass: full trips on one unit that make a lane a giant (0: never), idle lanes of its wave before they adopt its frames (S3DP_GIANT_AGE / _IDLE)
    uint32_t        fastTrips, fastMin;  // unit pass: step-only trips between two full trips (S3_UNIT_FAST_TRIPS), made while fastMin lanes of the wave are stepping
    uint32_t        firstFastTrips, firstFastMin;  // the same for the first pass and the trunk pass
    uint32_t        earlyCe;     // ranges of up to this many suffixes leave the exact lookup step for check-and-extend at once (0: never; s3dp_search_common.h)
};

#define S3_REC_TASK  (1u << 28)
#define S3_REC_PASS2 0x80000000u            // in the item word: record written by the second pass (records of a listed read without it are stale)
#define S3_TRUNK_VALID 0x80000000u          // trunkStates[].w: the trunk pass reached this position
#define S3_REC_PADDED (1u << 24)             // task of an item whose lookup range may hold a padded table element: its forward and reverse ranges need not hold the same occurrences
#define S3_REC_FWD   (1u << 30)              // task written in a FORWARD step: its candidates are looked up in the reverse index's flank table (recAux)
#define S3_REC_EARLY (1u << 29)              // task written inside the exact lookup step (single-suffix range): the rest of that step is verified too

struct SearchOut {
    uint4              *recs;    // {item, seq|kind<<24|nmis<<25, a, b};  S3_REC_TASK set: a check-and-extend task {item, first seq, SA left, size-1 | state<<8}
    uint4              *recAux;  // per record slot, written with a TASK record: {the first suffix of the task's range in the REVERSE index (what a task written in a
                                 // forward step looks its candidates up by in DevIndex::flankFwd; recs[].z stays the forward index's), the 48 characters of the
                                 // read NEXT to the matched part, the adjacent one first, 16 per word MSB first}: ceKernel's filter needs nothing else of the read
    uint32_t            recCap;
    uint32_t           *recCount;
    uint32_t           *itemCounts;
    uint32_t           *flags;
    unsigned long long *counters;   // optional: rank, lt, sa, ce tallies
    unsigned long long *workHead;   // v2 kernel: head of the item queue (zeroed before every launch)
    uint8_t            *heavyFlags; // per slot: listed for the second pass (zeroed before every first pass)
    uint32_t           *heavyList, *heavyCount;
    uint32_t           *heavyIndex; // per listed slot: its place in heavyList
    uint8_t            *wasHeavy;   // per READ of the batch (not slot): SearchIn::heavyLevel of the pass that listed it, kept over the phases of a ladder (nullptr: off)
    uint32_t           *cutDummy;    // a scratch word: where a lane's cut pointer points while it has no listed read
    uint32_t           *readCut;     // second pass, per listed read: the smallest unit (item * units + position) that holds maxOut occurrences by itself
    uint32_t           *itemAcc;     // second pass, per listed read and item: occurrences its units hold for sure (sums over the items give the cut too)
    uint4              *trunkStates; // second pass: {l, r, rev l, S3_TRUNK_VALID | occurrences of the unit} at [(item * units + position) * listed reads + read]
    // unit pass (s3dp_unit.hip)
    const uint32_t     *listedRead;  // per listed read: its index in readsT / readLengths
    uint4              *unitStack;   // depth-first frames: [wave of the grid][S3_UNIT_STACK][64 lanes]
    // A lane with nothing left to take adopts the OLDEST frame of a busy lane of its wave (the subtree that lane would have searched last).  The frame's
    // records count from zero in a counter of their own (a "node": itemCounts[dynBase + k]); chainNext links the counters in sequence order -- a node is
    // inserted right behind the counter of the lane that gave the frame away --, dynOwner names the unit the chain started from, and unitChainKernel
    // turns the chains into offsets (dynLocal) before the counting scatter runs.
    uint32_t           *chainNext;   // per counter (units and nodes): the counter whose records follow (0: none)
    uint32_t           *dynOwner;    // per node: the unit (index into itemCounts) at the head of its chain
    uint32_t           *dynUsed;     // nodes handed out so far (may run past dynCap: the excess was never used)
    uint32_t           *dynLocal;    // per node: where its records start inside the total of its chain (unitChainKernel)
    uint32_t            dynBase, dynCap; // nodes count in itemCounts[dynBase + k], k < dynCap (0: no adoption)
};

struct DevHit { uint32_t a, b, meta; };       // binary-identical to s3_hit (kind, strand, nmis, cas bytes)

struct MergeArgs {
    uint32_t        nActive, nCases, maxOut;
    const uint32_t *itemOffsets;
    // (round 6: the ordered array holds the NUMBER of the record that was scattered to a slot, not a copy of it -- 4 B per sequence number of a pass instead of 16: nine
    //  in ten sequence numbers belong to rejected check-and-extend candidates, and at 16 B each the array was what pushed the flank tables out of HBM on repeat-heavy
    //  batches; a slot is read only where its validity byte says a record sits)
    const uint32_t *ordIdx;
    const uint4    *recs;
    const uint8_t  *valid;        // one byte per slot of `ordered`: 1 = a record was scattered there (the rest: sequence numbers of rejected check-and-extend candidates)
    uint32_t       *keptRecs, *occCounts, *ceCounts, *totalOcc, *withError;
    const uint8_t  *decodeMask;   // optional per-slot state; only slots with state == decodeWant are decoded
    uint8_t         decodeWant;
    // reads whose slice of the ordered array is longer than S3_MERGE_LONG entries (repeat families: mostly holes of rejected candidates) are listed by
    // mergeCountKernel and walked by a whole wave each (mergeCountLongKernel / decodeLongKernel) instead of one lane
    uint32_t       *longList, *longCount;
    // reads the second search pass took apart count their records per unit: their slice of the ordered array starts at
    // itemOffsets[unitBase + heavyIndex[slot] * unitsPerRead] and is unitsPerRead offsets long
    const uint8_t  *heavyFlags;
    const uint32_t *heavyIndex;
    uint32_t        unitBase, unitsPerRead;
};
#define S3_MERGE_LONG 256u

// semi-global DP (s3dp_dp.hip)
// Traceback table of one 64-row chunk with T steps: H as int16, two steps per dword ([step / 2][lane], the even step in the low half), then one bit per
// cell, 32 steps per dword ([step / 32][lane], step t at bit 31 - t % 32): 2.125 B per cell, every store a whole 256-byte row.
static inline __host__ __device__ size_t dpTableXBase(uint32_t T) { return ((size_t)(T >> 1) + 1u) * 64u; }
static inline __host__ __device__ size_t dpTableChunkWords(uint32_t T) { return dpTableXBase(T) + ((size_t)(T >> 5) + 1u) * 64u; }
struct DpArgs {
    const uint32_t *dna, *dnaLengths, *reads, *readLengths;     // reference layouts (1-based, MSB first, 32-interleaved)
    const uint32_t *clipLt, *anchorLeft, *anchorRight;            // optional
    uint32_t       *clipRt;                                       // optional, in/out
    const int32_t  *cutoffs;
    int32_t        *scores;
    uint32_t       *hitLocs, *startOffsets, *maxScoreCounts, *tableCols;
    unsigned char  *pattern;
    uint32_t       *table;                                        // per problem and 64-row chunk: (maxDPTableLength + 64) steps x 64 int16 H, then one 64-bit word per step
    size_t          tableWordsPerProblem;
    uint32_t        n, maxDNALength, maxReadLength, maxDPTableLength, patternLen, ldsPerWave;
    uint32_t        ldsPerWavePair, first;                        // LDS of a wave that sweeps two problems (20 B per column); first problem of a launch of dpFillKernel
    int             pairMode;                                     // 0: one problem per wavefront whatever the scores (S3DP_DP_PAIRED=0)
    int             fuseMode, fuseTraceback;                      // the paired fill kernel goes on to walk its two tables (S3DP_DP_FUSED=0: dpBacktrackKernel as a launch of its own); set per launch
    int             match, mismatch, gapOpen, gapExtend, scheme;
    uint32_t       *patMax;                                       // optional: longest pattern of the launch in bytes (atomicMax; zeroed by the caller)
};
int launchDp(const DpArgs &a, hipStream_t s);        // returns a hipError_t

void launchOrientReads(const uint32_t *queries, const uint32_t *lengths, uint32_t *readsT, uint32_t n, uint32_t wpq, uint32_t S, hipStream_t s);
void launchZeroListedCounts(const uint32_t *heavyList, const uint32_t *heavyCount, uint32_t nHeavy, uint32_t groups, uint32_t *itemCounts, hipStream_t s);
void launchSaExpand(const DevBwt &b, const uint32_t *samples, uint32_t interval, uint32_t n, uint32_t *sa, hipStream_t s);
void launchRelayout(const uint32_t *bwt, size_t bwtWords, const uint32_t *occ, size_t occRecs, uint4 *lines, hipStream_t s);
int launchFirstPass(const DevIndex &ix, const SearchIn &in, const SearchOut &out, unsigned maxBlocks, hipStream_t s);    // s3dp_first.hip; returns a hipError_t
int launchTrunkPass(const DevIndex &ix, const SearchIn &in, const SearchOut &out, unsigned maxBlocks, hipStream_t s);    // s3dp_first.hip, the trunk-pass role
int launchUnitPass(const DevIndex &ix, const SearchIn &in, const SearchOut &out, unsigned maxBlocks, hipStream_t s);     // s3dp_unit.hip; returns a hipError_t
unsigned unitPassMaxWaves(unsigned maxBlocks);                                        // waves of the widest unit-pass grid (sizes SearchOut::unitStack)
void launchListedReads(const uint32_t *heavyList, const uint32_t *heavyCount, const uint32_t *active, uint32_t *listedRead, uint32_t nHeavy, hipStream_t s);
// the order of the listed reads (s3dp_unit.hip, "LOCALITY"): keys = the first 32 characters of read active[slots[k]] (slots == nullptr: k), vals = the slot
void launchListedKeys(const uint32_t *slots, uint32_t n, const uint32_t *active, const uint32_t *readsT, uint32_t stride, uint32_t words, unsigned long long *keys,
                      uint32_t *vals, hipStream_t s);
void launchListedPlace(const uint32_t *sortedSlots, uint32_t n, uint32_t *heavyList, uint32_t *heavyIndex, hipStream_t s);
void launchUnitChains(const SearchOut &out, uint32_t unitBase, uint32_t nUnits, hipStream_t s);
void launchCheckExtend(const DevIndex &ix, const SearchIn &in, const SearchOut &out, const uint32_t *taskRange, hipStream_t s);
int  launchFlankTables(DevIndex &d, int which, size_t reserveBytes, void **allocs, int *nAllocs, size_t *bytes, bool verbose, hipStream_t s);   // s3dp_flank.hip; returns a hipError_t as int
int  launchLongLookup(DevIndex &d, int extra, uint2 **backOut, uint4 **biOut, hipStream_t s);   // allocates the two tables (hipMalloc), returns a hipError_t as int
struct ScatterDyn { const uint32_t *owner, *local; uint32_t base, cap; };          // the counters of adopted frames (SearchOut::dynOwner / dynLocal / dynBase / dynCap)
void launchScatter(uint4 *recs, const uint32_t *recCount, uint32_t recCap, const uint32_t *itemOffsets, uint32_t *ordIdx, uint8_t *valid, const uint8_t *heavyFlags,
                   const uint32_t *heavyList, uint32_t groups, uint32_t unitBase, uint32_t unitsPerItem, ScatterDyn dyn, hipStream_t s);
void launchMergeCount(const MergeArgs &a, hipStream_t s);
void launchMergeFill(const MergeArgs &a, const uint32_t *hitOffsets, DevHit *hits, hipStream_t s);
void launchDecode(const MergeArgs &a, const DevIndex &ix, const uint32_t *occOffsets, uint32_t occBase, uint32_t occCap, uint32_t *occPos,
                  uint32_t *occMeta, uint32_t *flags, hipStream_t s);
void launchClassify(uint32_t nActive, const uint32_t *totalOcc, const uint32_t *withError, uint32_t mm, uint32_t allowSpecial, uint8_t *state,
                    hipStream_t s);

// paired-end bookkeeping (s3dp_pair.hip); DevPair / DevPairXa are the ABI's s3_pair_result / s3_pair_xa
struct DevPair {
    uint8_t  phase, min1, min2, minTotal, secMinTotal, bestS1, bestS2, bestM1, bestM2, pad[3];
    uint32_t x0_1, x1_1, x0_2, x1_2, numPairs, numMin, bestA1, bestA2;
};
struct DevPairXa { uint32_t a1, a2; uint8_t s1, s2, m1, m2; };

struct PairClassifyArgs {
    uint32_t        nPairs;
    const uint32_t *activePairs;                 // optional: slot -> pair index
    const uint32_t *totalOcc, *withError;        // per slot (2 per pair) of the search pass
    uint32_t        mm, maxHits;
    int             terminal, mapq;
    uint8_t        *slotState, *pairNext, *pairMin;   // pairMin is per read (global index)
    uint32_t       *occCount;                    // per read: reset here
};
struct PairSortArgs {
    uint32_t        nPairs;
    const uint32_t *activePairs;
    const uint8_t  *pairNext;
    const uint32_t *occStart, *occCount;
    uint32_t        regionBase;
    uint32_t       *occPos, *occMeta;
    uint64_t       *keys;
    uint32_t       *segBegin, *segEnd;
    uint32_t        idxBits;                     // bits of a hit's place in its list (lists hold at most maxHitsPerEnd hits): the sort keys are position << idxBits | place
};
struct PairSweepArgs {
    uint32_t        nPairs;
    const uint32_t *activePairs;
    uint8_t        *pairNext;
    const uint8_t  *pairMin;
    const uint32_t *finalWithError;
    const uint32_t *occStart;
    uint32_t       *occCount;
    const uint32_t *occCountIn;                  // per slot snapshot taken before the statistics pass
    const uint32_t *occPos, *occMeta, *readLengths;
    uint32_t        insertLow, insertHigh, peMaxOutputPerPair, mm;
    int             terminal;
    DevPair        *pairs;
    uint32_t       *bestIdx, *xaCount;           // per slot-pair
    uint32_t       *xaStart, *xaCountOut;        // per pair (global index)
    // pairs with long lists (more than heavyMin entries on both ends together) are listed by pairStatsKernel and taken by one wave each
    uint32_t        heavyMin, regionBase;
    uint32_t       *heavyList, *heavyCount;      // slots of the listed pairs
    uint32_t       *bestRank, *bestLocal;        // per slot-pair: the optimal pair as (rank of its left leg, number of the partner)
    uint32_t       *scratchW, *scratchR;         // one word each per occurrence of the phase's region (index - regionBase)
};
void launchPairExpand(uint32_t nPairs, const uint32_t *activePairs, uint32_t *readList, hipStream_t s);
void launchPairClassify(const PairClassifyArgs &a, hipStream_t s);
void launchPairSortKeys(const PairSortArgs &a, hipStream_t s);
void launchPairSortBack(const PairSortArgs &a, const uint64_t *keysSorted, const uint32_t *metaSorted, hipStream_t s);
void launchPairSnapshotCounts(uint32_t nPairs, const uint32_t *activePairs, const uint32_t *occCount, uint32_t *occCountIn, hipStream_t s);
void launchPairStats(const PairSweepArgs &a, hipStream_t s);
void launchPairEmit(const PairSweepArgs &a, const uint32_t *xaOffsets, uint32_t xaBase, DevPairXa *xa, hipStream_t s);
void launchPairGatherXa(uint32_t nPairs, const uint32_t *xaStart, const uint32_t *outOffsets, const DevPairXa *xa, DevPairXa *out, hipStream_t s);

// code objects loaded ahead of the first launch (s3dp_kernels.hip: preloadKernels; one function per file that holds kernels)
void preloadKernels();
void preloadFirst(); void preloadUnit(); void preloadCe(); void preloadApi(); void preloadLadder(); void preloadFlank(); void preloadPair(); void preloadRoute();
void preloadSeed(); void preloadDp(); void preloadDpApi();
int copyProbeVariants();
void launchCopyProbe(const uint4 *src, uint4 *dst, size_t n16, int variant, hipStream_t s);
void launchGatherProbe(const uint4 *table, uint64_t nLines, uint32_t blocks, uint32_t itersPerLane, int ilp, int loadsPerLine, int dependent,
                       uint32_t *sink, hipStream_t s);

}  // namespace s3
