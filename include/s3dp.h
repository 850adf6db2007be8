/* s3dp.h -- C-ABI of the MI355X-native SOAP3-dp alignment hot path (libs3dp.so).
 *
 * Plain pointers and sizes only.  Each entry point names the reference interface it replaces
 * (paths relative to aquaskyline/SOAP3-dp).  The reference has no plugin table: the boundary is
 * the set of host functions in alignment.h:40-59 that alignment.cu implements around the CUDA
 * kernels of DV-Kernel.cu, plus the CPU consumer (CPUfunctions.cpp) of their `answers` arrays.
 *
 * Error model: every function returns 0 on success, a negative S3_E* code otherwise, and
 * s3_last_error() gives a message.  (The reference prints and exit(1)s -- alignment.cu:38-42 --
 * the host wrapper reproduces that.)  There is no CPU fallback anywhere behind this ABI: if no
 * HIP device is usable the calls fail with S3_ENODEV.
 */
#ifndef S3DP_H
#define S3DP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S3_OK        0
#define S3_EINVAL   -1
#define S3_ENODEV   -2
#define S3_EHIP     -3
#define S3_ENOMEM   -4
#define S3_EOVERFLOW -5

#define S3_MAX_READ_LENGTH 1024     /* definitions.h:38 */
#define S3_MAX_MISMATCH    4

typedef struct s3_index s3_index;
typedef struct s3_batch s3_batch;
typedef struct s3_dp s3_dp;          /* semi-global DP aligner, see s3_dp_create */

/* One BWT of the 2BWT pair, host arrays in the on-disk layouts with file headers stripped:
 *   bwt : .bwt / .rev.bwt payload, 2-bit chars, 16 per word, MSB first, '$' not stored (2bwt-lib/BWT.c:169-194)
 *   occ : GPU-2BWT .fmv.gpu / .rev.fmv.gpu payload, (ceil(n/128)+1) x 4 words = Occ(c,128x)+C[c]
 *         (writer BGS-Build.cpp:139-192; reader IndexHandler.cpp:158-199)                                    */
typedef struct {
    const uint32_t *bwt;
    size_t          bwtWords;
    const uint32_t *occ;
    size_t          occRecs;
    uint32_t        inverseSa0;
    uint32_t        cumFreq[5];        /* cumulativeFreq[0..4], [4] = textLength */
} s3_bwt_desc;

typedef struct {
    s3_bwt_desc     fwd, rev;
    uint32_t        textLength;
    const uint32_t *sa;                /* suffix array, .sa payload after 6 words: textLength+1 entries, or every saInterval-th (see saInterval) */
    const uint32_t *packedDNA;         /* text, 16 chars/word MSB first as HSPLoad holds it (TextConverter.c:666-720) */
    size_t          packedWords;
    const uint32_t *lkt, *revLkt;      /* 13-mer cumulative lookup tables (.lkt/.rev.lkt payload after 1 word) */
    int             lktSize;           /* LOOKUP_SIZE = 13 (2bwt-flex/LT.h:49) */
    uint32_t        saInterval;        /* SaValueFreq of the .sa file (its header word 5): 0 or 1 = `sa` is the full array; k > 1 = `sa` holds
                                          SA[0], SA[k], SA[2k], ... ((textLength + k) / k entries, 2bwt-lib/BWT.c:254-255).  The device copy is always the
                                          full array: a sampled one is expanded at upload by the reference's own rule (BWTSaValue, BWT.c:1694-1724: walk
                                          backwards through the BWT to the next sampled index, add the steps) -- 12 GB of HBM buy one access per hit */
} s3_index_desc;

/* replaces GPUINDEXUpload (alignment.cu:27-107).  The device copy is re-laid out (one 64-byte line per
 * rank query); SA, packed text and both lookup tables are made resident too so that check-and-extend
 * and SA decode run on the device. */
int  s3_index_upload(int device, const s3_index_desc *desc, s3_index **out);
/* replaces GPUINDEXFree (alignment.cu:109-115) */
void s3_index_free(s3_index *ix);
size_t s3_index_device_bytes(const s3_index *ix);
/* characters the lookup step of an item starts from: the file format's table (lktSize) or the longer tables the library builds in HBM at upload
 * (S3DP_LT_EXTRA = 0 .. 3 more characters; default 3 = 16-mers, 103 GB, for texts of two gigabases and more, 2 from one gigabase, 0 below): entries hold
 * exactly the state the shorter table plus that many exact steps give, so results do not change -- an item just starts that many rank steps further on */
int  s3_index_lookup_chars(const s3_index *ix);
/* FLANK TABLES of check-and-extend (CEKernel's reference: 2bwt-flex/SRA2BWTCheckAndExtend.c:142, BGS-HostAlgnmtAlgo2.cpp:36-176 read the suffix array and then
 * the packed text per candidate).  The library builds, at upload, one 16-byte record per suffix-array entry -- {text position, the 48 characters next to it} -- in
 * the order of the forward index (tasks handed over in backward steps) and of the reverse index (forward steps; csrc/s3dp_flank.hip tells how its order is found
 * without a suffix array of the reverse index): a rejected candidate -- 97 in 100 -- then costs that record instead of two dependent lines.  Same records.
 * s3_index_flank_tables: 0, 1 (forward index only) or 2 -- S3DP_FLANK, and the device memory that is free (S3DP_FLANK_RESERVE_GB).
 * s3_index_fetch_flank: inspection aid (tests compare the tables with a suffix sort of the reversed text): entries [first, first + count) of table 0 (forward
 * index order) or 1 (reverse index order), four words each: position, then 3 x 16 characters MSB first, the adjacent character first. */
int  s3_index_flank_tables(const s3_index *ix);
int  s3_index_fetch_flank(s3_index *ix, int table, uint32_t first, uint32_t count, uint32_t *out);

/* One emitted record of the k-mismatch search, in the reference CPU path's emission order
 * (BGS-HostAlgnmtAlgo2.cpp: SA ranges -> SAList via addSAToSAList :1617,1692; check-and-extend hits
 * -> OCCList via addToOCCList :169).  kind 0: SA range [a,b];  kind 1: text position a.
 * strand 1 '+', 2 '-' (QUERY_POS_STRAND / QUERY_NEG_STRAND);  cas = case index in the model. */
typedef struct {
    uint32_t a, b;
    uint8_t  kind, strand, nmis, cas;
} s3_hit;

typedef struct {
    uint64_t rankQueries;      /* all-symbol rank queries issued (the roofline unit) */
    uint64_t ltLookups;        /* lookup-table entries read: one per item from the long tables, two / four 4-byte entries from the 13-mer tables */
    uint64_t saReads;          /* suffix-array entries read by check-and-extend (= candidates verified) */
    uint64_t ceWindows;        /* 32-character compare windows of check-and-extend */
    uint64_t records;          /* sequence numbers given out: SA-range records + check-and-extend candidates (accepted or not) */
} s3_counters;

/* Batch object: device-resident packed reads + all work buffers (persistent, no per-call allocation).
 * Replaces the per-batch cudaMalloc/cudaMemcpy of perform_round1_alignment (alignment.cu:118-219). */
int  s3_batch_create(s3_index *ix, uint32_t maxReads, uint32_t wordPerQuery, s3_batch **out);
void s3_batch_free(s3_batch *b);
/* queries: the reference's packed layout -- 32-interleaved words, 16 chars/word LSB first
 * (word w of read r at r/32*32*wpq + r%32 + 32*w; DV-Kernel.cu:4268-4270); readLengths[n]. Host pointers.
 * The layout works in groups of 32 reads: `queries` must hold ceil(n/32)*32 x wordPerQuery words (the reference allocates its read
 * arrays the same way, SOAP3-DP.cu:298-300); all of them are copied. */
int  s3_batch_upload(s3_batch *b, const uint32_t *queries, const uint32_t *readLengths, uint32_t n);
/* Long-read mode = the `seedLengths` argument of all_valid_alignment / all_best_single_alignment / best_single_alignment
 * (alignment.cu:534,1422,1377; the array is built at alignment.cu:2473-2491: 100 for reads longer than 120 bases, else the read
 * length).  Call after s3_batch_upload; NULL switches it off.  Searches then run on the first seedLengths[i] (16..128) bases of read
 * i without a hit limit, and s3_batch_align_single verifies the remaining bases of every hit against the packed text
 * (validateAlignments, CPUfunctions.cpp:1129-1222): positions and mismatch counts reported are those of the whole read. */
int  s3_batch_set_seed_lengths(s3_batch *b, const uint32_t *seedLengths);

/* All cases of the <= numMismatch model, both strands, for every read of the batch: the work of
 * kernel / kernel_4mismatch_{1,2} (DV-Kernel.cu:4249,4505,4741) + round 2 + the CPU re-search of
 * overflowing reads (collect_all_answers, CPUfunctions.cpp:1226-1416), with results identical to the
 * CPU path ProcessReadDoubleStrand2 (CPUfunctions.cpp:555).  Asynchronous on the batch stream. */
int  s3_batch_search(s3_batch *b, uint32_t numMismatch, uint32_t maxOutputPerRead);
/* Wait and fetch: hitOffsets[n+1] (CSR), hits, per read TotalOccurrences and WithError[8]
 * (SRAQueryResultCount).  Any pointer may be NULL.  hitCap = capacity of hits in records. */
int  s3_batch_fetch_hits(s3_batch *b, uint64_t *hitOffsets, s3_hit *hits, size_t hitCap,
                         uint32_t *totalOcc, uint32_t *withError, uint64_t *nHitsOut);

/* Single-end all-best ladder on the device (all_best_single_alignment, alignment.cu:1422-1465 +
 * the MAPQ special case of hostKernel, CPUfunctions.cpp:1755-1801): phases mm = 1..numMismatch, reads
 * whose best hit has exactly mm mismatches are deferred to phase mm+1 (or re-searched with the mm+1
 * model in the terminal phase), SA ranges decoded through the resident suffix array.
 * Results per read: phase it was resolved in (0 = unaligned), occurrence CSR in the CPU path's
 * final order (check-and-extend hits first, then SA ranges expanded; SAList.cpp:392-418). */
int  s3_batch_align_single_allbest(s3_batch *b, uint32_t numMismatch, uint32_t maxOutputPerRead);
/* same, for every report type of the reference's -h option: 1 all valid (one phase at numMismatch, alignment.cu:2530),
 * 2 all best, 3 unique best / 4 random best (best_single_alignment, alignment.cu:1377: phases 0..numMismatch, no MAPQ special
 * case; phase[] is then mismatch-phase + 1 so that 0 still means unaligned). */
int  s3_batch_align_single(s3_batch *b, uint32_t numMismatch, uint32_t maxOutputPerRead, int reportType);
int  s3_batch_fetch_alignments(s3_batch *b, uint8_t *phase, uint32_t *withError /* n x 8 */,
                               uint64_t *occOffsets /* n+1 */, uint32_t *occPos, uint8_t *occStrand, uint8_t *occMis,
                               size_t occCap, uint64_t *nOccOut);

/* ---- streamed single-end alignment ------------------------------------------------------------------------------------------
 * Replaces the batch loop of the reference's main (SOAP3-DP.cu:607-800: read a batch, align it, next) together with its overlap of the device
 * work of batch i+1 with the CPU threads of batch i (alignment.cu:555-561, 1005-1031).  A stream owns `depth` slots, each one s3_batch on its own
 * HIP stream + pinned host staging + a worker thread running upload -> s3_batch_align_single -> download; with depth 3 the upload of batch
 * i+1, the kernels of batch i and the download of batch i-1 overlap, and so do the caller's parsing and formatting.
 *   s3_stream_acquire   blocks until the next slot is free and returns its pinned input arrays (the layouts of s3_batch_upload; room for
 *                       maxReads reads; queries must be written for ceil(n/32)*32 reads, zero beyond n): parse straight into them
 *   s3_stream_submit    hands the acquired slot's n reads to its worker; seed lengths as in s3_batch_set_seed_lengths when useSeedLengths
 *   s3_stream_next      blocks until the OLDEST submitted batch is done and returns views of its results in pinned host memory (valid until
 *                       s3_stream_release): phase / withError as s3_batch_fetch_alignments, occOffsets[n + 1] (32-bit: a batch holds fewer than
 *                       2^32 occurrences), occPos, occMeta = strand | mismatches << 8; `batch` is the slot's batch with the reads still resident
 *                       (for the DP stages)
 *   s3_stream_release   gives the oldest slot back
 * One thread may drive all four calls as long as it never holds more than `depth` unreleased batches; or a producer thread acquires / submits
 * while a consumer thread takes results.  Results come back in submission order.
 * Host requirement: batches in flight run on their own HIP streams; the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues
 * (default 4), and streams that share a queue run in order.  A host that keeps more than two batches in flight should set GPU_MAX_HW_QUEUES=16 in its
 * environment BEFORE its first HIP call (the soap3-dp executable and the Python binding do); the library does not change its host's environment. */
typedef struct s3_stream s3_stream;
typedef struct {
    uint64_t        ticket;
    uint32_t        n;
    const uint8_t  *phase;
    const uint32_t *withError;
    const uint32_t *occOffsets, *occPos, *occMeta;
    float           deviceMs;        /* device time of the batch's ladder (HIP events on its stream) */
    s3_batch       *batch;
} s3_stream_result;
int  s3_stream_create(s3_index *ix, uint32_t maxReads, uint32_t wordPerQuery, int depth, s3_stream **out);
int  s3_stream_acquire(s3_stream *st, uint32_t **queries, uint32_t **readLengths, uint32_t **seedLengths /* may be NULL */);
int  s3_stream_submit(s3_stream *st, uint32_t n, int useSeedLengths, uint32_t numMismatch, uint32_t maxOutputPerRead, int reportType, uint64_t *ticket);
int  s3_stream_next(s3_stream *st, s3_stream_result *res);
int  s3_stream_release(s3_stream *st);
void s3_stream_free(s3_stream *st);
/* The streams' pinned host staging (the packed reads going in, the results and occurrence lists coming out) comes from a cache of the process: pinning is slow
 * (0.2-0.4 s per GB) and no kernel of any HIP stream starts while the runtime pins.  s3_host_staging_reserve puts what `depth` slots of s3_stream_create(ix,
 * maxReads, wordPerQuery, depth) pin into the cache ahead of time, with room for occPerRead occurrences per read in the lists (the soap3-dp executable: 48,
 * while the index files are read); a slot takes the smallest block that fits and pins a fresh one when none does (a list that outgrows its block); blocks a
 * stream gives back stay in the cache (up to 4 GB) until s3_host_staging_release.  No counterpart in the reference: its result buffers are pageable
 * (alignment.cu:1005-1031 copies them back with cudaMemcpy). */
int  s3_host_staging_reserve(uint32_t maxReads, uint32_t wordPerQuery, int depth, uint32_t occPerRead);
void s3_host_staging_release(void);

/* ---- several devices -------------------------------------------------------------------------------------------------------------
 * The reference takes ONE device (`-c <id>`, SOAP3-DP.cu:219-220, README.md:523-536).  The path shards by reads, so a group of devices each holds
 * the whole index (s3_group_create uploads it to every member at once) and aligns its own batches -- s3_group_member_of_batch is the
 * static deal (batch i to member i mod n) for callers that want a reproducible assignment; the `soap3-dp -c a,b` command line instead hands a chunk to
 * whichever member is free (which member's files a read lands in then depends on timing: the UNION of the members' files is the result, as the reference's
 * .gout.N files are the union over its CPU threads).  A batch holds whole pairs, mates stay together (CPUfunctions.cpp:1982,2029).  Batches, streams and DP aligners are created on
 * s3_group_index(g, k) like on any index.  The single exchange step is the sum of the counters at the end (the `[Main] Overall ...` numbers,
 * SOAP3-DP.cu:816-830): s3_group_allreduce takes nMembers rows of nCounters values and leaves the column sums in every row, computed by one
 * ncclAllReduce over the members' devices (RCCL; the communicators are made while the index uploads; a group of one member has nothing to exchange and
 * makes none; members that share a device -- a test on one GPU -- are summed on the host instead, RCCL needs distinct devices).  Output: one set of files per member, as the reference writes one .gout.N per CPU thread (SOAP3-DP.cu:251-258). */
typedef struct s3_group s3_group;
int  s3_group_create(const int *devices, int nDevices, const s3_index_desc *desc, s3_group **out);
int  s3_group_size(const s3_group *g);
int  s3_group_device(const s3_group *g, int member);
s3_index *s3_group_index(s3_group *g, int member);
int  s3_group_member_of_batch(const s3_group *g, uint64_t batchNumber);
int  s3_group_allreduce(s3_group *g, uint64_t *counters, int nCounters);
void s3_group_free(s3_group *g);

/* ---- default DP for half-aligned pairs -------------------------------------------------------------------------------------
 * Replaces HalfEndAlignmentEngine (DV-DPfunctions.cu:1888-2540): for every occurrence of the aligned end (HalfEndOccStream order) the
 * window the insert size allows is cut from the RESIDENT text (HalfEndAlgnBatch::pack :2027-2109: '+' aligned end -> mate to its right
 * on '-', '-' aligned end -> mate to its left on '+'; soft-clip allowances and anchors as there), the mate is packed in that strand
 * (packRead / repackDNA :2111-2167, on the device), the semi-global DP runs (s3_dp) and the traceback is decoded like
 * DP_Space::algnmtCPUThread (:2329-2441): CIGAR (CigarStringEncoder, DV-DPfunctions.h:548), edit distance
 * I + D + (L match + gapPenalty - score) / (match - mismatch), insert size.  `b` holds the reads (s3_batch_upload; reads 2p / 2p+1 are
 * a pair), `h` must have been created with maxReadLength >= the longest read and maxDNALength >= insertHigh.  One result per
 * candidate window, in input order; results and the CIGAR pool are malloc'ed: release with s3_free.  Pinned against the reference's own
 * engine (tests/test_gpu_dp_pipeline.py). */
typedef struct { uint32_t readID, pos; uint8_t strand, mismatchCount, pad[2]; } s3_occ;        /* SRAOccurrence of the ALIGNED end */
typedef struct {
    uint32_t readID;                /* first read of the pair */
    uint32_t algnmt_1, algnmt_2;    /* text positions of end 1 / end 2 (0xFFFFFFFF: the DP found nothing above the cutoff) */
    int32_t  score_1, score_2;      /* mismatches of the BWT-aligned end, DP score of the other */
    uint8_t  strand_1, strand_2, whichFromDP /* 0: end 1 from DP, 1: end 2, 2: none */, pad;
    uint32_t editdist, insertSize, num_sameScore;
    uint32_t cigarOffset;           /* into the CIGAR pool, 0xFFFFFFFF if none */
} s3_dp_half_result;
int  s3_dp_half_end(s3_dp *h, s3_batch *b, const uint32_t *readLengths, const s3_occ *occ, size_t nOcc, int insertLow, int insertHigh,
                    const int cutoffThreshold[2], int softClipLeft, int softClipRight, s3_dp_half_result **results, size_t *nResults, char **cigarPool);

/* ---- seeding for the DP stages (single reads) --------------------------------------------------------------------------------
 * Replaces SingleEndSeedingBatch::packSeeds (DV-DPfunctions.cu:1058-1100), single_1_mismatch_alignment2 (alignment.cu:1839-1894: every
 * seed searched exactly, the seeds without a hit once more with one mismatch), the hit limit of hostKernelSingle
 * (CPUfunctions.cpp:2843-2876: a seed with more than maxHitNum hits is dropped, status 4), decodePositions and singleMerge
 * (DV-DPfunctions.cu:1102-1219: hit -> estimated read position, sorted by strand / read / position, positions within 50 of the last kept
 * one collapse).  `reads` holds the uploaded reads, `seedBatch` is a second batch on the same index (wordPerQuery >= 8, room for every
 * seed: at most 12 per read below 400 bases).  Candidates come back sorted by (strand, read, position); seedStatus (optional) holds per
 * seed 0 = no hit, 1 = hits used, 4 = dropped for too many hits.  Both arrays are malloc'ed: release with s3_free.  Pinned through the reference's own engines
 * (tests/test_gpu_dp_pipeline.py). */
typedef struct { uint32_t readID, pos; uint8_t strand, pad[3]; } s3_seed_candidate;
int  s3_seed_candidates(s3_batch *reads, s3_batch *seedBatch, const uint32_t *readLengths, const uint32_t *readIDs /* NULL = every read */, size_t nReadIDs,
                        int stage, uint32_t maxHitNum, s3_seed_candidate **candidates, size_t *nCandidates, uint8_t **seedStatus, size_t *nSeeds);

/* ---- DP for single reads (after s3_seed_candidates) ------------------------------------------------------------------------------
 * Replaces SingleEndAlignmentEngine (DV-DPfunctions.cu:1373-1770): window = candidate position -/+ margin (readLength > 100 ? readLength / 4
 * : 25), read packed in the candidate's strand, results below the cutoff dropped; CIGAR / edit distance as in s3_dp_half_end.
 * Results and the CIGAR pool are malloc'ed: release with s3_free.  Pinned through the reference's own engines
 * (tests/test_gpu_dp_pipeline.py). */
typedef struct { uint32_t readID, algnmt; int32_t score; uint32_t editdist, num_sameScore, cigarOffset; uint8_t strand, pad[3]; } s3_dp_single_result;
int  s3_dp_single_end(s3_dp *h, s3_batch *b, const uint32_t *readLengths, const s3_seed_candidate *candidates, size_t nCandidates, int cutoffThreshold,
                      int softClipLeft, int softClipRight, s3_dp_single_result **results, size_t *nResults, char **cigarPool);

/* ---- deep DP for pairs whose ends both failed the BWT stage --------------------------------------------------------------------
 * Replaces DeepDP_Space (DV-DPfunctions.cu:2552-3900; driver DV-DPForBothUnalign.cu): seeds of both ends (PairEndSeedingBatch), their hits
 * sorted per end and strand (decodePositions), '+' hits of one end paired with '-' hits of the other inside the insert window
 * (pairEndMerge: left run compressed by 50, window [insertLow - L - margin, insertHigh - L + margin]), the left end aligned first, the right
 * end's window cut by the left hit and the insert bounds (PairEndAlgnBatch::packLeft / packRight), results where both scores pass their
 * cutoffs (DP2CPUAlgnThread).  pairFirstReads: index of the first read (even) of every pair to process, ascending.  stage =
 * S3_STAGE_DEEP_DP_ROUND1 / ROUND2.  tooManyHits (optional, one byte per input pair): bit 0 = a seed of the pair was dropped for too many hits,
 * bit 1 = the pair had candidate positions (pairs with bit 0 and without bit 1 are what the reference seeds again in round 2, :3110-3121).
 * Arrays are malloc'ed: release with s3_free.  Pinned through the reference's own engines
 * (tests/test_gpu_dp_pipeline.py). */
typedef struct {
    uint32_t readID;                 /* first read of the pair */
    uint32_t algnmt_1, algnmt_2;
    int32_t  score_1, score_2;
    uint32_t editdist_1, editdist_2, insertSize, num_sameScore_1, num_sameScore_2;
    uint32_t cigarOffset_1, cigarOffset_2;
    uint8_t  strand_1, strand_2, pad[2];
} s3_dp_deep_result;
int  s3_dp_deep(s3_dp *h, s3_batch *b, s3_batch *seedBatch, const uint32_t *readLengths, const uint32_t *pairFirstReads, size_t nPairs, int stage,
                uint32_t maxHitNum, int insertLow, int insertHigh, const int cutoffThreshold[2], int softClipLeft, int softClipRight,
                s3_dp_deep_result **results, size_t *nResults, char **cigarPool, uint8_t **tooManyHits);

/* ---- new default DP: pairs with an end the BWT stage aligned to too many places for the default DP ----------------------------------------
 * Replaces newSemiGlobalDP (DV-SemiDP.cu:345; SemiDPWrapper::run2 -> newDefaultDP :237-246): the aligned end's hits are taken as they are and join the
 * seed positions of BOTH ends (PairEndSeedingEngine's driver, DV-DPfunctions.cu:3056-3079, calls PairEndSeedingBatch::packSeeds ( occ, stage ) :2709-2731,
 * which packs the seeds of both reads of the pair -- packSeedsOneSide :2733-2776 has no caller; S3_STAGE_NEW_DEFAULT_DP), both lists go through the
 * deep DP's pairing and its two DP rounds (pairEndMerge, PairEndAlignmentEngine), so BOTH ends are aligned by the DP and the results are deep-DP
 * results.  occ: the aligned ends' hits, grouped by pair, pairs ascending (the ReadInputForDP stream).  pairFirstReads / pairFlags (optional
 * outputs): the pairs found in occ and their flags as in s3_dp_deep.  Pinned through the reference's own engines
 * (tests/test_gpu_dp_pipeline.py). */
int  s3_dp_new_default(s3_dp *h, s3_batch *b, s3_batch *seedBatch, const uint32_t *readLengths, const s3_occ *occ, size_t nOcc, uint32_t maxHitNum,
                       int insertLow, int insertHigh, const int cutoffThreshold[2], int softClipLeft, int softClipRight, s3_dp_deep_result **results,
                       size_t *nResults, char **cigarPool, uint8_t **pairFlags, uint32_t **pairFirstReads, size_t *nPairs);

/* ---- parameters of the DP stages (host-side integer rules) ------------------------------------------------------------------
 * s3_dp_stage_parameters replaces getParameterForSingleDP / DefaultDP / NewDefaultDP / DeepDP and getMaxHitNumForDefaultDP
 * (CPUfunctions.cpp:36-260) for one read length; s3_seed_positions replaces getSeedPositions (definitions.h:323-442): the seed length,
 * the number of seeds and their offsets in the read for the seeding of a DP stage. */
#define S3_STAGE_SINGLE_DP       1
#define S3_STAGE_DEFAULT_DP      2
#define S3_STAGE_NEW_DEFAULT_DP  3
#define S3_STAGE_DEEP_DP_ROUND1  4
#define S3_STAGE_DEEP_DP_ROUND2  5
typedef struct { int dpScoreThreshold /* < 0: DEFAULT = ceil(0.3 L) */, softClipLeft, softClipRight; } s3_dp_ini;   /* [DP] / [Clipping] ini keys */
typedef struct { int cutoffThreshold, maxHitNum, seedLength, sampleDist, tailTrimLen, softClipLeft, softClipRight, singleDPSeedNum; } s3_dp_stage_params;
int  s3_dp_stage_parameters(int stage, uint32_t readLength, const s3_dp_ini *ini, s3_dp_stage_params *out);
int  s3_seed_positions(int stage, int readLength, int *seedLength, int *positions /* room for 128 */, int *seedNum);

/* ---- paired-end reads, DP stages off (`soap3-dp pair <index> <r1> <r2> -u <max insert> -v <min insert> -s k`) -------------------------
 * Reads 2p and 2p+1 of the batch are the two ends of pair p (the interleaving loadPairReadsGz2 produces, QueryParser.cpp:177).
 * s3_batch_align_pair replaces soap3_dp_pair_align's ladders with DP disabled (alignment.cu:1965-2013: all_best_alignment :1236 = phases
 * 1, 2, k for -h 2; four_phases_alignment :1119 = phases 0, 1, 2, k for -h 3/4; one all_valid_alignment pass for -h 1), the PAIR_END_READ
 * branch of hostKernel (CPUfunctions.cpp:1980-2560: deferral of a pair whose first end's best hit has exactly the phase's mismatches,
 * the extra (mm+1) search of such an end, the carry-over of pairs without a proper pairing) and the pairing itself
 * (PEMappingOccurrences + PEStatsPEOutput, PEAlgnmt.cpp:480,834): '+' left leg, '-' right leg,
 * insertLow <= right.pos + len(2nd end) - left.pos <= insertHigh.  maxHitsPerEnd is the ini's MaxHitsEachEndForPairing
 * (CPUfunctions.cpp:845), peMaxOutputPerPair its MaxOutputPerPair. */
typedef struct {
    uint8_t  phase;               /* 0: no proper pairing (the occurrence lists of both ends are returned); else pairing phase + 1 */
    uint8_t  min1, min2;          /* fewest mismatches of each end in that phase (255: end without hit, or -h 3/4) */
    uint8_t  minTotal, secMinTotal;   /* mismatch sum of the optimal pair / of the last demoted optimal (127: none), PEStatsPEPairList */
    uint8_t  bestS1, bestS2, bestM1, bestM2;   /* the optimal pair: strand (1 '+', 2 '-') and mismatches of each end */
    uint8_t  pad[3];
    uint32_t x0_1, x1_1, x0_2, x1_2;  /* WithError[min], WithError[min + 1] of each end: X0 / X1 (-h 1/2) */
    uint32_t numPairs, numMin;    /* valid pairings; pairings with the optimal mismatch sum */
    uint32_t bestA1, bestA2;      /* the optimal pair: text position of each end */
} s3_pair_result;
typedef struct { uint32_t a1, a2; uint8_t s1, s2, m1, m2; } s3_pair_xa;   /* one more pairing with the optimal mismatch sum (XA:Z) */

int  s3_batch_align_pair(s3_batch *b, uint32_t numMismatch, uint32_t maxHitsPerEnd, int reportType, uint32_t insertLow, uint32_t insertHigh,
                         uint32_t peMaxOutputPerPair);
/* pairs[n/2]; xaOffsets[n/2 + 1] + xa[]: the other optimal pairings in pairing order, at most peMaxOutputPerPair - 1 per pair;
 * occOffsets[n + 1] + occ arrays: occurrence lists of the reads of pairs with phase == 0, in the order unproperlypairOutputSAMAPI
 * (BGS-IO.cpp:2582) walks them (sorted by position when both ends had hits).  Any output pointer may be NULL. */
int  s3_batch_fetch_pairs(s3_batch *b, s3_pair_result *pairs, uint64_t *xaOffsets, s3_pair_xa *xa, size_t xaCap, uint64_t *nXa,
                          uint64_t *occOffsets, uint32_t *occPos, uint8_t *occStrand, uint8_t *occMis, size_t occCap, uint64_t *nOcc);

/* fromSaRange[i] = 1 when occurrence i of the lists above was expanded from an SA range, 0 when it came from check-and-extend.  The
 * reference keeps the two kinds in separate lists (SAList / OCCList) and its default-DP stage reads the SA ranges of a pair first
 * (HalfEndOccStream::fetchNextOcc, DV-DPfunctions.cu:1901-1960), while the lists here hold the check-and-extend hits first. */
int  s3_batch_fetch_pair_origins(s3_batch *b, uint8_t *fromSaRange, size_t cap);

/* ---- routing of the pairs the BWT ladder left without a proper pairing, on the device (round 4) ------------------------------------------------------
 * Replaces the needProceedDP branch of hostKernel (CPUfunctions.cpp:2153-2262, 2444-2543) that decides, pair by pair, which DP stage gets which of the two
 * ends' hit lists: a list longer than the stage's hit limit keeps its best (keepSecondBest: and second-best) stratum (retainAllBest /
 * retainAllBestAndSecBest, SAList.cpp:140,290); no hits at all -> deep DP; hits on one end -> that end's list, SA ranges before check-and-extend hits
 * (HalfEndOccStream::fetchNextOcc, DV-DPfunctions.cu:1901-1960), to the default DP or -- still above maxHitNum -- to the new default DP; hits on both ends
 * -> both lists to the default DP when both are within the limit, else the shorter one (the second end's when equal) to the default / new default DP and
 * the other one kept for the unpaired output only.  Call it after s3_batch_align_pair on the same batch; the three lists (entries grouped by read, reads
 * ascending) stay on the device:
 *   s3_batch_fetch_routing    per pair the S3_ROUTE_* bits; any of the three lists (NULL: not wanted)
 *   s3_batch_fetch_kept       the kept lists of selected reads (what the unpaired output of a pair that every stage failed on prints): offsets[nReads + 1]
 *                             and the entries behind each other, malloc'ed (s3_free)
 *   s3_dp_new_default_routed  s3_dp_new_default on the new-default list where it lies (it never comes to the host)
 * maxHitNum[e]: the default DP's hit limit for end e (s3_dp_stage_parameters(S3_STAGE_DEFAULT_DP)).  Pinned: the routing of the reference's own hostKernel
 * (tests/test_gpu_cli_sam.py::test_cli_pair_dp_routing_matches_the_reference, fixtures g4 / g5). */
#define S3_ROUTE_UNPAIRED      1     /* the ladder found no proper pairing (every other bit implies this one) */
#define S3_ROUTE_DEEP          2     /* no hit on either end: deep DP */
#define S3_ROUTE_DEFAULT       4     /* the pair has entries in the default-DP list */
#define S3_ROUTE_NEW_DEFAULT   8     /* the pair has entries in the new-default-DP list ... */
#define S3_ROUTE_GIVEN_END_2  16     /* ... and they are its SECOND end's hits */
typedef struct {
    uint64_t nDefault, nNewDefault, nOther;                         /* entries of the three lists */
    uint32_t nUnpaired, nDeepPairs, nDefaultPairs, nNewDefaultPairs;
} s3_pair_routing;
int  s3_batch_route_pairs(s3_batch *b, const uint32_t maxHitNum[2], int keepSecondBest, s3_pair_routing *out);
/* the result of a ladder that was not run (pairs of reads above 150 bases skip the BWT stage, SOAP3-DP.cu:658-662): no pairing, no occurrence; routing it
 * sends every pair to the deep DP */
int  s3_batch_no_pair_result(s3_batch *b);
int  s3_batch_fetch_routing(s3_batch *b, uint8_t *pairRoute /* n / 2 */, s3_occ *defaultOcc, size_t defaultCap, s3_occ *newDefaultOcc, size_t newDefaultCap,
                            s3_occ *otherOcc, size_t otherCap);
int  s3_batch_fetch_kept(s3_batch *b, const uint32_t *reads, size_t nReads, uint64_t *offsets, s3_occ **occ, uint64_t *nOcc);
int  s3_dp_new_default_routed(s3_dp *h, s3_batch *b, s3_batch *seedBatch, const uint32_t *readLengths, uint32_t maxHitNum, int insertLow, int insertHigh,
                              const int cutoffThreshold[2], int softClipLeft, int softClipRight, s3_dp_deep_result **results, size_t *nResults,
                              char **cigarPool, uint8_t **pairFlags, uint32_t **pairFirstReads, size_t *nPairs);

/* Width of this batch's persistent search grids in workgroups per compute unit (0: all that are resident at once, four).  The reference sizes its grids by the
 * batch (alignment.cu:118-221, blocksNeeded); here the grids are persistent and as wide as the device, and a batch whose search shares the device with the DP
 * stages of earlier chunks (SOAP3-DP.cu:394-800 runs them one after the other) leaves a workgroup per compute unit to their kernels: the paired command line
 * with its DP stages on asks for three (10 M pairs: 3.40 -> 3.18 s, round 5).  S3DP_BLOCKS_PER_CU overrides it. */
int  s3_batch_set_grid(s3_batch *b, unsigned blocksPerCu);
int  s3_batch_sync(s3_batch *b);
/* device time of the work enqueued by the last search/align call, measured with HIP events on the batch stream;
 * searchMs = time inside the BWT-search kernels only. */
int  s3_batch_last_timing(s3_batch *b, float *totalMs, float *searchMs, uint32_t *searchLaunches);
/* device time inside the check-and-extend kernels of the last call (they run right behind each BWT-search kernel; 0 launches when check-and-extend
 * is kept inside the search kernel, S3DP_CE=inline) */
int  s3_batch_last_ce_timing(s3_batch *b, float *ceMs, uint32_t *ceLaunches);
int  s3_batch_counters(s3_batch *b, s3_counters *out);      /* device-side tallies of the last call (if enabled) */
/* The same per KERNEL of the search passes of the last call -- out[0] first pass (s3::firstKernel), out[1] trunk pass (s3::firstKernel, trunk role), out[2] unit
 * pass (s3::unitKernel), out[3] check-and-extend (s3::ceKernel) -- with each kernel's device time from HIP events around its launches on the batch's stream, so
 * that a kernel's algorithmic bytes (32 B per rank query ...) can be priced against ITS duration (bench.py roofline.per_kernel).  Measurement
 * aid; no counterpart in the reference (its kernels are timed by the CUDA profiler). */
/* listedRankQueries (first pass): rank queries spent on reads the pass then listed for the second pass -- thrown away, the second pass starts over.
 * accepted, ceRejected (check-and-extend): candidates accepted, and where the rejected ones were decided -- ceRejected[(dir * 3 + budget) * 3 + where]: dir 0 the
 * candidate was handed over in a backward step (the characters to compare lie to the LEFT of its suffix), 1 in a forward step; budget = mismatches it may still
 * take in that step (0, 1, 2 and more); where 0 = the 32 characters next to the matched part alone exceed the budget, 1 = the rest of that step decides, 2 = a
 * later step of the model. */
typedef struct { float ms; uint32_t launches; uint64_t rankQueries, ltLookups, saReads, ceWindows, listedRankQueries, accepted, ceRejected[18]; } s3_kernel_stats;
int  s3_batch_kernel_stats(s3_batch *b, s3_kernel_stats out[4]);
int  s3_batch_enable_counters(s3_batch *b, int on);

/* one-shot convenience (host in, host out), mirrors the call shape of perform_round1_alignment:
 * creates a temporary batch.  Arrays are malloc()ed for the caller, free with s3_free(). */
int  s3_bwt_search(s3_index *ix, const uint32_t *queries, const uint32_t *readLengths, uint32_t n, uint32_t wordPerQuery,
                   uint32_t numMismatch, uint32_t maxOutputPerRead,
                   uint64_t **hitOffsets, s3_hit **hits, uint32_t **totalOcc, uint32_t **withError);
void s3_free(void *p);

/* host-side model tables (SRAModelConstruct, 2bwt-flex/SRA2BWTMdl.c:4403) exposed for tests:
 * writes one text line per case in the same form as the reference dump used by the oracle harness. */
int  s3_model_describe(int readLength, int maxError, char *buf, size_t bufLen);

/* ------------------------------------------------------------------------------------------------------------------
 * Semi-global affine-gap DP extension -- replaces SemiGlobalAligner::{decideConfiguration,init,performAlignment,freeMemory}
 * (DV-DPfunctions.h:120-164, DV-DPfunctions.cu:538-741) and its kernels SemiGlobalAligntment / GPUBacktrack (:243-512).
 * Sequences use the reference's packing (DV-DPfunctions.cu:57-59): 1-based characters, 16 per word with the first one in
 * the most significant bits, word w of problem t at (t/32)*(ceil(maxLen/16)*32) + t%32 + 32*w.  Arrays are host pointers. */
typedef struct { int matchScore, mismatchScore, gapOpenScore, gapExtendScore; } s3_dp_scores;

/* decideConfiguration (:538-619): with 288 GB of HBM the full table (scheme 1, maxDPTableLength = maxDNALength) always
 * fits; scheme 2 (score pass + window of L + L/2 + 8 columns) is kept because its results differ and a caller may ask for it. */
int  s3_dp_config(int maxReadLength, int maxDNALength, int wantScheme, int *scheme, int *maxDPTableLength, int *patternLength);
int  s3_dp_create(int device, uint32_t maxProblems, uint32_t maxReadLength, uint32_t maxDNALength, uint32_t maxDPTableLength,
                  int scheme, const s3_dp_scores *sc, s3_dp **out);
/* performAlignment (:669-721).  clipRt is in/out (the kernel overwrites it with the actual right clip, :308-309); scores,
 * hitLocs (alignment START column when score >= cutoff), maxScoreCounts and pattern (patternLength bytes per problem:
 * reversed op string of M m D I S and V<n> run markers, 0-terminated) are outputs.  clipLt/clipRt/anchor arrays may be NULL. */
int  s3_dp_align(s3_dp *h, const uint32_t *packedDNA, const uint32_t *DNALengths, const uint32_t *packedReads, const uint32_t *readLengths,
                 const int32_t *cutoffThresholds, uint32_t nProblems,
                 const uint32_t *clipLtSizes, uint32_t *clipRtSizes, const uint32_t *anchorLeftLocs, const uint32_t *anchorRightLocs,
                 int32_t *scores, uint32_t *hitLocs, uint32_t *maxScoreCounts, uint8_t *pattern);
int  s3_dp_last_timing(s3_dp *h, float *ms);      /* device time of the last s3_dp_align (fill + backtrack), HIP events */
/* running totals of an aligner since s3_dp_create, over every launch of its kernels (s3_dp_align and the engines s3_dp_half_end / _deep /
 * _new_default / _single_end): device milliseconds, problems, cells (read length x window length per problem) -- what a giga-cell-updates
 * figure is made of; the reference's TimeRecorder (DV-DPfunctions.h:446-747) plays this part, unused in its release build */
int  s3_dp_totals(s3_dp *h, double *kernelMs, uint64_t *problems, uint64_t *cells);
void s3_dp_free(s3_dp *h);

/* measurement aid, not part of the alignment path: rate of random 64-byte-line gathers on this device
 * (the practical ceiling of a rank-query stream).  Reports the average kernel time and lines read per launch. */
int  s3_probe_gather(int device, size_t tableBytes, uint32_t blocks, uint32_t itersPerLane, int ilp, int loadsPerLine, int dependent,
                     int repeats, float *msOut, uint64_t *linesOut);

/* measurement aid: streaming bandwidth of this device with an own copy kernel (16-byte loads and stores, grid-stride): bytes READ PLUS WRITTEN per
 * second is the denominator of the roofline fraction (bench.py).  Reports the average time of one copy of `bytes`. */
int  s3_probe_copy(int device, size_t bytes, int repeats, float *msOut);

int  s3_device_count(void);
/* the host NUMA node device `device` is attached to (-1 if the system does not say): where a host program keeps the threads and buffers that feed it.
 * (No counterpart in the reference, which leaves placement to the operating system; soap3_dp_main.cpp binds to it unless S3DP_NUMA_BIND=0.) */
int  s3_device_host_node(int device, int *node);
const char *s3_last_error(void);
const char *s3_version(void);

#ifdef __cplusplus
}
#endif
#endif
