/* s3_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's 2BWT exact / k-mismatch search path:
 * the model-driven search of BGS-HostAlgnmtAlgo2.cpp driven by the case tables of
 * 2bwt-flex/SRA2BWTMdl.c, the 13-mer lookup table start, check-and-extend
 * (2bwt-flex/SRA2BWTCheckAndExtend.c), SA decode (SAList.cpp:392) and the single-end
 * all-best ladder + MAPQ of CPUfunctions.cpp / BGS-IO.cpp.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this
 * library, and only as the checker.  The product (soap3-dp_amd/) never links it.
 *
 * Parity pin: checked against oracle/_ref (the real reference CPU path compiled from
 * /root/reference by oracle/ref.mk) on every fixture in tests/golden/ -- see
 * tests/test_oracle_vs_reference.py.
 */
#ifndef S3_ORACLE_H
#define S3_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S3O_MAX_STEPS 5
#define S3O_MAX_CASES 10
#define S3O_MAX_READ  1024

enum { S3O_STEP_BACK_LOOKUP = 2, S3O_STEP_FWD_LOOKUP = 4, S3O_STEP_BACK_ONLY = 5, S3O_STEP_BIDIR = 6 };

typedef struct {
    int type, start, end, ceThreshold, ceStart, ceEnd, minErr, maxErr;
    int dir, len;                 /* computed: dir = +1 forward / -1 backward */
} s3o_step;

typedef struct {
    int id, nSteps;
    s3o_step steps[S3O_MAX_STEPS];
    int16_t leftMost[S3O_MAX_READ]; /* leftMostAligned[pos], SRA2BWTMdl.c:4543-4571 */
} s3o_case;

typedef struct {
    int readLength, maxError, nCases;
    s3o_case cases[S3O_MAX_CASES];
} s3o_model;

typedef struct {
    uint32_t  n;                  /* textLength */
    uint32_t  inverseSa0;
    uint32_t  C[5];               /* cumulativeFreq */
    uint32_t *bwt;                /* 2-bit chars, 16 per word, MSB first; >= 8 zero words of tail padding */
    size_t    bwtWords;
    uint32_t *occ;                /* GPU-2BWT: (ceil(n/128)+1) x 4, Occ(c,128x)+C[c]  (BGS-Build.cpp:139-161) */
    size_t    occRecs;
} s3o_bwt;

typedef struct {
    s3o_bwt   fwd, rev;
    uint32_t *sa;                 /* full SA, n+1 entries, sa[0] = 0xFFFFFFFF (BWT.c:286) */
    uint32_t *pac;                /* text, 16 chars/word MSB first, zero padded (TextConverter.c:666-720) */
    size_t    pacWords;
    uint32_t  dnaLength;
    uint32_t *lkt, *rlkt;         /* 4^13 cumulative 13-mer counts (2bwt-flex/LT.c) */
    int       ltSize;
} s3o_index;

/* one emitted record, in the CPU path's emission order */
typedef struct {
    uint32_t a, b;                /* kind 0: SA range [a,b];  kind 1: text position a (b == a) */
    uint8_t  kind, strand, nmis, cas;   /* strand 1 = '+', 2 = '-' (QUERY_POS_STRAND/NEG) */
} s3o_hit;

typedef struct {
    uint64_t rankQueries;         /* all-symbol rank queries issued (2 per search step) */
    uint64_t ltLookups;           /* lookup-table probes */
    uint64_t saReads;             /* SA[] reads (CE candidates + final decode) */
    uint64_t ceWords;             /* packed-text 64-bit windows compared in CE */
    uint64_t hits;                /* records emitted */
} s3o_stats;

s3o_index *s3o_index_load(const char *prefix);       /* <prefix>.bwt, .fmv.gpu, .rev.*, .sa, .pac, .lkt, .rev.lkt */
/* same index from caller arrays (copied): payloads of the files, headers stripped */
s3o_index *s3o_index_create(uint32_t n, const uint32_t *bwt, size_t bwtWords, const uint32_t *occ, uint32_t inverseSa0, const uint32_t cum[5],
                            const uint32_t *rbwt, size_t rbwtWords, const uint32_t *rocc, uint32_t rInverseSa0, const uint32_t rcum[5],
                            const uint32_t *sa, const uint32_t *pacWordsMsb, size_t pacWords,
                            const uint32_t *lkt, const uint32_t *rlkt, int ltSize);
void       s3o_index_free(s3o_index *);

void s3o_model_build(int readLength, int maxError, s3o_model *out);

/* Search one read (codes 0..3, forward orientation) for <= mm mismatches through all cases,
 * + strand then - strand per case (CPUfunctions.cpp:555-619).  Returns number of records written
 * (at most cap); total/withError mirror SRAQueryResultCount. */
int s3o_search_read(const s3o_index *ix, const s3o_model *m, const uint8_t *read, int L,
                    uint32_t maxOutputPerRead, s3o_hit *hits, int cap,
                    uint32_t *totalOcc, uint32_t withError[8], s3o_stats *st);

/* Batch over the reference's packed layout (32-interleaved words, chars LSB first; DV-Kernel.cu:4268).
 * hitOffsets has n+1 entries.  Returns total records, or -1 if hits capacity is exceeded. */
long s3o_search_batch(const s3o_index *ix, const uint32_t *queries, const uint32_t *readLengths, uint32_t n,
                      uint32_t wordPerQuery, int mm, uint32_t maxOutputPerRead,
                      s3o_hit *hits, size_t hitCap, uint64_t *hitOffsets,
                      uint32_t *totalOcc, uint32_t *withError /* n x 8 */, s3o_stats *st);

/* decode records to occurrences in the reference's final order: CE hits first, then SA ranges
 * expanded through SA[] (SAList.cpp:392-418).  Returns count written. */
int s3o_decode(const s3o_index *ix, const s3o_hit *hits, int nHits, uint32_t *pos, uint8_t *strand, uint8_t *nmis, int cap,
               s3o_stats *st);

/* Single-end all-best ladder (all_best_single_alignment, alignment.cu:1422-1465, with the per-read decisions of
 * hostKernel, CPUfunctions.cpp:1744-1801,1848-1867): phases mm = 1..numMismatch; a read whose best hit has exactly mm
 * mismatches is deferred to the next phase, or, in the terminal phase (mm < 4), re-searched with the (mm+1)-mismatch
 * model (ProcessOneMoreMismatchAllCases, CPUfunctions.cpp:624).  phase[r] = mm in which the read was reported (0 = none).
 * Occurrences are in the final order of the CPU path.  Returns total occurrences or -1 on overflow of occCap. */
long s3o_align_single_allbest(const s3o_index *ix, const uint32_t *queries, const uint32_t *readLengths, uint32_t n,
                              uint32_t wordPerQuery, int numMismatch, uint32_t maxOutputPerRead,
                              uint8_t *phase, uint32_t *withError /* n x 8 */, uint64_t *occOffsets /* n+1 */,
                              uint32_t *pos, uint8_t *strand, uint8_t *nmis, size_t occCap, s3o_stats *st);

/* every report type of the -h option (1 all valid, 2 all best, 3 unique best, 4 random best); see s3_oracle.c */
long s3o_align_single(const s3o_index *ix, const uint32_t *queries, const uint32_t *readLengths, uint32_t n,
                      uint32_t wordPerQuery, int numMismatch, uint32_t maxOutputPerRead, int reportType,
                      uint8_t *phase, uint32_t *withError /* n x 8 */, uint64_t *occOffsets /* n+1 */,
                      uint32_t *pos, uint8_t *strand, uint8_t *nmis, size_t occCap, s3o_stats *st);

/* the same with the reference's long-read mode: seedLengths[r] bases are searched, the rest verified (NULL = off) */
long s3o_align_single_long(const s3o_index *ix, const uint32_t *queries, const uint32_t *readLengths, const uint32_t *seedLengths, uint32_t n,
                           uint32_t wordPerQuery, int numMismatch, uint32_t maxOutputPerRead, int reportType,
                           uint8_t *phase, uint32_t *withError /* n x 8 */, uint64_t *occOffsets /* n+1 */,
                           uint32_t *pos, uint8_t *strand, uint8_t *nmis, size_t occCap, s3o_stats *st);

/* all-symbol rank on the GPU-2BWT layout (DV-Kernel.cu:85-153,282-299 / BWT.c:1211) : v[c] = Occ(c, i) + C[c] */
void s3o_rank_all(const s3o_bwt *b, uint32_t i, uint32_t v[4]);

#ifdef __cplusplus
}
#endif
#endif
