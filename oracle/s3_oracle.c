/* s3_oracle.c -- TEST INFRASTRUCTURE ONLY (see s3_oracle.h).
 *
 * Plain-C restatement of the reference CPU path for 2BWT exact / k-mismatch search.
 * Every function cites the reference lines it follows.  Written from the behaviour of the
 * reference, not copied: one bidirectional state (fl,fr,rl,rr) and one recursive walker
 * replace the reference's eight Exact/Mismatch x Backward/AnyDirection functions.
 */
#include "s3_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------------------------------
 * Model tables: SRAMismatchCasesPopulate (2bwt-flex/SRA2BWTMdl.c:1171-1700) for MaxError 0..4,
 * build mode COVER_ALL, and the post-compute of SRAModelPopulate (:4520-4602).
 * ------------------------------------------------------------------------------------------ */
enum { A_S, A_E, B_S, B_E, C_S, C_E, D_S, D_E, E_S, E_E, R_0, R_L1 };   /* region references */
enum { CE_NONE, CE_S_BEND_M9, CE_E_CSTART_P9, CE_S_AEND17, CE_E_BLEN17 };

typedef struct { int type, s, e, mn, mx, ce, mod; } stepdef;
typedef struct { int id, ns; stepdef st[S3O_MAX_STEPS]; } casedef;

#define BL S3O_STEP_BACK_LOOKUP
#define FL S3O_STEP_FWD_LOOKUP
#define BO S3O_STEP_BACK_ONLY
#define BI S3O_STEP_BIDIR

static const casedef M0[] = {
    {1, 1, {{BL, R_L1, R_0, 0, 0, 5, 0}}},
};
static const casedef M1[] = {       /* | .5 | .5 |  SRA2BWTMdl.c:1238-1278 */
    {11, 2, {{BL, B_E, B_S, 0, 0, 0, 0}, {BO, A_E, A_S, 0, 1, 20, 0}}},
    {12, 2, {{FL, A_S, A_E, 0, 0, 0, 0}, {BI, B_S, B_E, 1, 1, 20, 0}}},
};
static const casedef M2[] = {       /* | .3 | .3 | .4 |  :1280-1373 */
    {21, 2, {{BL, C_E, C_S, 0, 0, 0, 0}, {BO, B_E, A_S, 0, 2, 20, CE_S_BEND_M9}}},
    {22, 2, {{FL, R_0, B_E, 0, 0, 0, 0}, {BI, C_S, C_E, 1, 2, 20, CE_E_CSTART_P9}}},
    {23, 3, {{FL, R_0, A_E, 0, 0, 0, 0}, {BI, B_S, B_E, 1, 1, 10, CE_S_AEND17}, {BI, C_S, C_E, 1, 1, 10, CE_E_CSTART_P9}}},
    {24, 3, {{FL, B_S, B_E, 0, 0, 0, 0}, {BI, C_S, C_E, 1, 1, 10, CE_E_BLEN17}, {BO, A_E, A_S, 1, 1, 10, CE_E_CSTART_P9}}},
};
static const casedef M3[] = {       /* | .25 x 4 |  :1375-1520 */
    {31, 2, {{BL, D_E, C_S, 0, 0, 0, 0}, {BO, B_E, A_S, 0, 3, 20, 0}}},
    {32, 2, {{FL, A_S, B_E, 0, 0, 0, 0}, {BI, C_S, D_E, 1, 3, 20, 0}}},
    {33, 3, {{FL, A_S, A_E, 0, 0, 0, 0}, {BI, B_S, B_E, 1, 1, 20, 0}, {BI, C_S, D_E, 2, 2, 20, 0}}},
    {35, 3, {{FL, C_S, C_E, 0, 0, 0, 0}, {BI, D_S, D_E, 1, 1, 20, 0}, {BO, B_E, A_S, 1, 2, 20, 0}}},
    {36, 3, {{BL, D_E, D_S, 0, 0, 0, 0}, {BO, C_E, C_S, 1, 1, 20, 0}, {BO, B_E, A_S, 1, 2, 20, 0}}},
    {34, 3, {{FL, B_S, B_E, 0, 0, 0, 0}, {BI, A_E, A_S, 1, 1, 20, 0}, {BI, C_S, D_E, 2, 2, 20, 0}}},
};
static const casedef M4[] = {       /* | .2 x 5 |  :1522-1700 */
    {41, 2, {{BL, E_E, D_S, 0, 0, 0, 0}, {BO, C_E, A_S, 0, 4, 20, 0}}},
    {42, 2, {{FL, A_S, C_E, 0, 0, 0, 0}, {BI, D_S, E_E, 1, 4, 20, 0}}},
    {43, 3, {{FL, A_S, A_E, 0, 0, 0, 0}, {BI, B_S, C_E, 1, 1, 20, 0}, {BI, D_S, E_E, 1, 3, 20, 0}}},
    {44, 3, {{FL, B_S, C_E, 0, 0, 0, 0}, {BI, A_E, A_S, 1, 1, 20, 0}, {BI, D_S, E_E, 1, 3, 20, 0}}},
    {45, 3, {{FL, A_S, A_E, 0, 0, 0, 0}, {BI, B_S, C_E, 2, 2, 20, 0}, {BI, D_S, E_E, 1, 2, 20, 0}}},
    {46, 3, {{FL, B_S, C_E, 0, 0, 0, 0}, {BI, A_E, A_S, 2, 2, 20, 0}, {BI, D_S, E_E, 1, 2, 20, 0}}},
    {47, 4, {{FL, B_S, B_E, 0, 0, 0, 0}, {BI, A_E, A_S, 1, 1, 20, 0}, {BI, C_S, C_E, 1, 1, 20, 0}, {BI, D_S, E_E, 1, 2, 20, 0}}},
    {48, 4, {{FL, C_S, C_E, 0, 0, 0, 0}, {BI, B_E, B_S, 1, 1, 20, 0}, {BI, A_E, A_S, 1, 1, 20, 0}, {BI, D_S, E_E, 1, 2, 20, 0}}},
    {49, 3, {{FL, D_S, D_E, 0, 0, 0, 0}, {BI, E_S, E_E, 1, 1, 20, 0}, {BO, C_E, A_S, 3, 3, 20, 0}}},
    {410, 3, {{BL, E_E, E_S, 0, 0, 0, 0}, {BO, D_E, D_S, 1, 1, 20, 0}, {BO, C_E, A_S, 3, 3, 20, 0}}},
};
static const casedef *const MDEF[5] = {M0, M1, M2, M3, M4};
static const int MCASES[5] = {1, 2, 4, 6, 10};      /* definitions.h:116-120 */

/* cell boundaries: float arithmetic exactly as SRA2BWTMdl.c:1177-1203 (unsigned * float, then -1) */
static void regions(unsigned int L, int maxError, unsigned int r[12])
{
    static const float frac[5] = {1.0f, 0.5f, 0.3f, 0.25f, 0.2f};
    int cells = maxError + 1, k;
    float f = frac[maxError];
    memset(r, 0, 12 * sizeof(unsigned int));
    r[R_0] = 0; r[R_L1] = L - 1;
    if (maxError == 0) { r[A_S] = 0; r[A_E] = L - 1; return; }
    r[A_S] = 0;
    r[A_E] = (L * f) - 1;
    for (k = 1; k < cells; k++) {
        r[2 * k] = r[2 * k - 1] + 1;
        if (k == cells - 1) r[2 * k + 1] = L - 1;
        else r[2 * k + 1] = r[2 * k] + (L * f) - 1;
    }
}

void s3o_model_build(int readLength, int maxError, s3o_model *m)
{
    unsigned int r[12];
    int c, s, l;
    unsigned int L = (unsigned int)readLength;
    memset(m, 0, sizeof(*m));
    m->readLength = readLength; m->maxError = maxError; m->nCases = MCASES[maxError];
    regions(L, maxError, r);
    for (c = 0; c < m->nCases; c++) {
        const casedef *cd = &MDEF[maxError][c];
        s3o_case *cs = &m->cases[c];
        int leftMost = -1;
        cs->id = cd->id; cs->nSteps = cd->ns;
        for (s = 0; s < cd->ns; s++) {
            const stepdef *sd = &cd->st[s];
            s3o_step *st = &cs->steps[s];
            st->type = sd->type;
            st->start = (int)r[sd->s]; st->end = (int)r[sd->e];
            st->minErr = sd->mn; st->maxErr = sd->mx;
            st->ceThreshold = sd->ce;
            st->ceStart = 0; st->ceEnd = readLength - 1;            /* defaults, :4458-4459 */
            switch (sd->mod) {
            case CE_S_BEND_M9:   st->ceStart = (int)(r[B_E] - 9); break;                       /* :1303 */
            case CE_E_CSTART_P9: st->ceEnd = (int)(r[C_S] + 9); break;                         /* :1322,1348,1370 */
            case CE_S_AEND17:    st->ceStart = (r[A_E] >= 17) ? 0 : 17; break;                 /* :1339 */
            case CE_E_BLEN17:    st->ceEnd = (r[B_E] - r[B_S] + 1 >= 17) ? readLength - 1 : (int)(r[B_E] - 18); break;   /* :1362 */
            default: break;
            }
            /* post-compute, :4529-4580 */
            st->dir = (st->start > st->end) ? -1 : 1;
            st->len = (st->end - st->start) * st->dir + 1;
            if (st->dir > 0) {
                if (leftMost == -1) leftMost = st->start;
                for (l = st->start; l <= st->end; l++) cs->leftMost[l] = (int16_t)leftMost;
            } else {
                if (leftMost == -1) leftMost = st->start + 1;
                for (l = st->start; l >= st->end; l--) cs->leftMost[l] = (int16_t)(leftMost--);
            }
            if (st->ceThreshold > 32) st->ceThreshold = 32;          /* MAX_CE_THRESHOLD */
            if (st->ceStart > st->ceEnd) { int t = st->ceEnd; st->ceEnd = st->ceStart; st->ceStart = t; }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Index files (formats: SURVEY.md 5.9; readers BWT.c:169-288, IndexHandler.cpp:158-199,
 * TextConverter.c:666-720, LT.c:37-57)
 * ------------------------------------------------------------------------------------------ */
static void *slurp(const char *name, size_t skipBytes, size_t *bytesOut, size_t padBytes)
{
    FILE *f = fopen(name, "rb");
    long sz;
    char *buf;
    if (!f) { fprintf(stderr, "[s3_oracle] cannot open %s\n", name); return NULL; }
    fseek(f, 0, SEEK_END); sz = ftell(f); fseek(f, (long)skipBytes, SEEK_SET);
    sz -= (long)skipBytes;
    buf = (char *)calloc((size_t)sz + padBytes + 16, 1);
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fprintf(stderr, "[s3_oracle] short read %s\n", name); }
    fclose(f);
    if (bytesOut) *bytesOut = (size_t)sz;
    return buf;
}

static int load_half(const char *prefix, const char *mid, s3o_bwt *b)
{
    char name[4096];
    uint32_t hdr[5];
    size_t bytes;
    FILE *f;
    snprintf(name, sizeof name, "%s%s.bwt", prefix, mid);
    f = fopen(name, "rb");
    if (!f) { fprintf(stderr, "[s3_oracle] cannot open %s\n", name); return -1; }
    if (fread(hdr, 4, 5, f) != 5) { fclose(f); return -1; }
    fclose(f);
    b->inverseSa0 = hdr[0];
    b->C[0] = 0; b->C[1] = hdr[1]; b->C[2] = hdr[2]; b->C[3] = hdr[3]; b->C[4] = hdr[4];
    b->n = hdr[4];
    b->bwt = (uint32_t *)slurp(name, 20, &bytes, 64 * 4);
    b->bwtWords = bytes / 4;
    snprintf(name, sizeof name, "%s%s.fmv.gpu", prefix, mid);
    b->occ = (uint32_t *)slurp(name, 20, &bytes, 0);
    b->occRecs = bytes / 16;
    if (!b->bwt || !b->occ) return -1;
    if (b->occRecs != (size_t)(b->n + 127) / 128 + 1) { fprintf(stderr, "[s3_oracle] occ size mismatch\n"); return -1; }
    return 0;
}

s3o_index *s3o_index_load(const char *prefix)
{
    s3o_index *ix = (s3o_index *)calloc(1, sizeof(*ix));
    char name[4096];
    size_t bytes, i;
    unsigned char *raw;
    if (load_half(prefix, "", &ix->fwd) || load_half(prefix, ".rev", &ix->rev)) { free(ix); return NULL; }
    snprintf(name, sizeof name, "%s.sa", prefix);
    ix->sa = (uint32_t *)slurp(name, 24, &bytes, 0);            /* 5-word header + saInterval */
    if (!ix->sa) { free(ix); return NULL; }
    ix->sa[0] = 0xFFFFFFFFu;                                    /* BWT.c:286 */
    snprintf(name, sizeof name, "%s.pac", prefix);
    raw = (unsigned char *)slurp(name, 0, &bytes, 64);
    if (!raw) { free(ix); return NULL; }
    ix->dnaLength = (uint32_t)((bytes - 2) * 4 + raw[bytes - 1]);
    raw[bytes - 1] = 0;                                         /* the length byte is not text */
    ix->pacWords = (ix->dnaLength + 63) / 64 * 4 + 8;
    ix->pac = (uint32_t *)calloc(ix->pacWords + 4, 4);
    for (i = 0; i < ix->pacWords && i * 4 < bytes - 1; i++)
        ix->pac[i] = ((uint32_t)raw[4 * i] << 24) | ((uint32_t)raw[4 * i + 1] << 16) | ((uint32_t)raw[4 * i + 2] << 8) | raw[4 * i + 3];
    free(raw);
    snprintf(name, sizeof name, "%s.lkt", prefix);
    {
        FILE *f = fopen(name, "rb");
        int32_t ts = 0;
        if (f) {
            if (fread(&ts, 4, 1, f) != 1) ts = 0;
            fclose(f);
        }
        ix->ltSize = ts;
        if (ts > 0) {
            ix->lkt = (uint32_t *)slurp(name, 4, &bytes, 0);
            snprintf(name, sizeof name, "%s.rev.lkt", prefix);
            ix->rlkt = (uint32_t *)slurp(name, 4, &bytes, 0);
        }
        if (!ix->lkt || !ix->rlkt) { fprintf(stderr, "[s3_oracle] lookup tables missing\n"); s3o_index_free(ix); return NULL; }
    }
    return ix;
}

static void *dupw(const uint32_t *src, size_t words, size_t padWords)
{
    uint32_t *p = (uint32_t *)calloc(words + padWords + 4, 4);
    memcpy(p, src, words * 4);
    return p;
}

s3o_index *s3o_index_create(uint32_t n, const uint32_t *bwt, size_t bwtWords, const uint32_t *occ, uint32_t inverseSa0, const uint32_t cum[5],
                            const uint32_t *rbwt, size_t rbwtWords, const uint32_t *rocc, uint32_t rInverseSa0, const uint32_t rcum[5],
                            const uint32_t *sa, const uint32_t *pacWordsMsb, size_t pacWords,
                            const uint32_t *lkt, const uint32_t *rlkt, int ltSize)
{
    s3o_index *ix = (s3o_index *)calloc(1, sizeof(*ix));
    size_t recs = ((size_t)n + 127) / 128 + 1, lt = (size_t)1 << (2 * ltSize);
    ix->fwd.n = ix->rev.n = n;
    ix->fwd.inverseSa0 = inverseSa0; ix->rev.inverseSa0 = rInverseSa0;
    memcpy(ix->fwd.C, cum, 20); memcpy(ix->rev.C, rcum, 20);
    ix->fwd.bwt = (uint32_t *)dupw(bwt, bwtWords, 64); ix->fwd.bwtWords = bwtWords;
    ix->rev.bwt = (uint32_t *)dupw(rbwt, rbwtWords, 64); ix->rev.bwtWords = rbwtWords;
    ix->fwd.occ = (uint32_t *)dupw(occ, recs * 4, 0); ix->fwd.occRecs = recs;
    ix->rev.occ = (uint32_t *)dupw(rocc, recs * 4, 0); ix->rev.occRecs = recs;
    ix->sa = (uint32_t *)dupw(sa, (size_t)n + 1, 0);
    ix->sa[0] = 0xFFFFFFFFu;
    ix->pac = (uint32_t *)dupw(pacWordsMsb, pacWords, 8); ix->pacWords = pacWords + 4;
    ix->dnaLength = n;
    ix->lkt = (uint32_t *)dupw(lkt, lt, 0); ix->rlkt = (uint32_t *)dupw(rlkt, lt, 0); ix->ltSize = ltSize;
    return ix;
}

void s3o_index_free(s3o_index *ix)
{
    if (!ix) return;
    free(ix->fwd.bwt); free(ix->fwd.occ); free(ix->rev.bwt); free(ix->rev.occ);
    free(ix->sa); free(ix->pac); free(ix->lkt); free(ix->rlkt); free(ix);
}

/* ------------------------------------------------------------------------------------------
 * Rank: GPUBWTAllOccValue / GPUDNAAllOccCount (DV-Kernel.cu:85-153,282-299); same values as the
 * CPU's BWTAllOccValue (BWT.c:1211-1268) + cumulativeFreq.
 * ------------------------------------------------------------------------------------------ */
static inline void count4(uint64_t w, uint64_t mask, uint32_t cnt[4])
{
    uint64_t lo = w & 0x5555555555555555ULL, hi = (w >> 1) & 0x5555555555555555ULL;
    cnt[0] += (uint32_t)__builtin_popcountll(~hi & ~lo & mask);
    cnt[1] += (uint32_t)__builtin_popcountll(~hi & lo & mask);
    cnt[2] += (uint32_t)__builtin_popcountll(hi & ~lo & mask);
    cnt[3] += (uint32_t)__builtin_popcountll(hi & lo & mask);
}

void s3o_rank_all(const s3o_bwt *b, uint32_t i, uint32_t v[4])
{
    uint32_t k, p, d, cnt[4] = {0, 0, 0, 0};
    const uint32_t *w;
    uint64_t w0, w1;
    const uint64_t M5 = 0x5555555555555555ULL;
    i -= (i > b->inverseSa0);
    k = (i + 63) / 128; p = k * 128;
    v[0] = b->occ[4 * k]; v[1] = b->occ[4 * k + 1]; v[2] = b->occ[4 * k + 2]; v[3] = b->occ[4 * k + 3];
    if (i == p) return;
    if (i > p) {
        d = i - p;                                   /* first d chars of the 64 after the sample */
        w = b->bwt + p / 16;
        w0 = ((uint64_t)w[0] << 32) | w[1]; w1 = ((uint64_t)w[2] << 32) | w[3];
        if (d >= 32) { count4(w0, M5, cnt); if (d > 32) count4(w1, M5 & (~0ULL << (64 - 2 * (d - 32))), cnt); }
        else count4(w0, M5 & (~0ULL << (64 - 2 * d)), cnt);
        v[0] += cnt[0]; v[1] += cnt[1]; v[2] += cnt[2]; v[3] += cnt[3];
    } else {
        d = p - i;                                   /* last d chars of the 64 before the sample */
        w = b->bwt + p / 16 - 4;
        w0 = ((uint64_t)w[0] << 32) | w[1]; w1 = ((uint64_t)w[2] << 32) | w[3];
        if (d >= 32) { count4(w1, M5, cnt); if (d > 32) count4(w0, M5 & ~(~0ULL << (2 * (d - 32))), cnt); }
        else count4(w1, M5 & ~(~0ULL << (2 * d)), cnt);
        v[0] -= cnt[0]; v[1] -= cnt[1]; v[2] -= cnt[2]; v[3] -= cnt[3];
    }
}

/* ------------------------------------------------------------------------------------------
 * Search
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const s3o_index *ix;
    const s3o_case *cs;
    const uint8_t *read;            /* oriented read (already reverse-complemented for '-') */
    int L, strand, caseIdx;
    uint32_t maxOut;
    uint32_t *total, *withError;
    s3o_hit *hits; int cap, nHits, overflow;
    s3o_stats *st;
} ctx_t;

typedef struct { uint32_t fl, fr, rl, rr; } rng_t;     /* forward-index range, reverse-index range */

static void emit(ctx_t *c, int kind, uint32_t a, uint32_t b, int nmis)
{
    if (c->nHits < c->cap) {
        s3o_hit *h = &c->hits[c->nHits];
        h->a = a; h->b = b; h->kind = (uint8_t)kind; h->strand = (uint8_t)c->strand; h->nmis = (uint8_t)nmis; h->cas = (uint8_t)c->caseIdx;
    } else c->overflow = 1;
    c->nHits++;
    if (c->st) c->st->hits++;
}

/* 32 text chars starting at pos, MSB first (chars beyond the text are the zero padding) */
static inline uint64_t text64(const s3o_index *ix, uint64_t pos)
{
    uint64_t idx = pos / 16; int sh = (int)(pos % 16);
    uint64_t w;
    if (idx + 2 >= ix->pacWords + 4) return 0;
    w = ((uint64_t)ix->pac[idx] << 32) | ix->pac[idx + 1];
    if (sh) { w <<= 2 * sh; w |= ix->pac[idx + 2] >> (2 * (16 - sh)); }
    return w;
}

/* mismatches between read[left..left+len) and text[tpos..tpos+len)   (CEPackedMismatchMatchingWithQuality,
 * SRA2BWTCheckAndExtend.c:142-222: XOR of 2-bit packed windows, fold, popcount) */
static int ce_mismatches(ctx_t *c, uint32_t tpos, int left, int len)
{
    int mm = 0, o = 0;
    while (o < len) {
        int chunk = len - o > 32 ? 32 : len - o, j;
        uint64_t key = 0, t = text64(c->ix, (uint64_t)tpos + (uint64_t)o), x;
        for (j = 0; j < chunk; j++) key = (key << 2) | c->read[left + o + j];
        if (chunk < 32) t >>= 64 - 2 * chunk;
        x = t ^ key;
        x = (x | (x >> 1)) & 0x5555555555555555ULL;
        mm += __builtin_popcountll(x);
        if (c->st) c->st->ceWords++;
        o += chunk;
    }
    return mm;
}

/* BWTMismatchModelAnyDirection_CE2, BGS-HostAlgnmtAlgo2.cpp:36-176 */
static void check_and_extend(ctx_t *c, int s, int i, int inserted, uint32_t l, uint32_t r, int occMis)
{
    const s3o_case *cs = c->cs;
    uint32_t pos32[32];
    int mis[32], n = 0, k, j;
    int p0 = cs->steps[s].start + cs->steps[s].dir * i;
    int leftAlign = cs->leftMost[p0];
    uint32_t q;
    if (*c->total >= c->maxOut) return;
    for (q = l; q <= r && n < 32; q++) {
        pos32[n] = c->ix->sa[q] - (uint32_t)leftAlign;        /* 32-bit wrap as in the reference */
        mis[n] = occMis; n++;
        if (c->st) c->st->saReads++;
    }
    for (; s < cs->nSteps; s++) {
        const s3o_step *st = &cs->steps[s];
        int pos = st->start + st->dir * i;
        int left = pos < st->end ? pos : st->end, right = pos < st->end ? st->end : pos;
        int len = right - left + 1, m = 0;
        for (j = 0; j < n; j++) {
            uint32_t rp = pos32[j];
            int mce, tot;
            if ((uint32_t)(rp + (uint32_t)left) > c->ix->fwd.n) mce = 16;     /* MAX_NUM_OF_ERROR+1, :122-125 */
            else mce = ce_mismatches(c, rp + (uint32_t)left, left, len);
            tot = mce + inserted;
            if (tot >= st->minErr && tot <= st->maxErr) { pos32[m] = rp; mis[m] = mis[j] + tot; m++; }
        }
        n = m; i = 0; inserted = 0;
    }
    if (*c->total + (uint32_t)n > c->maxOut) n = (int)(c->maxOut - *c->total);
    for (k = 0; k < n; k++) { emit(c, 1, pos32[k], pos32[k], mis[k]); c->withError[mis[k] & 7]++; }
    *c->total += (uint32_t)n;
}

/* one BWT step on character ch; returns the oCount-style passive update (BGS-HostAlgnmtAlgo2.cpp:994-1048) */
static inline void extend_all(ctx_t *c, int dir, const rng_t *g, uint32_t oL[4], uint32_t oR[4])
{
    const s3o_bwt *b = dir < 0 ? &c->ix->fwd : &c->ix->rev;
    uint32_t al = dir < 0 ? g->fl : g->rl, ar = dir < 0 ? g->fr : g->rr;
    s3o_rank_all(b, al, oL);
    s3o_rank_all(b, ar + 1, oR);
    if (c->st) c->st->rankQueries += 2;
}

static inline void child(int dir, const rng_t *g, const uint32_t oL[4], const uint32_t oR[4], int ch, rng_t *out)
{
    /* active range moves by rank; passive range shrinks by the count of larger symbols (oCount) */
    uint32_t al = oL[ch] + 1, ar = oR[ch], oc = 0, pr, pl;
    int k;
    for (k = 3; k > ch; k--) oc += oR[k] - oL[k];
    pr = (dir < 0 ? g->rr : g->fr) - oc;
    pl = pr - (ar - al);
    if (dir < 0) { out->fl = al; out->fr = ar; out->rl = pl; out->rr = pr; }
    else { out->rl = al; out->rr = ar; out->fl = pl; out->fr = pr; }
}

/* BWTModelSwitch{AnyDirection,Backward}2 + Exact/Mismatch models, BGS-HostAlgnmtAlgo2.cpp:725-1181,1582-1728 */
static void walk(ctx_t *c, int s, int i, int inserted, rng_t g, int occMis)
{
    const s3o_case *cs = c->cs;
    const s3o_step *st;
    int len, pos, exact;
    if (*c->total >= c->maxOut) return;
    if (s >= cs->nSteps) {                                     /* SRA_STEP_TYPE_COMPLETE, :1603-1625 */
        uint32_t l = g.fl, r = g.fr;
        if (r >= l) {
            if (*c->total + (r - l + 1) > c->maxOut) r = l + c->maxOut - *c->total - 1;
            emit(c, 0, l, r, inserted + occMis);
            *c->total += r - l + 1;
            c->withError[(inserted + occMis) & 7] += r - l + 1;
        }
        return;
    }
    st = &cs->steps[s];
    exact = (inserted >= st->maxErr);                          /* :1639 (lookup steps never come here) */
    len = st->len;
    if (!exact && st->minErr > inserted + 1) len -= st->minErr - inserted - 1;     /* :976,1109 */
    pos = st->start + st->dir * i;
    while (i < len && g.fl <= g.fr) {
        uint32_t oL[4], oR[4];
        int ch, k;
        if ((uint32_t)(g.fr - g.fl) < (uint32_t)st->ceThreshold && pos >= st->ceStart && pos <= st->ceEnd) {
            check_and_extend(c, s, i, inserted, g.fl, g.fr, occMis);
            return;
        }
        ch = c->read[pos];
        extend_all(c, st->dir, &g, oL, oR);
        if (!exact) {
            for (k = 0; k < 4; k++) {
                rng_t e;
                if (k == ch) continue;
                child(st->dir, &g, oL, oR, k, &e);
                if (e.fl <= e.fr) walk(c, s, i + 1, inserted + 1, e, occMis);
            }
        }
        child(st->dir, &g, oL, oR, ch, &g);
        i++; pos += st->dir;
    }
    if (inserted >= st->minErr && g.fl <= g.fr) walk(c, s + 1, 0, 0, g, occMis + inserted);
}

static uint32_t lt_range(const uint32_t *tab, int tableSize, const uint8_t *chars, int n, int stride, uint32_t *rOut, s3o_stats *stt)
{
    /* BGS-HostAlgnmtAlgo2.cpp:311-327 : pack n chars, pad low with 0s (l) / 1s (r) */
    uint64_t lp = 0, rp;
    int i;
    for (i = 0; i < n; i++) lp = (lp << 2) | (chars[i * stride] & 3);
    rp = lp;
    for (i = n; i < tableSize; i++) { lp <<= 2; rp = (rp << 2) | 3; }
    if (stt) stt->ltLookups += 2;
    *rOut = tab[rp];
    return lp ? tab[lp - 1] + 1 : 1;
}

/* BWTExactModelBackward_Lookup2 (:437-548) and BWTExactModelForward_Lookup2 (:270-433) */
static void run_case(ctx_t *c)
{
    const s3o_case *cs = c->cs;
    const s3o_step *st = &cs->steps[0];
    const s3o_index *ix = c->ix;
    int lookupLength = ix->ltSize > st->len ? st->len : ix->ltSize;
    rng_t g;
    int i, pos;
    g.fl = 0; g.fr = ix->fwd.n; g.rl = 0; g.rr = ix->fwd.n;
    if (st->type == S3O_STEP_BACK_LOOKUP) {
        g.fl = lt_range(ix->lkt, ix->ltSize, c->read + st->start - lookupLength + 1, lookupLength, 1, &g.fr, c->st);
        /* the reverse range is not tracked by the backward-only chain; keep sizes consistent */
        g.rr = g.fr; g.rl = g.fl;
        i = lookupLength; pos = st->start - lookupLength;
    } else {
        uint32_t l, r, rl, rr;
        l = lt_range(ix->lkt, ix->ltSize, c->read + st->start, lookupLength, 1, &r, c->st);
        rl = lt_range(ix->rlkt, ix->ltSize, c->read + st->start + lookupLength - 1, lookupLength, -1, &rr, c->st);
        if (r - l < rr - rl) rl = rr - (r - l);                /* :349-356 */
        else if (r - l > rr - rl) l = r - (rr - rl);
        g.fl = l; g.fr = r; g.rl = rl; g.rr = rr;
        i = lookupLength; pos = st->start + lookupLength;
    }
    while (i < st->len && g.fl <= g.fr) {
        uint32_t oL[4], oR[4];
        if ((uint32_t)(g.fr - g.fl) < (uint32_t)st->ceThreshold && pos >= st->ceStart && pos <= st->ceEnd) {
            check_and_extend(c, 0, i, 0, g.fl, g.fr, 0);
            return;
        }
        extend_all(c, st->dir, &g, oL, oR);
        child(st->dir, &g, oL, oR, c->read[pos], &g);
        i++; pos += st->dir;
    }
    if (g.fl <= g.fr) walk(c, 1, 0, 0, g, 0);
}

int s3o_search_read(const s3o_index *ix, const s3o_model *m, const uint8_t *read, int L,
                    uint32_t maxOutputPerRead, s3o_hit *hits, int cap,
                    uint32_t *totalOcc, uint32_t withError[8], s3o_stats *st)
{
    uint8_t rc[S3O_MAX_READ];
    ctx_t c;
    int cas, i;
    for (i = 0; i < L; i++) rc[i] = (uint8_t)(3 - read[L - 1 - i]);       /* soap3DnaComplement, SRACore.h:44 */
    memset(&c, 0, sizeof c);
    c.ix = ix; c.L = L; c.maxOut = maxOutputPerRead; c.total = totalOcc; c.withError = withError;
    c.hits = hits; c.cap = cap; c.st = st;
    *totalOcc = 0; memset(withError, 0, 8 * sizeof(uint32_t));
    for (cas = 0; cas < m->nCases; cas++) {                     /* ProcessReadDoubleStrand2, CPUfunctions.cpp:555-619 */
        c.cs = &m->cases[cas]; c.caseIdx = cas;
        c.read = read; c.strand = 1; run_case(&c);
        c.read = rc;   c.strand = 2; run_case(&c);
    }
    return c.overflow ? -c.nHits : c.nHits;
}

static void unpack_read(const uint32_t *queries, uint32_t r, uint32_t wpq, int L, uint8_t *out)
{
    const uint32_t *p = queries + (size_t)(r / 32) * 32 * wpq + r % 32;      /* retrieve2BWTQueryFormat, CPUfunctions.cpp:1093 */
    int i;
    for (i = 0; i < L; i++) out[i] = (uint8_t)((p[(size_t)(i / 16) * 32] >> (2 * (i % 16))) & 3);
}

long s3o_search_batch(const s3o_index *ix, const uint32_t *queries, const uint32_t *readLengths, uint32_t n,
                      uint32_t wordPerQuery, int mm, uint32_t maxOutputPerRead,
                      s3o_hit *hits, size_t hitCap, uint64_t *hitOffsets,
                      uint32_t *totalOcc, uint32_t *withError, s3o_stats *st)
{
    s3o_model *models[S3O_MAX_READ + 1];
    uint8_t read[S3O_MAX_READ];
    size_t used = 0;
    uint32_t r;
    long rc = 0;
    memset(models, 0, sizeof models);
    for (r = 0; r < n; r++) {
        int L = (int)readLengths[r], k;
        size_t room = hitCap - used;
        if (!models[L]) { models[L] = (s3o_model *)malloc(sizeof(s3o_model)); s3o_model_build(L, mm, models[L]); }
        unpack_read(queries, r, wordPerQuery, L, read);
        hitOffsets[r] = used;
        k = s3o_search_read(ix, models[L], read, L, maxOutputPerRead, hits + used, room > 0x7fffffff ? 0x7fffffff : (int)room,
                            &totalOcc[r], &withError[(size_t)r * 8], st);
        if (k < 0) { rc = -1; break; }
        used += (size_t)k;
    }
    hitOffsets[n] = used;
    for (r = 0; r <= S3O_MAX_READ; r++) free(models[r]);
    return rc < 0 ? rc : (long)used;
}

int s3o_decode(const s3o_index *ix, const s3o_hit *hits, int nHits, uint32_t *pos, uint8_t *strand, uint8_t *nmis, int cap, s3o_stats *st)
{
    int i, n = 0;
    uint32_t k;
    for (i = 0; i < nHits; i++) if (hits[i].kind == 1) { if (n < cap) { pos[n] = hits[i].a; strand[n] = hits[i].strand; nmis[n] = hits[i].nmis; } n++; }
    for (i = 0; i < nHits; i++) if (hits[i].kind == 0)
        for (k = hits[i].a; k <= hits[i].b; k++) {
            if (n < cap) { pos[n] = ix->sa[k]; strand[n] = hits[i].strand; nmis[n] = hits[i].nmis; }
            n++;
            if (st) st->saReads++;
            if (k == 0xFFFFFFFFu) break;
        }
    return n;
}

long s3o_align_single_allbest(const s3o_index *ix, const uint32_t *queries, const uint32_t *readLengths, uint32_t n,
                              uint32_t wordPerQuery, int numMismatch, uint32_t maxOutputPerRead,
                              uint8_t *phase, uint32_t *withError, uint64_t *occOffsets,
                              uint32_t *pos, uint8_t *strand, uint8_t *nmis, size_t occCap, s3o_stats *st)
{
    return s3o_align_single_long(ix, queries, readLengths, NULL, n, wordPerQuery, numMismatch, maxOutputPerRead, 2, phase, withError, occOffsets, pos,
                                 strand, nmis, occCap, st);
}

static unsigned text_char(const s3o_index *ix, uint64_t p) { return (ix->pac[p >> 4] >> (2 * (15 - (p & 15)))) & 3u; }

/* validateAlignments (CPUfunctions.cpp:1129-1222): long-read mode aligned only the first seedLen bases; every hit is extended
 * over the remaining readLen - seedLen bases by counting mismatching characters against the packed text (numMismatchNew,
 * PE.cpp:148), kept when seed + extension mismatches <= maxMismatch, moved to the leftmost text position of the whole read.
 * Returns the new list size (filtered in place, order kept). */
static int validate_long(const s3o_index *ix, const uint8_t *read, int seedLen, int readLen, uint32_t *pos, uint8_t *strand, uint8_t *nmis, int cnt,
                         int onlyBest, int minSeedMis, int maxMis, uint32_t maxHit)
{
    const int ext = readLen - seedLen;
    int i, out = 0, pre = maxMis;
    if (readLen <= seedLen) return cnt;
    for (i = 0; i < cnt; i++) {
        int mis = maxMis + 1, tot, t;
        if ((int)nmis[i] < minSeedMis) continue;
        if (strand[i] == 1 && (uint64_t)pos[i] + (uint32_t)readLen <= ix->dnaLength) {
            for (mis = 0, t = 0; t < ext; t++) mis += read[seedLen + t] != text_char(ix, (uint64_t)pos[i] + seedLen + t);
        }
        if (strand[i] == 2 && pos[i] >= (uint32_t)ext) {
            for (mis = 0, t = 0; t < ext; t++) mis += (3u - read[readLen - 1 - t]) != text_char(ix, (uint64_t)pos[i] - ext + t);
        }
        tot = mis + nmis[i];
        if (tot > maxMis) continue;
        if (onlyBest && tot > pre) continue;
        if (onlyBest && tot < pre) { out = 0; pre = tot; }
        pos[out] = strand[i] == 1 ? pos[i] : pos[i] - (uint32_t)ext;
        strand[out] = strand[i];
        nmis[out] = (uint8_t)tot;
        out++;
        if (pre == minSeedMis && (uint32_t)out >= maxHit) break;
    }
    return out;
}

/* reportType (the -h option): 1 all valid = one terminal phase at numMismatch (alignment.cu:2530); 2 all best (alignment.cu:1422);
 * 3/4 unique / random best = phases 0..numMismatch without the MAPQ special case (best_single_alignment, alignment.cu:1377;
 * needOutputMAPQ is false there, CPUfunctions.cpp:1637); phase[] is then phase + 1. */
long s3o_align_single(const s3o_index *ix, const uint32_t *queries, const uint32_t *readLengths, uint32_t n,
                      uint32_t wordPerQuery, int numMismatch, uint32_t maxOutputPerRead, int reportType,
                      uint8_t *phase, uint32_t *withError, uint64_t *occOffsets,
                      uint32_t *pos, uint8_t *strand, uint8_t *nmis, size_t occCap, s3o_stats *st)
{
    return s3o_align_single_long(ix, queries, readLengths, NULL, n, wordPerQuery, numMismatch, maxOutputPerRead, reportType, phase, withError, occOffsets,
                                 pos, strand, nmis, occCap, st);
}

/* seedLengths != NULL is the reference's long-read mode (maxReadLength > 120, alignment.cu:2473-2491 and hostKernel,
 * CPUfunctions.cpp:1649-1657,1813-1840): the search runs on the first seedLengths[r] bases without a hit limit, the MAPQ special case
 * looks at the seed statistics, then validateAlignments filters; a read whose list comes out empty is "nohit" and moves on. */
long s3o_align_single_long(const s3o_index *ix, const uint32_t *queries, const uint32_t *readLengths, const uint32_t *seedLengths, uint32_t n,
                           uint32_t wordPerQuery, int numMismatch, uint32_t maxOutputPerRead, int reportType,
                           uint8_t *phase, uint32_t *withError, uint64_t *occOffsets,
                           uint32_t *pos, uint8_t *strand, uint8_t *nmis, size_t occCap, s3o_stats *st)
{
    const int bestMode = reportType >= 3, allValid = reportType == 1;
    const int needMapq = !bestMode;                                     /* SAM output, CPUfunctions.cpp:1642 */
    const uint32_t searchLimit = seedLengths ? 0xFFFFFFFFu : maxOutputPerRead;
    static const int HCAP = 1 << 16;
    s3o_model *models[5][S3O_MAX_READ + 1];
    s3o_hit *hits = (s3o_hit *)malloc(sizeof(s3o_hit) * (size_t)HCAP);
    uint8_t read[S3O_MAX_READ];
    size_t used = 0;
    uint32_t r;
    int last = (bestMode || allValid) ? numMismatch : (numMismatch > 0 ? numMismatch : 1), e, overflow = 0;
    const int firstPhase = bestMode ? 0 : (allValid ? numMismatch : 1);
    memset(models, 0, sizeof models);
    for (r = 0; r < n && !overflow; r++) {
        const int fullLen = (int)readLengths[r];
        int L = seedLengths ? (int)seedLengths[r] : fullLen, mm;
        unpack_read(queries, r, wordPerQuery, fullLen, read);
        phase[r] = 0;
        memset(&withError[(size_t)r * 8], 0, 32);
        occOffsets[r] = used;
        for (mm = firstPhase; mm <= last; mm++) {
            uint32_t total, we[8];
            int k, best = 999, terminal = mm >= numMismatch, use = mm;
            if (!models[mm][L]) { models[mm][L] = (s3o_model *)malloc(sizeof(s3o_model)); s3o_model_build(L, mm, models[mm][L]); }
            k = s3o_search_read(ix, models[mm][L], read, L, searchLimit, hits, HCAP, &total, we, st);
            if (k < 0) { overflow = 1; break; }
            if (total == 0) continue;                                   /* nohit: carried to the next phase */
            if (!bestMode && mm < 4) {                                  /* MAPQ special case, CPUfunctions.cpp:1756-1801 */
                for (e = 0; e <= mm; e++) if (we[e] > 0) { best = e; break; }
                if (best == mm) {
                    if (!terminal) continue;                            /* deferred */
                    use = mm + 1;
                    if (!models[use][L]) { models[use][L] = (s3o_model *)malloc(sizeof(s3o_model)); s3o_model_build(L, use, models[use][L]); }
                    k = s3o_search_read(ix, models[use][L], read, L, searchLimit, hits, HCAP, &total, we, st);
                    if (k < 0) { overflow = 1; break; }
                }
            }
            if (total > 0) {
                size_t room = occCap - used;
                int d = s3o_decode(ix, hits, k, pos + used, strand + used, nmis + used, room > 0x7fffffff ? 0x7fffffff : (int)room, st);
                if ((size_t)d > room) { overflow = 1; break; }
                if (seedLengths) {
                    int maxMis = (int)ceil(0.02 * fullLen);             /* MISMATCH_RATIO_FOR_LONG, CPUfunctions.cpp:1819 */
                    if (needMapq) maxMis *= 2;
                    d = validate_long(ix, read, L, fullLen, pos + used, strand + used, nmis + used, d, bestMode, bestMode ? mm : 0, maxMis, maxOutputPerRead);
                    if ((uint32_t)d > maxOutputPerRead) d = (int)maxOutputPerRead;
                    if (d == 0) { if (terminal) break; continue; }      /* nohit after validation: next phase */
                }
                used += (size_t)d;
                phase[r] = (uint8_t)(bestMode ? mm + 1 : mm);
                memcpy(&withError[(size_t)r * 8], we, 32);
            }
            break;
        }
    }
    occOffsets[n] = used;
    for (r = n; r < n; r++) occOffsets[r] = used;
    for (e = 0; e < 5; e++) for (r = 0; r <= S3O_MAX_READ; r++) free(models[e][r]);
    free(hits);
    return overflow ? -1 : (long)used;
}
