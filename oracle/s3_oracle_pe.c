/* s3_oracle_pe.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's paired-end path with the DP stages off (`soap3-dp pair ... -s k`):
 *   phase ladders        soap3_dp_pair_align            alignment.cu:1965-2013
 *                        all_best_alignment             alignment.cu:1236-1376   (all best + SAM: phases 1, 2, k)
 *                        four_phases_alignment          alignment.cu:1119-1234   (unique / random best: phases 0, 1, 2, k)
 *   per pair decisions   hostKernel, PAIR_END_READ      CPUfunctions.cpp:1980-2560
 *   pairing              PEMappingOccurrences / Core    PEAlgnmt.cpp:201-291,480-560 (stable radix sort by position, two-pointer sweep)
 *                        PEIsPairEndMatch / OutOfRange  PEAlgnmt.cpp:566-606
 *   pair statistics      PEStatsPEPairList              PEAlgnmt.cpp:777-832
 * PINNED through the SAM the reference itself writes for the fixtures (tests/golden/g4, made by oracle/ref_harness.cpp with the
 * reference's own hostKernel + PEAlgnmt + pairOutputSAMAPI): the product's command line must reproduce those records, and the
 * product's device results must equal this restatement.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "s3_oracle.h"

typedef struct {
    uint8_t  phase;               /* 0 = no proper pair; else (mismatch phase of the pairing) + 1 */
    uint8_t  min1, min2;          /* fewest mismatches of each end in that phase (255 = end without hit / not computed) */
    uint8_t  minTotal, secMinTotal;   /* optimal / "suboptimal" pair mismatch sums (127 = none), PEStatsPEPairList */
    uint8_t  bestS1, bestS2, bestM1, bestM2;
    uint8_t  pad[3];
    uint32_t x0_1, x1_1, x0_2, x1_2;  /* WithError[min], WithError[min+1] of each end (MAPQ modes) */
    uint32_t numPairs, numMin;    /* valid pairs; pairs with the optimal mismatch sum */
    uint32_t bestA1, bestA2;      /* the optimal pair: text positions of end 1 / end 2 */
} s3o_pair;

typedef struct { uint32_t a1, a2; uint8_t s1, s2, m1, m2; } s3o_pair_xa;

typedef struct { uint32_t pos; uint8_t strand, mis; } occ_t;

static void unpack(const uint32_t *queries, uint32_t r, uint32_t wpq, int L, uint8_t *out)
{
    const uint32_t *p = queries + (size_t)(r / 32) * 32 * wpq + r % 32;
    int i;
    for (i = 0; i < L; i++) out[i] = (uint8_t)((p[(size_t)(i / 16) * 32] >> (2 * (i % 16))) & 3);
}

/* stable sort by position (PERadixSort is an LSD radix sort: equal positions keep their order) */
static void stable_sort(occ_t *a, int n, occ_t *tmp)
{
    int w, i;
    for (w = 1; w < n; w *= 2) {
        for (i = 0; i < n; i += 2 * w) {
            int l = i, m = i + w < n ? i + w : n, r = i + 2 * w < n ? i + 2 * w : n, x = l, y = m, o = l;
            while (x < m && y < r) tmp[o++] = a[y].pos < a[x].pos ? a[y++] : a[x++];
            while (x < m) tmp[o++] = a[x++];
            while (y < r) tmp[o++] = a[y++];
        }
        memcpy(a, tmp, sizeof(occ_t) * (size_t)n);
    }
}

typedef struct { s3o_pair_xa *v; size_t n, cap; } pairvec;
static void pv_push(pairvec *pv, const occ_t *o1, const occ_t *o2)
{
    if (pv->n == pv->cap) { pv->cap = pv->cap ? pv->cap * 2 : 1024; pv->v = (s3o_pair_xa *)realloc(pv->v, pv->cap * sizeof(s3o_pair_xa)); }
    pv->v[pv->n].a1 = o1->pos; pv->v[pv->n].s1 = o1->strand; pv->v[pv->n].m1 = o1->mis;
    pv->v[pv->n].a2 = o2->pos; pv->v[pv->n].s2 = o2->strand; pv->v[pv->n].m2 = o2->mis;
    pv->n++;
}

/* left leg `l` (must be '+'), right leg `r` (must be '-'); PEAlgnmt.cpp:566-597 */
static int is_match(const occ_t *l, const occ_t *r, uint32_t patLen, uint32_t lb, uint32_t ub)
{
    const uint32_t gap = r->pos + patLen - 1u - l->pos + 1u;
    return lb <= gap && gap <= ub && l->strand == 1 && r->strand == 2;
}
static int out_of_range(const occ_t *l, const occ_t *r, uint32_t patLen, uint32_t ub)
{
    if (l->strand == r->strand) return 0;
    return (uint32_t)(l->pos + ub) < (uint32_t)(r->pos + patLen - 1u);
}

static void pe_core(const occ_t *l1, int n1, const occ_t *l2, int n2, uint32_t patLen, uint32_t lb, uint32_t ub, pairvec *pv)
{
    int i1 = 0, i2 = 0, i;
    while (i1 < n1 && i2 < n2) {
        if (l1[i1].pos <= l2[i2].pos) {
            if (l1[i1].strand == 1)
                for (i = i2; i < n2; i++) {
                    if (is_match(&l1[i1], &l2[i], patLen, lb, ub)) pv_push(pv, &l1[i1], &l2[i]);
                    if (out_of_range(&l1[i1], &l2[i], patLen, ub)) break;
                }
            i1++;
        } else {
            if (l2[i2].strand == 1)
                for (i = i1; i < n1; i++) {
                    if (is_match(&l2[i2], &l1[i], patLen, lb, ub)) pv_push(pv, &l1[i], &l2[i2]);
                    if (out_of_range(&l2[i2], &l1[i], patLen, ub)) break;
                }
            i2++;
        }
    }
}

/* Reads 2p and 2p+1 are the two ends of pair p.  reportType as -h (1..4).  maxHitsPerEnd = MaxHitsEachEndForPairing,
 * peMaxOutputPerPair caps the XA list (best pair included).  Outputs:
 *   pairs[nPairs]; xaOffsets[nPairs+1] + xa[] (pairs with the optimal mismatch sum, the best one left out, in pairing order);
 *   occOffsets[2*nPairs+1] + occ arrays: the occurrence lists of pairs that end without a proper pair (in the order the reference's
 *   unproperlypairOutputSAMAPI sees them: position-sorted when both ends had hits).  Returns 0, or -1 when a capacity is too small. */
long s3o_align_pair(const s3o_index *ix, const uint32_t *queries, const uint32_t *readLengths, uint32_t nReads, uint32_t wordPerQuery,
                    int numMismatch, uint32_t maxHitsPerEnd, int reportType, uint32_t insertLow, uint32_t insertHigh, uint32_t peMaxOutputPerPair,
                    s3o_pair *pairs, uint64_t *xaOffsets, s3o_pair_xa *xa, size_t xaCap,
                    uint64_t *occOffsets, uint32_t *occPos, uint8_t *occStrand, uint8_t *occMis, size_t occCap, s3o_stats *st)
{
    const int bestMode = reportType >= 3, mapq = !bestMode;
    static const int HCAP = 1 << 16;
    s3o_model *models[5][S3O_MAX_READ + 1];
    s3o_hit *hits = (s3o_hit *)malloc(sizeof(s3o_hit) * (size_t)HCAP);
    int phases[4], nPhases = 0, overflow = 0, e;
    uint32_t p;
    size_t xaUsed = 0, occUsed = 0, lcap = 1 << 16;
    occ_t *l1 = (occ_t *)malloc(sizeof(occ_t) * lcap), *l2 = (occ_t *)malloc(sizeof(occ_t) * lcap), *tmp = (occ_t *)malloc(sizeof(occ_t) * lcap);
    uint32_t *dpos = (uint32_t *)malloc(4 * lcap);
    uint8_t *dstr = (uint8_t *)malloc(lcap), *dmis = (uint8_t *)malloc(lcap);
    pairvec pv = {0, 0, 0};
    memset(models, 0, sizeof models);
    if (reportType == 2) { phases[nPhases++] = 1; if (numMismatch > 1) phases[nPhases++] = 2; if (numMismatch > 2) phases[nPhases++] = numMismatch; }
    else if (reportType == 1) phases[nPhases++] = numMismatch;
    else { phases[nPhases++] = 0; if (numMismatch > 0) phases[nPhases++] = 1; if (numMismatch > 1) phases[nPhases++] = 2; if (numMismatch > 2) phases[nPhases++] = numMismatch; }

    for (p = 0; p < nReads / 2 && !overflow; p++) {
        const uint32_t r1 = 2 * p, r2 = 2 * p + 1;
        const int L1 = (int)readLengths[r1], L2 = (int)readLengths[r2];
        uint8_t q1[S3O_MAX_READ], q2[S3O_MAX_READ];
        s3o_pair *P = &pairs[p];
        int ph;
        unpack(queries, r1, wordPerQuery, L1, q1);
        unpack(queries, r2, wordPerQuery, L2, q2);
        memset(P, 0, sizeof *P);
        P->min1 = P->min2 = 255; P->minTotal = P->secMinTotal = 127;
        xaOffsets[p] = xaUsed;
        occOffsets[r1] = occOffsets[r2] = occUsed;
        for (ph = 0; ph < nPhases; ph++) {
            const int mm = phases[ph], T = ph + 1 == nPhases;
            uint32_t t1, t2, we1[8], we2[8];
            int k1, k2, n1 = 0, n2 = 0, i, min1 = 999, min2 = 999;
#define MODEL(m_, L_) (models[m_][L_] ? models[m_][L_] : (s3o_model_build(L_, m_, models[m_][L_] = (s3o_model *)malloc(sizeof(s3o_model))), models[m_][L_]))
            k1 = s3o_search_read(ix, MODEL(mm, L1), q1, L1, maxHitsPerEnd, hits, HCAP, &t1, we1, st);
            if (k1 < 0) { overflow = 1; break; }
            if (t1 == 0 && !T) continue;                                             /* CPUfunctions.cpp:1985-1996 */
            if (mapq && !T) {                                                        /* :2003-2027 */
                for (e = 0; e <= mm; e++) if (we1[e] > 0) { min1 = e; break; }
                if (min1 == mm && t1 < maxHitsPerEnd) continue;
            }
            if (mapq && t1 > 0) {                                                    /* :2059-2098 */
                min1 = 999;
                for (e = 0; e <= mm; e++) if (we1[e] > 0) { min1 = e; break; }
                if (min1 == mm && t1 < maxHitsPerEnd && mm < 4) {
                    k1 = s3o_search_read(ix, MODEL(mm + 1, L1), q1, L1, maxHitsPerEnd, hits, HCAP, &t1, we1, st);
                    if (k1 < 0) { overflow = 1; break; }
                }
                P->x0_1 = we1[min1]; P->x1_1 = we1[min1 + 1];
            }
            n1 = s3o_decode(ix, hits, k1, dpos, dstr, dmis, (int)lcap, st);
            if ((size_t)n1 > lcap) { overflow = 1; break; }
            for (i = 0; i < n1; i++) { l1[i].pos = dpos[i]; l1[i].strand = dstr[i]; l1[i].mis = dmis[i]; }
            k2 = s3o_search_read(ix, MODEL(mm, L2), q2, L2, maxHitsPerEnd, hits, HCAP, &t2, we2, st);
            if (k2 < 0) { overflow = 1; break; }
            if (mapq && t2 > 0) {                                                    /* :2100-2138 */
                for (e = 0; e <= mm; e++) if (we2[e] > 0) { min2 = e; break; }
                if (min2 == mm && t2 < maxHitsPerEnd && mm < 4) {
                    k2 = s3o_search_read(ix, MODEL(mm + 1, L2), q2, L2, maxHitsPerEnd, hits, HCAP, &t2, we2, st);
                    if (k2 < 0) { overflow = 1; break; }
                }
                P->x0_2 = we2[min2]; P->x1_2 = we2[min2 + 1];
            }
            n2 = s3o_decode(ix, hits, k2, dpos, dstr, dmis, (int)lcap, st);
            if ((size_t)n2 > lcap) { overflow = 1; break; }
            for (i = 0; i < n2; i++) { l2[i].pos = dpos[i]; l2[i].strand = dstr[i]; l2[i].mis = dmis[i]; }
            P->min1 = min1 > 254 ? 255 : (uint8_t)min1; P->min2 = min2 > 254 ? 255 : (uint8_t)min2;
            pv.n = 0;
            if (t1 > 0 && t2 > 0) {                                                  /* :2281-2289 */
                if (n1 > 1) stable_sort(l1, n1, tmp);
                if (n2 > 1) stable_sort(l2, n2, tmp);
                pe_core(l1, n1, l2, n2, (uint32_t)L2, insertLow, insertHigh, &pv);   /* pe_in->patternLength = readLength of the 2nd end */
            }
            if (pv.n > 0) {
                /* PEStatsPEPairList: optimal = fewest mismatches, ties by |m1 - m2|, first wins; suboptimal = the last demoted optimal */
                size_t best = 0, sub = (size_t)-1, j;
                unsigned stats[16];
                int oc = 255, od = 255;
                memset(stats, 0, sizeof stats);
                for (j = 0; j < pv.n; j++) {
                    const int tot = pv.v[j].m1 + pv.v[j].m2;
                    int d = (int)pv.v[j].m1 - (int)pv.v[j].m2;
                    if (d < 0) d = -d;
                    stats[tot]++;
                    if (tot < oc) { sub = oc == 255 ? (size_t)-1 : best; best = j; oc = tot; od = d; }
                    else if (tot == oc && d < od) { best = j; od = d; }
                }
                P->phase = (uint8_t)(mm + 1);
                P->numPairs = (uint32_t)pv.n;
                P->minTotal = (uint8_t)oc;
                P->numMin = stats[oc];
                if (sub != (size_t)-1) P->secMinTotal = (uint8_t)(pv.v[sub].m1 + pv.v[sub].m2);
                P->bestA1 = pv.v[best].a1; P->bestS1 = pv.v[best].s1; P->bestM1 = pv.v[best].m1;
                P->bestA2 = pv.v[best].a2; P->bestS2 = pv.v[best].s2; P->bestM2 = pv.v[best].m2;
                {
                    uint32_t total = 1;
                    for (j = 0; j < pv.n && total < peMaxOutputPerPair; j++) {
                        if (j == best || pv.v[j].m1 + pv.v[j].m2 > oc) continue;
                        if (xaUsed >= xaCap) { overflow = 1; break; }
                        xa[xaUsed++] = pv.v[j];
                        total++;
                    }
                }
                break;
            }
            if (!T) continue;                                                        /* :2443-2449: both ends go on to the next phase */
            if (occUsed + (size_t)n1 + (size_t)n2 > occCap) { overflow = 1; break; }
            for (i = 0; i < n1; i++) { occPos[occUsed] = l1[i].pos; occStrand[occUsed] = l1[i].strand; occMis[occUsed] = l1[i].mis; occUsed++; }
            occOffsets[r2] = occUsed;
            for (i = 0; i < n2; i++) { occPos[occUsed] = l2[i].pos; occStrand[occUsed] = l2[i].strand; occMis[occUsed] = l2[i].mis; occUsed++; }
            break;
        }
    }
    xaOffsets[nReads / 2] = xaUsed;
    occOffsets[nReads / 2 * 2] = occUsed;
    for (e = 0; e < 5; e++) { uint32_t r; for (r = 0; r <= S3O_MAX_READ; r++) free(models[e][r]); }
    free(hits); free(l1); free(l2); free(tmp); free(dpos); free(dstr); free(dmis); free(pv.v);
    return overflow ? -1 : 0;
}
