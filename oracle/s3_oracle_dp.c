/* s3_oracle_dp.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's semi-global affine-gap DP (DV-DPfunctions.cu:61-512):
 *   score pass      DPScoreNHitPos      :61-144
 *   table fill      GenerateDPTable     :146-241
 *   driver          SemiGlobalAligntment:243-314   (scheme 1: full table; scheme 2: score pass + windowed refill)
 *   traceback       GPUBacktrack        :316-512
 *
 * PINNED on the GPU box: the reference's DP exists only as device code, and that device code itself is the check -- oracle/ref.mk builds
 * DV-DPfunctions.cu:35-512 with hipcc for gfx950 from where it lies (oracle/_ref/libdpref.so, driver oracle/ref_dp_driver.hip), and
 * tests/test_gpu_dp_reference.py requires this restatement and the product's kernels to equal it bit for bit (scores, start columns, start
 * offsets, actual right clips, tie counts, traceback op strings; both schemes, clips, anchors).  On the CPU the restatement is checked by
 * self-consistency (tests/test_oracle_dp.py: the score recomputed from the emitted pattern equals the reported score) and by the known
 * answers quoted in SURVEY.md 8c-5.  Domain: readLength < maxReadLength, as in the reference's engines (its table rows have maxReadLength
 * cells for read positions 0..readLength).
 *
 * Input layout = the reference's: sequences are 1-based, 2 bits per character, 16 per word, first character of a
 * word in the most significant bits, words of problem t interleaved by 32:
 *   word w of problem t at (t/32) * (ceil16(maxLen) * 32) + (t%32) + 32*w          (MC_DnaUnpack / MC_ReadUnpack :57-59)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-32000)

typedef struct { int match, mismatch, gapOpen, gapExtend; } s3o_dp_scores;

static inline int clampS(int x) { return x > NEG_INF ? x : NEG_INF; }
static inline int imax(int a, int b) { return a > b ? a : b; }

static inline unsigned seqChar(const uint32_t *packed, unsigned t, unsigned maxLen, unsigned i)
{
    const unsigned base = (t >> 5) * (((maxLen + 15) / 16) << 5) + (t & 31);
    return (packed[base + ((i >> 4) << 5)] >> ((15 - (i & 15)) << 1)) & 3u;
}

/* one problem; tables H and E are (W+1) x (L+1) int16 when tableH != NULL, otherwise a score-only pass */
static void fill(const uint32_t *dna, unsigned t, unsigned maxDNA, const uint32_t *rd, unsigned maxRead, unsigned W, unsigned L,
                 const s3o_dp_scores *sc, int clipLt, int clipRtLoc, unsigned anchorLeft, unsigned anchorRight, unsigned refOffset,
                 int16_t *H, int16_t *E, int *maxScore, unsigned *hitPos, unsigned *scRight, unsigned *maxCount)
{
    const int go = sc->gapOpen, ge = sc->gapExtend, gi = go - ge;
    const unsigned S = L + 1;
    int16_t *h0 = (int16_t *)malloc(sizeof(int16_t) * S), *e0 = (int16_t *)malloc(sizeof(int16_t) * S);   /* rolling column */
    unsigned i, j;
    int up = gi;
    *maxScore = NEG_INF; *maxCount = 0; *hitPos = 0; *scRight = 0;
    h0[0] = (int16_t)clampS(0); e0[0] = (int16_t)clampS(gi);
    for (i = 1; i <= L; i++) {
        if ((int)i <= clipLt) { h0[i] = (int16_t)clampS(go); e0[i] = (int16_t)clampS(go + gi); }
        else { up += ge; h0[i] = (int16_t)clampS(up); e0[i] = (int16_t)clampS(up + gi); }
    }
    if (H) { memcpy(H, h0, sizeof(int16_t) * S); memcpy(E, e0, sizeof(int16_t) * S); }
    {
        int prevInit = 0;
        for (j = 1; j <= W; j++) {
            const unsigned refChar = seqChar(dna, t, maxDNA, j + refOffset);
            const int init = (j + refOffset >= anchorLeft) ? NEG_INF : 0;
            int upScore = init, openUp = init + gi, prevUp = prevInit;
            if (H) { H[(size_t)j * S] = (int16_t)clampS(upScore); E[(size_t)j * S] = (int16_t)clampS(openUp); }
            for (i = 1; i <= L; i++) {
                const int prevR = h0[i];
                int gapped = imax(go + prevR, ge + e0[i]);
                const int eStore = clampS(gapped);
                openUp = imax(ge + openUp, go + upScore);
                gapped = imax(openUp, gapped);
                upScore = imax(gapped, prevUp + (refChar == seqChar(rd, t, maxRead, i) ? sc->match : sc->mismatch));
                e0[i] = (int16_t)eStore;
                h0[i] = (int16_t)clampS(upScore);
                if (H) { H[(size_t)j * S + i] = h0[i]; E[(size_t)j * S + i] = e0[i]; }
                prevUp = prevR;
                if ((int)i <= clipLt) { openUp = imax(init + gi, openUp); prevUp = imax(prevInit, prevUp); }
                if ((int)i >= clipRtLoc && j + refOffset >= anchorRight) {
                    if (upScore > *maxScore) { *maxScore = upScore; *hitPos = j; *scRight = L - i; *maxCount = 1; }
                    else if (upScore == *maxScore) ++*maxCount;
                }
            }
            prevInit = init;
        }
    }
    free(h0); free(e0);
}

/* returns 0.  Outputs per problem: score, hitLoc (start column when score >= cutoff, else end column), startOffset,
 * clipRt (in/out), maxScoreCount, pattern (patternLen bytes per problem, zero-terminated op string as the reference emits it) */
int s3o_dp_align(const uint32_t *packedDNA, const uint32_t *dnaLengths, unsigned maxDNALength, unsigned maxDPTableLength,
                 const uint32_t *packedReads, const uint32_t *readLengths, unsigned maxReadLength, unsigned n,
                 const uint32_t *clipLt, uint32_t *clipRt, const uint32_t *anchorLeft, const uint32_t *anchorRight,
                 const s3o_dp_scores *sc, const int32_t *cutoffs, int scheme,
                 int32_t *scores, uint32_t *hitLocs, uint32_t *startOffsets, uint32_t *maxScoreCounts, uint8_t *pattern, unsigned patternLen)
{
    unsigned t;
    for (t = 0; t < n; t++) {
        const unsigned cl = clipLt ? clipLt[t] : 0, cr = clipRt ? clipRt[t] : 0;
        const unsigned aL = anchorLeft ? anchorLeft[t] : maxDNALength, aR = anchorRight ? anchorRight[t] : 0;
        const unsigned L = readLengths[t];
        unsigned W = dnaLengths[t], refOffset = 0, hitPos = 0, scRight = 0, cnt = 0;
        int maxScore;
        const s3o_dp_scores s = *sc;
        const int go = s.gapOpen, ge = s.gapExtend;
        if (scheme == 2) {
            fill(packedDNA, t, maxDNALength, packedReads, maxReadLength, W, L, &s, (int)cl, (int)(L - cr), aL, aR, 0, NULL, NULL, &maxScore, &hitPos, &scRight, &cnt);
            if (W > maxDPTableLength - 1) W = maxDPTableLength - 1;
            if (hitPos > W) {
                refOffset = hitPos - W;
                if (refOffset >= aL) refOffset = aL > 0 ? aL - 1 : 0;
            }
        }
        {
            const unsigned S = L + 1;
            int16_t *H = (int16_t *)malloc(sizeof(int16_t) * (size_t)(W + 1) * S), *E = (int16_t *)malloc(sizeof(int16_t) * (size_t)(W + 1) * S);
            uint8_t *pat = pattern + (size_t)t * patternLen;
            unsigned p = 0;
            fill(packedDNA, t, maxDNALength, packedReads, maxReadLength, W, L, &s, (int)cl, (int)(L - cr), aL, aR, refOffset, H, E, &maxScore, &hitPos, &scRight, &cnt);
            startOffsets[t] = refOffset; hitLocs[t] = hitPos; scores[t] = maxScore; maxScoreCounts[t] = cnt;
            if (clipRt) clipRt[t] = scRight;
            pat[0] = 0;
            if (maxScore >= cutoffs[t]) {
#define HH(j, i) H[(size_t)(j) * S + (i)]
#define EE(j, i) E[(size_t)(j) * S + (i)]
#define APP(x) do { if (p < patternLen) pat[p] = (uint8_t)(x); ++p; } while (0)
                const unsigned clipRtLen = clipRt ? clipRt[t] : 0;
                unsigned readPos = L - clipRtLen, refIndex = hitPos;
                unsigned readChar, refChar;
                int16_t cur, next = 0, init, prevInit;
                enum { NORMAL, I_EXT, D_EXT, SM_EXIT, SI_EXIT } state = NORMAL;
                if (clipRtLen > 0) { APP('S'); APP('V'); APP(clipRtLen); }
                readChar = seqChar(packedReads, t, maxReadLength, readPos);
                refChar = seqChar(packedDNA, t, maxDNALength, refOffset + refIndex);
                cur = HH(refIndex, readPos);
                init = (refOffset + refIndex >= aL) ? NEG_INF : 0;
                prevInit = (refOffset + refIndex > aL) ? NEG_INF : 0;
#define NEXT_REF() do { --refIndex; refChar = seqChar(packedDNA, t, maxDNALength, refOffset + refIndex); init = prevInit; \
                        prevInit = (refOffset + refIndex > aL) ? NEG_INF : 0; } while (0)
#define NEXT_READ() do { --readPos; readChar = seqChar(packedReads, t, maxReadLength, readPos); } while (0)
#define DIST() (refChar == readChar ? s.match : s.mismatch)
                while (readPos > 0 && refIndex > 0) {
                    if (state == NORMAL) {
                        if (cur == DIST() + (next = HH(refIndex - 1, readPos - 1))) {
                            APP('m' - ((refChar == readChar) << 5)); NEXT_REF(); NEXT_READ(); cur = next;
                        } else if (cur == go + (next = HH(refIndex - 1, readPos))) {
                            APP('D'); NEXT_REF(); cur = next;
                        } else if (cur == ge + EE(refIndex - 1, readPos)) {
                            APP('D'); NEXT_REF(); cur = (int16_t)(cur - ge); state = D_EXT;
                        } else {
                            if (readPos <= cl + 1) {
                                if (cur == prevInit + DIST()) { state = SM_EXIT; break; }
                                else if (cur == init + go) { state = SI_EXIT; break; }
                            }
                            if (cur == go + (next = HH(refIndex, readPos - 1))) { APP('I'); NEXT_READ(); cur = next; }
                            else { APP('I'); NEXT_READ(); cur = (int16_t)(cur - ge); state = I_EXT; }
                        }
                    } else {
                        if (state == D_EXT) { APP('D'); NEXT_REF(); }
                        else {
                            if (readPos <= cl + 1 && cur == init + go) { state = SI_EXIT; break; }
                            APP('I'); NEXT_READ();
                        }
                        if (cur == go + (next = HH(refIndex, readPos))) { state = NORMAL; cur = next; }
                        else cur = (int16_t)(cur - ge);
                    }
                }
                if (refIndex == 0) {
                    const unsigned scNum = cl < readPos ? cl : readPos;
                    if (scNum < readPos) { APP('I'); APP('V'); APP(readPos - scNum); }
                    APP('S'); APP('V'); APP(scNum);
                } else if (state == SI_EXIT) { APP('I'); APP('S'); APP('V'); APP(readPos - 1); }
                else if (state == SM_EXIT) { APP('m' - ((refChar == readChar) << 5)); APP('S'); APP('V'); APP(readPos - 1); refIndex -= 1; }
                APP(0);
                hitLocs[t] = refOffset + refIndex;
            }
            free(H); free(E);
        }
    }
    return 0;
}
