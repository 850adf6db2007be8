// s3dp_model.cpp -- host builder of the search-model tables (see s3dp_model.h).
//
// Behaviour follows SRAMismatchCasesPopulate (2bwt-flex/SRA2BWTMdl.c:1171-1700, COVER_ALL build
// mode, 16G model) and the post-compute of SRAModelPopulate (:4520-4602).  The case list is kept
// as a small text grammar instead of ~500 assignment statements:
//
//   case  := step { ' ' step }
//   step  := TYPE ':' FROM '>' TO ':' MIN MAX ':' CE [ ':' MOD ]
//   TYPE  := 'L' backward lookup | 'F' forward lookup (bidirectional) | 'B' backward-only BWT | 'X' bidirectional BWT
//   FROM/TO := cell letter a..e + '^' (first position of the cell) or '$' (last position); "0" = 0, "N" = L-1
//   MOD   := the four 2-mismatch check-and-extend window tweaks (SRA2BWTMdl.c:1303,1322,1339,1362)
#include "s3dp_model.h"
#include <stdio.h>
#include <string.h>

namespace {

const char *const kCases0[] = {"L:N>0:00:05"};
const char *const kCases1[] = {
    "L:b$>b^:00:00 B:a$>a^:01:20",
    "F:a^>a$:00:00 X:b^>b$:11:20"};
const char *const kCases2[] = {
    "L:c$>c^:00:00 B:b$>a^:02:20:p",
    "F:0>b$:00:00 X:c^>c$:12:20:q",
    "F:0>a$:00:00 X:b^>b$:11:10:r X:c^>c$:11:10:q",
    "F:b^>b$:00:00 X:c^>c$:11:10:s B:a$>a^:11:10:q"};
const char *const kCases3[] = {
    "L:d$>c^:00:00 B:b$>a^:03:20",
    "F:a^>b$:00:00 X:c^>d$:13:20",
    "F:a^>a$:00:00 X:b^>b$:11:20 X:c^>d$:22:20",
    "F:c^>c$:00:00 X:d^>d$:11:20 B:b$>a^:12:20",
    "L:d$>d^:00:00 B:c$>c^:11:20 B:b$>a^:12:20",
    "F:b^>b$:00:00 X:a$>a^:11:20 X:c^>d$:22:20"};
const char *const kCases4[] = {
    "L:e$>d^:00:00 B:c$>a^:04:20",
    "F:a^>c$:00:00 X:d^>e$:14:20",
    "F:a^>a$:00:00 X:b^>c$:11:20 X:d^>e$:13:20",
    "F:b^>c$:00:00 X:a$>a^:11:20 X:d^>e$:13:20",
    "F:a^>a$:00:00 X:b^>c$:22:20 X:d^>e$:12:20",
    "F:b^>c$:00:00 X:a$>a^:22:20 X:d^>e$:12:20",
    "F:b^>b$:00:00 X:a$>a^:11:20 X:c^>c$:11:20 X:d^>e$:12:20",
    "F:c^>c$:00:00 X:b$>b^:11:20 X:a$>a^:11:20 X:d^>e$:12:20",
    "F:d^>d$:00:00 X:e^>e$:11:20 B:c$>a^:33:20",
    "L:e$>e^:00:00 B:d$>d^:11:20 B:c$>a^:33:20"};
const char *const *const kCases[5] = {kCases0, kCases1, kCases2, kCases3, kCases4};
const int kIds[5][10] = {{1}, {11, 12}, {21, 22, 23, 24}, {31, 32, 33, 35, 36, 34}, {41, 42, 43, 44, 45, 46, 47, 48, 49, 410}};

struct Cells { unsigned lo[5], hi[5]; unsigned L; };

// The reference computes cell ends as `unsigned = (unsigned * float) - 1` and following cells as
// `unsigned = unsigned + (unsigned * float) - 1`  (SRA2BWTMdl.c:1177-1203): float arithmetic, truncation on store.
Cells cellsFor(unsigned L, int maxError)
{
    static const float kFrac[5] = {1.0f, 0.5f, 0.3f, 0.25f, 0.2f};
    Cells c;
    memset(&c, 0, sizeof c);
    c.L = L;
    const int nc = maxError + 1;
    const float f = kFrac[maxError];
    for (int k = 0; k < nc; k++) {
        c.lo[k] = k ? c.hi[k - 1] + 1 : 0;
        if (k == nc - 1) c.hi[k] = L - 1;
        else if (k == 0) { float t = L * f; t = t - 1; c.hi[k] = (unsigned)t; }
        else { float t = c.lo[k] + (L * f); t = t - 1; c.hi[k] = (unsigned)t; }
    }
    return c;
}

int refPos(const Cells &c, const char *&p)
{
    if (*p == '0') { p++; return 0; }
    if (*p == 'N') { p++; return (int)c.L - 1; }
    int cell = *p - 'a';
    char side = p[1];
    p += 2;
    return (int)(side == '^' ? c.lo[cell] : c.hi[cell]);
}

}  // namespace

void s3_model_build(int readLength, int maxError, S3Model *m)
{
    memset(m, 0, sizeof *m);
    m->readLength = readLength;
    m->nCases = s3_num_cases(maxError);
    const Cells cells = cellsFor((unsigned)readLength, maxError);
    for (int ci = 0; ci < m->nCases; ci++) {
        S3Case &cs = m->cs[ci];
        cs.id = kIds[maxError][ci];
        const char *p = kCases[maxError][ci];
        int leftMost = -1;
        while (*p) {
            S3Step &st = cs.st[cs.nSteps++];
            char t = *p; p += 2;
            st.type = t == 'L' ? S3_T_BACK_LOOKUP : t == 'F' ? S3_T_FWD_LOOKUP : t == 'B' ? S3_T_BACK_ONLY : S3_T_BIDIR;
            st.start = (int16_t)refPos(cells, p); p++;          // '>'
            st.end = (int16_t)refPos(cells, p); p++;            // ':'
            st.minErr = (int8_t)(p[0] - '0'); st.maxErr = (int8_t)(p[1] - '0'); p += 3;
            int ce = (p[0] - '0') * 10 + (p[1] - '0'); p += 2;
            int ceStart = 0, ceEnd = readLength - 1;
            if (*p == ':') {
                const unsigned aHi = cells.hi[0], bLo = cells.lo[1], bHi = cells.hi[1], cLo = cells.lo[2];
                switch (p[1]) {
                case 'p': ceStart = (int)(bHi - 9); break;
                case 'q': ceEnd = (int)(cLo + 9); break;
                case 'r': ceStart = aHi >= 17 ? 0 : 17; break;
                case 's': ceEnd = (bHi - bLo + 1 >= 17) ? readLength - 1 : (int)(bHi - 18); break;
                }
                p += 2;
            }
            if (*p == ' ') p++;
            st.dir = st.start > st.end ? -1 : 1;
            st.len = (int16_t)((st.end - st.start) * st.dir + 1);
            if (st.dir > 0) {
                if (leftMost < 0) leftMost = st.start;
                st.lm0 = (int16_t)leftMost;
            } else {
                if (leftMost < 0) leftMost = st.start + 1;
                st.lm0 = (int16_t)leftMost;
                leftMost -= st.len;
            }
            if (ce > 32) ce = 32;                                // MAX_CE_THRESHOLD, SRA2BWTMdl.h:52
            st.ceThr = (int8_t)ce;
            if (ceStart > ceEnd) { int x = ceStart; ceStart = ceEnd; ceEnd = x; }
            st.ceStart = (int16_t)ceStart; st.ceEnd = (int16_t)ceEnd;
        }
    }
}

// text form used by tests: identical to the oracle harness' dump of the reference's SRAModel
int s3_model_text(int readLength, int maxError, char *buf, unsigned long bufLen)
{
    S3Model m;
    s3_model_build(readLength, maxError, &m);
    unsigned long o = 0;
    for (int ci = 0; ci < m.nCases; ci++) {
        const S3Case &cs = m.cs[ci];
        int n = snprintf(buf + o, o < bufLen ? bufLen - o : 0, "L %d E %d C %d id %d :", readLength, maxError, ci, cs.id);
        o += n;
        unsigned long long h = 1469598103934665603ULL;
        for (int s = 0; s < cs.nSteps; s++) {
            const S3Step &st = cs.st[s];
            int lookup = st.type == S3_T_BACK_LOOKUP || st.type == S3_T_FWD_LOOKUP;
            n = snprintf(buf + o, o < bufLen ? bufLen - o : 0, " [%d %d %d ce %d %d %d err %d %d %d]", st.type, st.start, st.end, st.ceThr,
                         st.ceStart, st.ceEnd, lookup ? 0 : 1, st.minErr, st.maxErr);
            o += n;
            for (int q = 0; q < st.len; q++) {
                int lm = st.dir > 0 ? st.lm0 : st.lm0 - q;
                h = (h ^ (unsigned long long)(unsigned int)lm) * 1099511628211ULL;
            }
        }
        n = snprintf(buf + o, o < bufLen ? bufLen - o : 0, " lma %016llx\n", h);
        o += n;
    }
    return (int)o;
}
