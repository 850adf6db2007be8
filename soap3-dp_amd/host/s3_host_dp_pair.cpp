// s3_host_dp_pair.cpp -- SAM records of the DP stages for pairs (host side of the drop-in program).  Mirrors:
//   outputRead2              OutputDPResult.cpp:217-447   which default-DP result of a pair is reported
//   pairDPOutputSAMAPI       BGS-IO.cpp:4504-5554         the two records of a default-DP pair (one end from the BWT stage, one from the DP)
//   outputDeepDPResult2      OutputDPResult.cpp:564-845   which deep-DP result is reported
//   pairDeepDPOutputSAMAPI   BGS-IO.cpp:3824-4502         the two records of a deep-DP pair (both ends from the DP)
//   getMapQualScore / ForDP / 2 / ForDP2 / ForPair, bwaLikePairQualScore   BGS-IO.cpp:2277-2580
// The two reference writers are near copies of each other, written out per end; here one end is described once (EndView) and the
// differences (which hits count, how pairs are compared) stay with the writer they belong to.  Quirks kept on purpose: the BWT-stage
// statistics of the second end only count when its X0 is above ONE (:5104, :4170) while the first end's count above zero; a pair
// whose ends read through each other loses the end with more mismatches (:4667-4690).
// Pinned: tests/golden/dp2 holds the reference's own records for given result vectors (tests/test_host_dp_sam_cpu.py).
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "s3_host.h"

namespace s3host {

static const uint32_t NONE = 0xFFFFFFFFu;

struct SamWriter::EndView {
    uint32_t r = 0, len = 0;
    bool aligned = false, fromDP = false, negative = false;
    uint32_t pos = NONE;
    int score = 0;                                        // mismatches (BWT end) or DP score
    uint64_t tp = 0;
    uint32_t chr = 0;
    int trim = 0, span = 0;                               // boundary trim; reference span the insert size is computed with
    std::string cigar, md;                                // cigar empty: "<len>M"
    int mis = 0, gapOpen = 0, gapExt = 0, avgQual = 20;
    int x0 = 0, x1 = 0, mapq = 0;
    int bestScore = 0, secBestScore = 0, isBest = 1;
};

static int spanOfCigar(const std::string &c)              // readLengthWithCigar, BGS-IO.cpp:3795
{
    int len = 0, x = 0;
    char op = ' ';
    for (size_t i = 0; i < c.size();) {
        x = 0;
        while (i < c.size() && c[i] <= '9') { x = x * 10 + (c[i] - '0'); i++; }
        if (i >= c.size()) break;
        op = c[i++];
        if (op == 'M' || op == 'm' || op == 'D') len += x;
    }
    if (op == 'D') len -= x;
    return len;
}

static int mapqFromDP(int n, int dp, int maxDP, int avg, int maxQ, int minQ)                 // getMapQualScoreForDP, BGS-IO.cpp:2501
{
    if (n != 1) return minQ;
    int idx = 0;
    if (dp < maxDP) idx = (int)((1.0 - (double)dp / maxDP) * 100.0 - 1.0) / 5 + 1;
    return tableScore(idx, avg, maxQ, minQ);
}

static int mapqFromDP2(int dp, int maxDP, int avg, int x0, int x1, int best, int secBest, int isBest, int validPairs, int maxQ, int minQ)   // :2535
{
    if (!(x0 == 1 && validPairs == 1)) return minQ;
    if (!isBest && x1 > 1) return minQ;
    int idx = 0;
    if (dp < maxDP) idx = (int)((1.0 - (double)dp / maxDP) * 100.0 - 1.0) / 4 + 1;
    int q = tableScore(idx, avg, maxQ, minQ);
    if (best > secBest && ((double)best - secBest) / maxDP < 0.2) q = minQ;
    return q < minQ ? minQ : q;
}

static int mapqFromMis2(int mis, int avg, int x0, int x1, int isBest, int validPairs, int maxQ, int minQ)                                  // getMapQualScore2, :2465
{
    if (!(x0 == 1 && validPairs == 1)) return minQ;
    if (!isBest && x1 > 1) return minQ;
    return tableScore(mis, avg, maxQ, minQ);
}

static void bwaLikePair(int x0a, int x1a, int x0b, int x1b, const int *gLogN, int op, int opNum, int sub, int subNum, int lenA, int lenB, int &qa, int &qb)   // :2415
{
    int q0 = bwaLikeSingle(x0a, x1a, gLogN), q1 = bwaLikeSingle(x0b, x1b, gLogN), qp = 0;
    op *= 10; sub *= 10;
    if (q0 > 0 && q1 > 0) { qp = q0 + q1; if (qp > 60) qp = 60; q0 = q1 = qp; }
    else {
        if (opNum == 1) {
            if (subNum == 0) qp = 29;
            else if (op - sub > 0.3 * ((lenA + lenB) / 2)) qp = 23;
            else { subNum = subNum > 255 ? 255 : subNum; qp = (op - sub) / 2 - gLogN[subNum]; if (qp < 0) qp = 0; }
        }
        if (q0 == 0) q0 = qp + 7 < q1 ? qp + 7 : q1;
        if (q1 == 0) q1 = qp + 7 < q0 ? qp + 7 : q0;
    }
    qa = q0; qb = q1;
}

// best / second-best tally of one end over the other results of the pair (BGS-IO.cpp:4852-4942 and its three copies).  bigger: DP scores
// (a larger one is better, equal positions are one hit, a hit counts num_sameScore times); otherwise mismatch counts.
struct Other { int score; uint32_t pos; int same; };
static void tally(const std::vector<Other> &o, bool bigger, int &bestScore, uint32_t bestPos, int &bestNum, int &secScore, int &secNum, int &isBest)
{
    uint32_t secPos = NONE;
    for (const Other &h : o) {
        const int w = bigger ? h.same : 1;
        const bool beatsBest = bigger ? h.score >= bestScore : h.score <= bestScore;
        const bool beatsSec = bigger ? h.score >= secScore : h.score <= secScore;
        if (beatsBest) {
            if (h.score == bestScore) { if (h.pos != bestPos) bestNum += w; }
            else { secScore = bestScore; secNum = bestNum; secPos = bestPos; bestScore = h.score; bestNum = w; bestPos = h.pos; isBest = 0; }
        } else if (beatsSec) {
            if (h.score == secScore) { if (h.pos != secPos) secNum += w; }
            else { secScore = h.score; secPos = h.pos; secNum = w; }
        }
    }
}

// the two records once both ends are described (the second half of both reference writers)
void SamWriter::pairRecords(const ReadSet &rs, uint32_t r1, EndView *e, int bestInsert, const std::string xa[2])
{
    (void)r1;
    const int type = opt.alignmentType;
    for (int k = 0; k < 2; k++) {
        EndView &a = e[k], &b = e[1 - k];
        int x0 = a.x0, x1 = a.x1;
        if (type == 4) x0 = x1 = -1;
        else if (type != 1 && type != 2) x1 = -1;
        Rec c;
        c.r = a.r;
        c.flag = 1 | (k == 0 ? 0x40 : 0x80);
        if (a.aligned && b.aligned) c.flag |= 2;
        if (!a.aligned) c.flag |= 4;
        if (!b.aligned) c.flag |= 8;
        if (a.aligned && a.negative) c.flag |= 0x10;
        if (b.aligned && b.negative) c.flag |= 0x20;
        c.tid = a.chr == 0 ? (b.chr == 0 ? -1 : (int)b.chr - 1) : (int)a.chr - 1;
        c.pos0 = a.tp == 0 ? (b.tp == 0 ? -1 : (long long)b.tp - 1) : (long long)a.tp - 1;
        c.mtid = b.chr == 0 ? (a.chr == 0 ? -1 : (int)a.chr - 1) : (int)b.chr - 1;
        c.mpos0 = b.tp == 0 ? (a.tp == 0 ? -1 : (long long)a.tp - 1) : (long long)b.tp - 1;
        if (bestInsert > 0) c.isize = a.tp > b.tp ? -(long long)(a.tp + (uint64_t)a.span - b.tp) : (long long)(b.tp + (uint64_t)b.span - a.tp);
        c.negative = a.negative;                                         // an end lost to read-through is still written in the strand it had (best_strand)
        if (a.aligned) {
            int mapq = a.mapq;
            if (!b.aligned) {                                            // "should not come to here": the mate was lost
                const int avg = rs.isFastq ? a.avgQual : 20;
                mapq = a.fromDP ? mapqFromDP(x0, a.score, (int)a.len * opt.dpMatchScore, avg, opt.maxMAPQ, opt.minMAPQ)
                                : (x0 == 1 ? tableScore(a.score, avg, opt.maxMAPQ, opt.minMAPQ) : opt.minMAPQ);
                if (a.trim) mapq = 0;
            }
            c.mapq = mapq;
            c.cigar = a.cigar.empty() ? std::to_string(a.len) + "M" : a.cigar;
            std::string t;
            if (opt.printMDNM) t += "\tNM:i:" + std::to_string(a.mis + a.gapExt);
            if (x0 >= 0) t += "\tX0:i:" + std::to_string(x0);
            if (x1 >= 0) t += "\tX1:i:" + std::to_string(x1);
            t += "\tXM:i:" + std::to_string(a.mis) + "\tXO:i:" + std::to_string(a.gapOpen) + "\tXG:i:" + std::to_string(a.gapExt);
            if (opt.printMDNM && !a.md.empty()) t += "\tMD:Z:" + a.md;
            if (!xa[k].empty()) t += "\tXA:Z:" + xa[k];
            c.tags = t;
        }
        emit(rs, c);
    }
}

void SamWriter::pairDP(const ReadSet &rs, uint32_t r1, const PairDpHit *H, size_t n, const BwtEndStats st[2])
{
    EndView e[2];
    e[0].r = r1; e[1].r = r1 + 1;
    e[0].len = rs.lengths[r1]; e[1].len = rs.lengths[r1 + 1];
    std::string xa[2];
    if (n == 0) { pairRecords(rs, r1, e, 0, xa); return; }
    auto posOf = [](const PairDpHit &h, int k) { return k ? h.pos2 : h.pos1; };
    auto scoreOf = [](const PairDpHit &h, int k) { return k ? h.score2 : h.score1; };
    // ---- which result is reported (outputRead2) ----
    size_t bi = 0;
    int minMis = 0, maxScore = 0;
    for (size_t i = 0; i < n; i++) {
        const bool un1 = H[i].pos1 == NONE, un2 = H[i].pos2 == NONE, proper = !un1 && !un2;
        const signed char misI = (signed char)(un1 && !un2 ? H[i].score2 : un2 && !un1 ? H[i].score1 : H[i].whichFromDP == 1 ? H[i].score1 : H[i].score2);
        const int scoreI = proper ? (H[i].whichFromDP == 1 ? H[i].score2 : H[i].score1) : -127;
        if (i == 0) { minMis = misI; maxScore = scoreI; continue; }
        if (!proper && (un1 != un2)) { if (misI < minMis) { bi = i; minMis = misI; maxScore = -127; } }
        else if (proper) { if ((misI == minMis && scoreI > maxScore) || misI < minMis) { bi = i; minMis = misI; maxScore = scoreI; } }
    }
    const PairDpHit &B = H[bi];
    const int which = B.whichFromDP;
    int bestInsert = 0;
    for (int k = 0; k < 2; k++) {
        EndView &a = e[k];
        a.pos = posOf(B, k);
        a.aligned = a.pos != NONE;
        if (!a.aligned) continue;
        a.negative = (k ? B.strand2 : B.strand1) == 2;
        a.fromDP = which == k;
        a.score = scoreOf(B, k);
        a.span = (int)a.len;
        if (a.fromDP) {
            std::string tmp;
            a.trim = boundaryCheckDP(a.len, a.pos, B.cigar, a.tp, a.chr, &tmp);
            const std::string &use = a.trim ? tmp : B.cigar;
            a.cigar = convertCigar(use);
            a.span = spanOfCigar(B.cigar);
            misInfoDP(rs, a.r, a.pos, a.negative, use, a.trim, a.md, a.mis, a.gapOpen, a.gapExt, a.avgQual);
        } else {
            a.trim = chrAndPosChecked(a.len, a.pos, a.tp, a.chr, &a.cigar);
            mdString(ix, &rs.codes[(size_t)a.r * rs.maxReadLength], &rs.quals[(size_t)a.r * rs.maxReadLength], a.len, a.pos, a.negative, a.score, a.trim, a.md,
                     a.avgQual);
            a.mis = a.score;
            if (a.trim && a.mis) { a.mis = 0; for (char c : a.md) a.mis += c > '9'; }
        }
    }
    if (e[0].aligned && e[1].aligned) bestInsert = B.insertSize;
    // read-through pairs lose the end with more mismatches (:4667-4690)
    if (e[0].aligned && e[1].aligned) {
        const uint64_t a1 = (uint64_t)e[0].pos + (e[0].trim > 0 ? e[0].trim : 0), a2 = (uint64_t)e[1].pos + (e[1].trim > 0 ? e[1].trim : 0);
        const uint64_t s1 = (uint64_t)(long long)e[0].span, s2 = (uint64_t)(long long)e[1].span;
        if ((B.strand1 == 1 && (a1 > a2 || a1 + s1 > a2 + s2)) || (B.strand1 == 2 && (a2 > a1 || a2 + s2 > a1 + s1))) {
            EndView &lost = e[0].mis <= e[1].mis ? e[1] : e[0];
            lost.aligned = false; lost.pos = NONE; lost.tp = 0; lost.chr = 0;
            bestInsert = 0;
        }
    }
    const bool both = e[0].aligned && e[1].aligned;
    // ---- pairs with the same / the next scores (:4692-4777) ----
    int bestPairNum = 0, bestPS1 = 0, bestPS2 = 0, secPairNum = 0, secPS1 = 0, secPS2 = 0;
    if (both) {
        bestPairNum = 1; bestPS1 = B.score1; bestPS2 = B.score2;
        for (size_t i = 0; i < n; i++) {
            if (i == bi || H[i].whichFromDP != which) continue;
            if (H[i].score1 == bestPS1 && H[i].score2 == bestPS2) { bestPairNum++; continue; }
            if (secPairNum == 0) { secPS1 = H[i].score1; secPS2 = H[i].score2; secPairNum = 1; }
            else if (which == 0) {                                       // first end from the DP: fewer mismatches of the second, then the larger score
                if (H[i].score2 < secPS2) { secPS1 = H[i].score1; secPS2 = H[i].score2; secPairNum = 1; }
                else if (H[i].score2 == secPS2 && H[i].score1 > secPS1) { secPS1 = H[i].score1; secPairNum = 1; }
                else if (H[i].score2 == secPS2 && H[i].score1 == secPS1) secPairNum++;
            } else if (which == 1) {
                if (H[i].score1 < secPS1) { secPS1 = H[i].score1; secPS2 = H[i].score2; secPairNum = 1; }
                else if (H[i].score1 == secPS1 && H[i].score2 > secPS2) { secPS2 = H[i].score2; secPairNum = 1; }
                else if (H[i].score1 == secPS1 && H[i].score2 == secPS2) secPairNum++;
            }
        }
    }
    int numSimilar = 0;
    if (both) {
        numSimilar = 1;
        for (size_t i = 0; i < n; i++) {
            if (i == bi || H[i].whichFromDP != which) continue;
            bool sim;
            if (which == 0) sim = H[i].score1 >= bestPS1 + opt.dpMismatchScore && H[i].score2 <= bestPS2 + 1;
            else if (which == 1) sim = H[i].score1 <= bestPS1 + 1 && H[i].score2 >= bestPS2 + opt.dpMismatchScore;
            else sim = H[i].score1 <= bestPS1 + 1 && H[i].score2 <= bestPS2 + 1;
            numSimilar += sim;
        }
    }
    // ---- per end: best / second-best hit numbers (:4838-5128) ----
    for (int k = 0; k < 2; k++) {
        EndView &a = e[k];
        int bestNum = 0, secNum = 0;
        a.bestScore = 0; a.secBestScore = 0; a.isBest = 1;
        const bool dpEnd = which == k;
        if (a.aligned) {
            a.bestScore = scoreOf(B, k); bestNum = 1;
            a.secBestScore = dpEnd ? 0 : 999;
            if (n > 1) {
                std::vector<Other> o;
                for (size_t i = 0; i < n; i++) if (i != bi && H[i].whichFromDP == which) o.push_back({scoreOf(H[i], k), posOf(H[i], k), H[i].numSameScore});
                // which end of the OTHER result came from the DP decides how its score is read; it equals `which` for every result kept
                tally(o, dpEnd, a.bestScore, posOf(B, k), bestNum, a.secBestScore, secNum, a.isBest);
            }
        }
        const BwtEndStats &s = st[k];
        if (k == 0 ? s.x0 > 0 : s.x0 > 1) {
            if (!dpEnd) {
                bestNum = s.x0; secNum = s.x1;
                if (s.mismatch < scoreOf(B, k)) a.isBest = 0;
            } else {
                const int x0Score = s.mismatch * opt.dpMismatchScore + ((int)a.len - s.mismatch) * opt.dpMatchScore;
                if (x0Score >= a.bestScore) {
                    bestNum = bestNum > s.x0 ? bestNum : s.x0;
                    secNum = secNum > s.x1 ? secNum : s.x1;
                    if (x0Score > a.bestScore) a.isBest = 0;
                }
            }
        }
        a.x0 = bestNum; a.x1 = secNum;
    }
    // ---- MAPQ (:5130-5188) ----
    if (both) {
        const int type = opt.alignmentType;
        if (type != 1 && type != 2) e[0].mapq = e[1].mapq = 255;
        else {
            if (opt.bwaLikeScore) {
                int op, sub = 0;
                const int lenB = (int)e[which == 0 ? 1 : 0].len;          // length of the BWT end
                if (which == 0) {
                    op = bestPS2 * opt.dpMismatchScore + (lenB - bestPS2) * opt.dpMatchScore + bestPS1;
                    if (secPairNum > 0) sub = secPS2 * opt.dpMismatchScore + (lenB - secPS2) * opt.dpMatchScore + secPS1;
                } else {
                    op = bestPS1 * opt.dpMismatchScore + (lenB - bestPS1) * opt.dpMatchScore + bestPS2;
                    if (secPairNum > 0) sub = secPS1 * opt.dpMismatchScore + (lenB - secPS1) * opt.dpMatchScore + secPS2;
                }
                bwaLikePair(e[0].x0, e[0].x1, e[1].x0, e[1].x1, gLogN, op, bestPairNum, sub, secPairNum, (int)e[0].len, (int)e[1].len, e[0].mapq, e[1].mapq);
            } else {
                int q[2];
                for (int k = 0; k < 2; k++) {
                    const EndView &a = e[k];
                    const int avg = rs.isFastq ? a.avgQual : 20;
                    q[k] = which == k ? mapqFromDP2(scoreOf(B, k), (int)a.len * opt.dpMatchScore, avg, a.x0, a.x1, a.bestScore, a.secBestScore, a.isBest, numSimilar,
                                                    opt.maxMAPQ, opt.minMAPQ)
                                      : mapqFromMis2(scoreOf(B, k), avg, a.x0, a.x1, a.isBest, numSimilar, opt.maxMAPQ, opt.minMAPQ);
                }
                e[0].mapq = e[1].mapq = q[0] > q[1] ? (int)(q[0] * 0.2 + q[1] * 0.8) : (int)(q[0] * 0.8 + q[1] * 0.2);
            }
            if (e[0].trim) e[0].mapq = 0;
            if (e[1].trim) e[1].mapq = 0;
        }
    }
    // ---- XA:Z (:5196-5253, :5361-5417) ----
    for (int k = 0; k < 2; k++) {
        if (!e[k].aligned || n <= 1) continue;
        for (size_t i = 0; i < n; i++) {
            if (i == bi || H[i].whichFromDP != which) continue;
            if (opt.alignmentType == 2 && (H[i].score1 != bestPS1 || H[i].score2 != bestPS2)) continue;
            uint64_t tp; uint32_t ch;
            chrAndPos(posOf(H[i], k), tp, ch);
            std::string &s = xa[k];
            s += ix.seqName[ch - 1];
            s.push_back(',');
            s.push_back((k ? H[i].strand2 : H[i].strand1) == 2 ? '-' : '+');
            s += std::to_string((unsigned long long)tp);
            s.push_back(',');
            if (H[i].whichFromDP != k) s += std::to_string(e[k].len) + "M," + std::to_string(scoreOf(H[i], k));
            else s += convertCigar(H[i].cigar) + "," + std::to_string(H[i].editdist);
            s.push_back(';');
        }
    }
    pairRecords(rs, r1, e, bestInsert, xa);
}

void keepFirstResultPerPosition(std::vector<DeepDpHit> &hits)
{
    if (hits.empty()) return;
    std::sort(hits.begin(), hits.end(), [](const DeepDpHit &x, const DeepDpHit &y) { return x.pos1 < y.pos1; });
    size_t u = 0;
    for (size_t q = 1; q < hits.size(); q++) if (hits[u].pos1 < hits[q].pos1) hits[++u] = hits[q];
    hits.resize(u + 1);
}

void SamWriter::pairDeepDP(const ReadSet &rs, uint32_t r1, const DeepDpHit *H, size_t n, const BwtEndStats st[2])
{
    EndView e[2];
    e[0].r = r1; e[1].r = r1 + 1;
    e[0].len = rs.lengths[r1]; e[1].len = rs.lengths[r1 + 1];
    std::string xa[2];
    if (n == 0) { pairRecords(rs, r1, e, 0, xa); return; }
    auto posOf = [](const DeepDpHit &h, int k) { return k ? h.pos2 : h.pos1; };
    auto scoreOf = [](const DeepDpHit &h, int k) { return k ? h.score2 : h.score1; };
    // ---- which result is reported (outputDeepDPResult2) ----
    size_t bi = 0;
    int maxScore = -127;
    for (size_t i = 0; i < n; i++) {
        const bool un1 = H[i].pos1 == NONE, un2 = H[i].pos2 == NONE;
        int sc;
        if (un1 && !un2) sc = H[i].score2;
        else if (un2 && !un1) sc = H[i].score1;
        else if (!un1 && !un2) sc = H[i].score1 + H[i].score2;
        else { if (i == 0) maxScore = -127; continue; }
        if (i == 0) maxScore = sc;
        else if (sc > maxScore) { bi = i; maxScore = sc; }
    }
    const DeepDpHit &B = H[bi];
    int bestInsert = 0;
    for (int k = 0; k < 2; k++) {
        EndView &a = e[k];
        a.pos = posOf(B, k);
        a.aligned = a.pos != NONE;
        if (!a.aligned) continue;
        a.negative = (k ? B.strand2 : B.strand1) == 2;
        a.fromDP = true;
        a.score = scoreOf(B, k);
        const std::string &cig = k ? B.cigar2 : B.cigar1;
        std::string tmp;
        a.trim = boundaryCheckDP(a.len, a.pos, cig, a.tp, a.chr, &tmp);
        const std::string &use = a.trim ? tmp : cig;
        a.cigar = convertCigar(use);
        misInfoDP(rs, a.r, a.pos, a.negative, use, a.trim, a.md, a.mis, a.gapOpen, a.gapExt, a.avgQual);
        a.span = spanOfCigar(cig);
    }
    if (e[0].aligned && e[1].aligned) {
        bestInsert = B.insertSize;
        const uint64_t a1 = (uint64_t)e[0].pos + (e[0].trim > 0 ? e[0].trim : 0), a2 = (uint64_t)e[1].pos + (e[1].trim > 0 ? e[1].trim : 0);
        const uint64_t s1 = (uint64_t)(long long)e[0].span, s2 = (uint64_t)(long long)e[1].span;
        if ((B.strand1 == 1 && (a1 > a2 || a1 + s1 > a2 + s2)) || (B.strand1 == 2 && (a2 > a1 || a2 + s2 > a1 + s1))) {
            EndView &lost = e[0].mis <= e[1].mis ? e[1] : e[0];
            lost.aligned = false; lost.pos = NONE; lost.tp = 0; lost.chr = 0;
            bestInsert = 0;
        }
    }
    const bool both = e[0].aligned && e[1].aligned;
    int bestPairNum = 0, bestPairScore = 0, secPairScore = 0, numSimilar = 0;
    if (both) {
        bestPairNum = 1; bestPairScore = B.score1 + B.score2; numSimilar = 1;
        for (size_t i = 0; i < n; i++) {
            if (i == bi) continue;
            const int s = H[i].score1 + H[i].score2;
            if (s == bestPairScore) bestPairNum++;
            else if (s > secPairScore) secPairScore = s;
            if (H[i].score1 >= B.score1 + opt.dpMismatchScore && H[i].score2 >= B.score2 + opt.dpMismatchScore) numSimilar++;
        }
    }
    for (int k = 0; k < 2; k++) {
        EndView &a = e[k];
        int bestNum = 0, secNum = 0;
        a.bestScore = 0; a.secBestScore = 0; a.isBest = 1;
        if (a.aligned) {
            a.bestScore = scoreOf(B, k); bestNum = 1;
            if (n > 1) {
                std::vector<Other> o;
                for (size_t i = 0; i < n; i++) if (i != bi) o.push_back({scoreOf(H[i], k), posOf(H[i], k), k ? H[i].numSame2 : H[i].numSame1});
                tally(o, true, a.bestScore, posOf(B, k), bestNum, a.secBestScore, secNum, a.isBest);
            }
        }
        const BwtEndStats &s = st[k];
        if (k == 0 ? s.x0 > 0 : s.x0 > 1) {
            const int x0Score = s.mismatch * opt.dpMismatchScore + ((int)a.len - s.mismatch) * opt.dpMatchScore;
            if (x0Score >= a.bestScore) {
                bestNum = bestNum > s.x0 ? bestNum : s.x0;
                secNum = secNum > s.x1 ? secNum : s.x1;
                if (x0Score > a.bestScore) a.isBest = 0;
            }
        }
        a.x0 = bestNum; a.x1 = secNum;
    }
    const int type = opt.alignmentType;
    if (type == 1 || type == 2) {
        if (opt.bwaLikeScore)
            bwaLikePair(e[0].x0, e[0].x1, e[1].x0, e[1].x1, gLogN, bestPairScore, bestPairNum, secPairScore, (int)n - bestPairNum, (int)e[0].len, (int)e[1].len, e[0].mapq,
                        e[1].mapq);
        else {
            int q[2];
            for (int k = 0; k < 2; k++) {
                const EndView &a = e[k];
                q[k] = mapqFromDP2(scoreOf(B, k), (int)a.len * opt.dpMatchScore, rs.isFastq ? a.avgQual : 20, a.x0, a.x1, a.bestScore, a.secBestScore, a.isBest, numSimilar,
                                   opt.maxMAPQ, opt.minMAPQ);
            }
            e[0].mapq = e[1].mapq = q[0] > q[1] ? (int)(q[0] * 0.2 + q[1] * 0.8) : (int)(q[0] * 0.8 + q[1] * 0.2);
        }
        if (e[0].trim) e[0].mapq = 0;
        if (e[1].trim) e[1].mapq = 0;
    } else e[0].mapq = e[1].mapq = 255;
    for (int k = 0; k < 2; k++) {
        if (!e[k].aligned || n <= 1) continue;
        for (size_t i = 0; i < n; i++) {
            if (i == bi) continue;
            if (type == 2 && H[i].score1 + H[i].score2 < bestPairScore) continue;
            uint64_t tp; uint32_t ch;
            chrAndPos(posOf(H[i], k), tp, ch);
            std::string &s = xa[k];
            s += ix.seqName[ch - 1];
            s.push_back(',');
            s.push_back((k ? H[i].strand2 : H[i].strand1) == 2 ? '-' : '+');
            s += std::to_string((unsigned long long)tp) + "," + convertCigar(k ? H[i].cigar2 : H[i].cigar1) + "," + std::to_string(k ? H[i].editdist2 : H[i].editdist1) + ";";
        }
    }
    pairRecords(rs, r1, e, bestInsert, xa);
}

void SamWriter::unpairedDP(const ReadSet &rs, uint32_t r1, const DpHit *hits1, size_t n1, const DpHit *hits2, size_t n2, int cutoff)
{
    const int type = opt.alignmentType;
    const DpHit *H[2] = {hits1, hits2};
    const size_t N[2] = {n1, n2};
    struct End {
        bool aligned = false, negative = false;
        uint64_t tp = 0; uint32_t chr = 0;
        int trim = 0, deletedEnd = 0, mis = 0, gapOpen = 0, gapExt = 0, avg = 20, mapq = 0, x0 = 0, x1 = 0;
        std::string cigar, md, xa;
        uint32_t len = 0, r = 0;
    } e[2];
    for (int k = 0; k < 2; k++) {
        End &a = e[k];
        a.r = r1 + (uint32_t)k; a.len = rs.lengths[a.r];
        const DpHit *h = H[k];
        const size_t n = N[k];
        size_t best = 0;
        int bestScore = 0, bestNum = 0;
        if (n > 0) {
            bestScore = h[0].score; bestNum = 1;
            for (size_t i = 1; i < n; i++) {
                if (h[i].score > bestScore) { best = i; bestScore = h[i].score; bestNum = 1; }
                else if (h[i].score == bestScore) bestNum++;
            }
        }
        int x1t1 = 0, x1t2 = 0, secBest = -9999;
        const int thres = (int)(0.7 * bestScore);
        for (size_t i = 0; i < n; i++)
            if (h[i].score < bestScore) {
                if (h[i].score > secBest) secBest = h[i].score;
                if (h[i].score >= thres) x1t1++; else x1t2++;
            }
        a.x0 = bestNum;
        a.x1 = (type == 3 || type == 4) ? -1 : x1t1 + x1t2;
        if (n > 0 && (type != 3 || bestNum == 1)) {
            a.aligned = true;
            a.negative = h[best].strand == 2;
            if (h[best].fromDP) {
                std::string tmp;
                a.trim = boundaryCheckDP(a.len, h[best].pos, h[best].cigar, a.tp, a.chr, &tmp);
                a.cigar = a.trim ? convertCigar(tmp) : convertCigar(h[best].cigar, &a.deletedEnd);
                misInfoDP(rs, a.r, h[best].pos, a.negative, a.trim ? tmp : h[best].cigar, a.trim, a.md, a.mis, a.gapOpen, a.gapExt, a.avg);
            } else {
                a.trim = chrAndPosChecked(a.len, h[best].pos, a.tp, a.chr, &a.cigar);
                mdString(ix, &rs.codes[(size_t)a.r * rs.maxReadLength], &rs.quals[(size_t)a.r * rs.maxReadLength], a.len, h[best].pos, a.negative, h[best].editdist, a.trim,
                         a.md, a.avg);
                a.mis = 0;
                for (char c : a.md) a.mis += c > '9';
            }
            if (type == 3 || type == 4) a.mapq = 255;
            else {
                a.mapq = mapqSingleDP((int)a.len * opt.dpMatchScore, rs.isFastq ? a.avg : 20, bestNum, x1t1, x1t2, bestScore, secBest, cutoff);
                if (!opt.bwaLikeScore) a.mapq >>= 1;
                if (a.mapq < opt.minMAPQ) a.mapq = opt.minMAPQ;
            }
            // the first end loses its MAPQ to a boundary trim in every mode, the second only where a MAPQ is computed (:3127 vs :3186)
            if (a.trim && (k == 0 || !(type == 3 || type == 4))) a.mapq = 0;
        }
        if (type == 1 || type == 2)
            for (size_t i = 0; i < n; i++) {
                if (a.aligned && i == best) continue;
                if (type == 2 && h[i].score < bestScore) continue;
                uint64_t tp; uint32_t ch;
                chrAndPos(h[i].pos, tp, ch);
                a.xa += ix.seqName[ch - 1];
                a.xa.push_back(',');
                a.xa.push_back(h[i].strand == 2 ? '-' : '+');
                a.xa += std::to_string((unsigned long long)tp) + "," + convertCigar(h[i].cigar) + "," + std::to_string(h[i].editdist) + ";";
            }
    }
    // the XA:Z buffer is shared and only rebuilt in the all-valid / all-best modes: nothing to carry over in the others
    for (int k = 0; k < 2; k++) {
        End &a = e[k], &b = e[1 - k];
        Rec c;
        c.r = a.r;
        c.flag = 1 | (k == 0 ? 0x40 : 0x80);
        if (!a.aligned) c.flag |= 4;
        if (!b.aligned) c.flag |= 8;
        if (a.aligned && a.negative) c.flag |= 0x10;
        if (b.aligned && b.negative) c.flag |= 0x20;
        c.tid = a.chr == 0 ? (b.chr == 0 ? -1 : (int)b.chr - 1) : (int)a.chr - 1;
        c.pos0 = a.tp == 0 ? (b.tp == 0 ? -1 : (long long)b.tp - 1) : (long long)a.tp - 1;
        c.mtid = b.chr == 0 ? (a.chr == 0 ? -1 : (int)a.chr - 1) : (int)b.chr - 1;
        c.mpos0 = b.tp == 0 ? (a.tp == 0 ? -1 : (long long)a.tp - 1) : (long long)b.tp - 1;
        if (e[0].chr > 0 && e[0].chr == e[1].chr)
            c.isize = b.tp > a.tp ? (long long)(b.tp - (uint64_t)b.deletedEnd + b.len - a.tp) : -(long long)(a.tp - (uint64_t)a.deletedEnd + a.len - b.tp);
        if (a.aligned) {
            c.negative = a.negative;
            c.mapq = a.mapq;
            c.cigar = a.cigar.empty() ? std::to_string(a.len) + "M" : a.cigar;
            std::string t;
            if (opt.printMDNM) t += "\tNM:i:" + std::to_string(a.mis + a.gapExt);
            t += "\tX0:i:" + std::to_string(a.x0);
            if (a.x1 >= 0) t += "\tX1:i:" + std::to_string(a.x1);
            t += "\tXM:i:" + std::to_string(a.mis) + "\tXO:i:" + std::to_string(a.gapOpen) + "\tXG:i:" + std::to_string(a.gapExt);
            if (opt.printMDNM && !a.md.empty()) t += "\tMD:Z:" + a.md;
            if (!a.xa.empty()) t += "\tXA:Z:" + a.xa;
            c.tags = t;
        }
        emit(rs, c);
    }
}

}  // namespace s3host
