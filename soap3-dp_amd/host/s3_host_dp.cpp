// s3_host_dp.cpp -- SAM records of DP results (host side of the drop-in program).  Mirrors, for single reads:
//   SingleDPOutputSAMAPI            BGS-IO.cpp:5857-6118   best / second-best bookkeeping, XA:Z, MAPQ, tags
//   getChrAndPosWithBoundaryCheckDP BGS-IO.cpp:2009, BoundaryCheckDP :1807-1975   trimming of alignments that cross a sequence end
//   convertToCigarStr               PE.cpp:421-481         engine CIGAR ('m' = mismatch run) -> SAM CIGAR
//   getMisInfoForDP                 PE.cpp:499-690         MD:Z, mismatch / gap counts, average mismatch quality
//   getMapQualScoreForSingleDP      BGS-IO.cpp:2370-2413   (tables BGS-IO.cpp:42-45)
// Pinned: tests/golden/dp1 holds the reference's own output for given DP results (tests/test_host_dp_sam_cpu.py).
#include <ctype.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "s3_host.h"

namespace s3host {

static const float kPenaltyAvgMisQual[41] = {3, 2.85f, 2.71f, 2.57f, 2.43f, 2.3f, 2.17f, 2.04f, 1.92f, 1.8f, 1.69f, 1.58f, 1.47f, 1.37f, 1.27f, 1.17f, 1.08f, 0.99f,
                                             0.91f, 0.83f, 0.75f, 0.68f, 0.61f, 0.54f, 0.48f, 0.42f, 0.37f, 0.32f, 0.27f, 0.23f, 0.19f, 0.15f, 0.12f, 0.09f, 0.07f,
                                             0.05f, 0.03f, 0.02f, 0.01f, 0, 0};
static float penaltyRatioX1(int x1)
{
    static const struct { int upto; float v; } steps[] = {{0, 1}, {1, 0.5f}, {2, 0.33f}, {3, 0.25f}, {4, 0.2f}, {5, 0.17f}, {6, 0.14f}, {7, 0.13f}, {8, 0.11f}, {9, 0.1f},
                                                          {10, 0.09f}, {12, 0.08f}, {14, 0.07f}, {17, 0.06f}, {21, 0.05f}, {27, 0.04f}, {39, 0.03f}, {65, 0.02f},
                                                          {100, 0.01f}};
    if (x1 > 100) x1 = 100;
    for (const auto &s : steps) if (x1 <= s.upto) return s.v;
    return 0.01f;
}

std::string SamWriter::convertCigar(const std::string &sp, int *deletedEnd)
{
    std::string out;
    int cur = 0, m = 0;
    const size_t n = sp.size();
    if (deletedEnd) *deletedEnd = 0;
    for (size_t i = 0; i < n; i++) {
        const char c = sp[i];
        if (isdigit((unsigned char)c)) { cur = cur * 10 + (c - '0'); continue; }
        if (c == 'M' || c == 'm') { m += cur; cur = 0; continue; }
        if (c == 'D' && ((out.empty() && m == 0) || i == n - 1)) {               // a deletion in front or at the very end is dropped (the count stays pending, as there)
            if (i == n - 1 && deletedEnd) *deletedEnd = m;                       // the reference stores the pending MATCH count here, not the deletion's length (PE.cpp:452)
            continue;
        }
        if (c == 'D' || c == 'I' || c == 'S') {
            if (m > 0) { out += std::to_string(m) + "M"; m = 0; }
            out += std::to_string(cur);
            out.push_back(c);
            cur = 0;
        }
    }
    if (m > 0) out += std::to_string(m) + "M";
    return out;
}

// BoundaryCheckDP: 0 = fits; < 0 = right part trimmed (soft clip appended); > 0 = left part trimmed, the position moves to the next sequence
int SamWriter::boundaryCheckDP(uint32_t readLen, uint32_t pacPos, const std::string &cigar, uint64_t &tp, uint32_t &chr, std::string *newCigar) const
{
    uint32_t segEnd = chrAndPos(pacPos, tp, chr);
    const uint32_t chrEnd = ix.seqEnd[chr - 1];
    if ((uint64_t)pacPos + readLen * 2ull <= (uint64_t)chrEnd + 1 && (uint64_t)pacPos + readLen * 2ull <= (uint64_t)segEnd + 1) return 0;
    segEnd = chrEnd < segEnd ? chrEnd : segEnd;
    std::string left, right;
    int leftLen = 0, rightLen = 0, rightOff = 0, leftS = 0, rightS = 0;
    uint32_t ref = pacPos;
    const char *p = cigar.c_str();
    while (*p) {
        int num = 0;
        while (*p && *p <= '9') { num = num * 10 + (*p - '0'); p++; }
        const char op = *p;
        if (!op) break;
        p++;
        if (op == 'S') {
            if (ref <= segEnd) { left += std::to_string(num) + "S"; leftS += num; }
            else { right += std::to_string(num) + "S"; rightS += num; }
        } else if (op == 'M' || op == 'm') {
            if (ref > segEnd) { rightLen += num; right += std::to_string(num) + op; }
            else if (ref + (uint32_t)num <= segEnd + 1) { leftLen += num; left += std::to_string(num) + op; }
            else {
                const int a = (int)(segEnd - ref + 1);
                leftLen += a; left += std::to_string(a) + op;
                rightLen += num - a; right += std::to_string(num - a) + op;
            }
            ref += (uint32_t)num;
        } else if (op == 'D') {
            if (ref > segEnd) { if (right.empty()) rightOff = num; else right += std::to_string(num) + "D"; }
            else if (ref + (uint32_t)num <= segEnd + 1) left += std::to_string(num) + "D";
            else {
                const int a = (int)(segEnd - ref + 1);
                left += std::to_string(a) + "D";
                if (right.empty()) rightOff = num - a; else right += std::to_string(num - a) + "D";
            }
            ref += (uint32_t)num;
        } else if (op == 'I') {
            if (ref == chrEnd + 1) { /* exactly at the boundary: dropped on both sides */ }
            else if (ref <= segEnd) { leftLen += num; left += std::to_string(num) + "I"; }
            else { rightLen += num; right += std::to_string(num) + "I"; }
        }
    }
    if (!leftLen || !rightLen) return 0;
    if (leftLen >= rightLen) {
        if (newCigar) *newCigar = left + std::to_string((int)readLen - leftLen - leftS) + "S";
        return -((int)readLen - leftLen - leftS);
    }
    if (newCigar) *newCigar = std::to_string((int)readLen - rightLen - rightS) + "S" + right;
    const uint32_t corrected = segEnd + 1 + (uint32_t)rightOff;
    if (corrected > pacPos) chrAndPos(corrected, tp, chr);
    return (int)readLen - rightLen - rightS;
}

// getMisInfoForDP: MD string along the engine CIGAR; counts of mismatches, gap opens, gap bases; average quality at the mismatches
int SamWriter::misInfoDP(const ReadSet &rs, uint32_t r, uint32_t pos, bool negative, const std::string &sp, int trim, std::string &md, int &numMis, int &gapOpen,
                         int &gapExt, int &avgQual) const
{
    uint32_t qlen = rs.lengths[r];
    const uint8_t *query = &rs.codes[(size_t)r * rs.maxReadLength];
    const char *qual = &rs.quals[(size_t)r * rs.maxReadLength];
    (void)query;
    // a trimmed left part moves the start on the reference; the quality array is NOT moved there (only the unused query pointer is)
    if (trim > 0) pos += (uint32_t)trim;
    (void)qlen;
    avgQual = 20;                                                         // DEFAULT_QUAL_VALUE, PE.h:27
    numMis = gapOpen = gapExt = 0;
    md.clear();
    static const char dna[] = "ACGT";
    auto refChar = [&](uint64_t t) -> char { return t < ix.n ? dna[(ix.pac[t >> 4] >> (2 * (15 - (t & 15)))) & 3u] : 'A'; };
    double sum = 0;
    int cur = 0, match = 0, qPos = 0;
    uint64_t tPos = pos;
    const size_t l = sp.size();
    for (size_t i = 0; i < l; i++) {
        const char c = sp[i];
        if (isdigit((unsigned char)c)) { cur = cur * 10 + (c - '0'); continue; }
        switch (c) {
        case 'M': match += cur; qPos += cur; tPos += (uint64_t)cur; cur = 0; break;
        case 'm':
            md += std::to_string(match);
            md.push_back(refChar(tPos));
            sum += qual[qPos];                                           // the read's qualities in FILE order, indexed by the alignment column, as there
            for (int j = 1; j < cur; j++) { md.push_back('0'); md.push_back(refChar(tPos + (uint64_t)j)); sum += qual[qPos + j]; }
            qPos += cur; tPos += (uint64_t)cur; numMis += cur; match = 0; cur = 0;
            break;
        case 'I': qPos += cur; gapOpen++; gapExt += cur; cur = 0; break;
        case 'D':
            if (i == l - 1) break;                                       // a final deletion is ignored
            md += std::to_string(match);
            md.push_back('^');
            for (int j = 0; j < cur; j++) md.push_back(refChar(tPos + (uint64_t)j));
            tPos += (uint64_t)cur; gapOpen++; gapExt += cur; match = 0; cur = 0;
            break;
        case 'S': qPos += cur; cur = 0; break;
        }
    }
    md += std::to_string(match);
    if (numMis > 0) avgQual = (int)(sum / numMis);
    return (int)md.size();
}

// getMapQualScoreForSingleDP (BGS-IO.cpp:2370-2412), float arithmetic as there
int SamWriter::mapqSingleDP(int maxDP, int avgMismatchQual, int x0, int x1t1, int x1t2, int bestDP, int secDP, int cutoff) const
{
    if (opt.bwaLikeScore) return bwaLikeSingle(x0, x1t1 + x1t2, gLogN);
    if (x0 > 1 || x1t1 > 0) return opt.minMAPQ;
    float R1 = 1.0f;
    if (x1t2 > 0) R1 = (float)(1.0 - ((float)(secDP - cutoff)) / (0.7 * bestDP - cutoff));
    const float R2 = penaltyRatioX1(x1t1 + x1t2);
    const float R3 = ((float)(bestDP - cutoff)) / (maxDP - cutoff);
    const int aq = avgMismatchQual < 0 ? 0 : avgMismatchQual > 40 ? 40 : avgMismatchQual;
    const int mapq = (int)(opt.maxMAPQ * R1 * R2 * R3 - kPenaltyAvgMisQual[aq]);
    return mapq < opt.minMAPQ ? opt.minMAPQ : mapq;
}

void SamWriter::singleDP(const ReadSet &rs, uint32_t r, const DpHit *hits, size_t n, int singleDPcutoff)
{
    if (n == 0 || hits[0].pos == 0xFFFFFFFFu) { unaligned(rs, r); return; }
    const uint32_t readLen = rs.lengths[r];
    const int type = opt.alignmentType;
    size_t best = 0;
    uint32_t bestScore = (uint32_t)hits[0].score, secBestScore = 0;       // unsigned, as the reference keeps them
    int bestHitNum = 1;
    for (size_t k = 1; k < n; k++) {
        const uint32_t cur = (uint32_t)hits[k].score;
        if (cur >= bestScore) {
            if (cur == bestScore) bestHitNum++;
            else { secBestScore = bestScore; bestScore = cur; bestHitNum = 1; best = k; }
        } else if (cur > secBestScore) secBestScore = cur;
    }
    uint64_t bestTP = 0;
    uint32_t bestChr = 0;
    int boundTrim = boundaryCheckDP(readLen, hits[best].pos, hits[best].cigar, bestTP, bestChr, nullptr);
    if (boundTrim) {
        // "re-find best alignment" (BGS-IO.cpp:5954-5987): the sentinels -9998 / -9999 live in unsigned variables there, so no other hit can
        // replace the stored one; what remains is the changed score bookkeeping
        bestScore = (uint32_t)-9998; secBestScore = (uint32_t)-9999;
        boundTrim = boundaryCheckDP(readLen, hits[best].pos, hits[best].cigar, bestTP, bestChr, nullptr);
    }
    int x1t1 = 0, x1t2 = 0;
    const double thres7 = 0.7 * (double)bestScore;
    const int suboptThres = thres7 >= 2147483648.0 ? (int)0x80000000u : (int)thres7;      // (int) of an out-of-range double: x86 gives INT_MIN
    std::string xa;
    if (n > 1 && (type == 1 || type == 2) && !boundTrim) {
        for (size_t k = 0; k < n; k++) {
            if (k == best) continue;
            if (type == 2 && (uint32_t)hits[k].score < bestScore) continue;
            uint64_t tp; uint32_t ch;
            if (boundaryCheckDP(readLen, hits[k].pos, hits[k].cigar, tp, ch, nullptr)) continue;
            xa += ix.seqName[ch - 1];
            xa.push_back(',');
            xa.push_back(hits[k].strand == 2 ? '-' : '+');
            xa += std::to_string((unsigned long long)tp) + "," + convertCigar(hits[k].cigar) + "," + std::to_string(hits[k].editdist) + ";";
            if ((uint32_t)hits[k].score < bestScore) { if ((uint32_t)hits[k].score >= (uint32_t)suboptThres) x1t1++; else x1t2++; }
        }
    }
    const int secBestHitNum = x1t1 + x1t2;
    const bool negative = hits[best].strand == 2;
    std::string newSp;
    boundTrim = boundaryCheckDP(readLen, hits[best].pos, hits[best].cigar, bestTP, bestChr, &newSp);
    const std::string &useSp = boundTrim ? newSp : hits[best].cigar;
    const std::string cigar = convertCigar(useSp);
    std::string md;
    int numMis = 0, gapOpen = 0, gapExt = 0, avgQual = 20;
    misInfoDP(rs, r, hits[best].pos, negative, useSp, boundTrim, md, numMis, gapOpen, gapExt, avgQual);
    const int editDist = gapExt + numMis;
    const int mapq = mapqSingleDP((int)readLen * opt.dpMatchScore, rs.isFastq ? avgQual : 20, bestHitNum, x1t1, x1t2, (int)bestScore, (int)secBestScore, singleDPcutoff);
    std::string line = rs.names[r];
    line.push_back('\t'); line += std::to_string(negative ? 16 : 0);
    line.push_back('\t'); line += ix.seqName[bestChr - 1];
    line.push_back('\t'); line += std::to_string((unsigned long long)bestTP);
    line.push_back('\t'); line += std::to_string(mapq);
    line.push_back('\t'); line += cigar;
    line += "\t*\t0\t0\t";
    seqAndQual(rs, r, negative, line);
    line += "\tRG:Z:" + opt.readGroup;
    if (opt.printMDNM) line += "\tNM:i:" + std::to_string(editDist);
    line += "\tX0:i:" + std::to_string(bestHitNum) + "\tX1:i:" + std::to_string(secBestHitNum) + "\tXM:i:" + std::to_string(numMis) + "\tXO:i:" +
            std::to_string(gapOpen) + "\tXG:i:" + std::to_string(gapExt);
    if (opt.printMDNM && !md.empty()) line += "\tMD:Z:" + md;
    if (n > 1 && (type == 1 || type == 2) && !xa.empty()) line += "\tXA:Z:" + xa;
    line.push_back('\n');
    put(line);
}

}  // namespace s3host
