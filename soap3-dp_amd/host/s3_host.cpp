// s3_host.cpp -- index files, read files and SAM records for the drop-in `soap3-dp` program (see s3_host.h).
#include "s3_host.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <algorithm>

namespace s3host {

// ------------------------------------------------------------------------------------------------
// index files (formats: SURVEY.md 5.9)
// ------------------------------------------------------------------------------------------------
static bool slurpWords(const std::string &name, size_t skipWords, std::vector<uint32_t> &out, std::vector<uint32_t> *header, std::string &err)
{
    FILE *f = fopen(name.c_str(), "rb");
    if (!f) { err = "cannot open " + name; return false; }
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (bytes < (long)(skipWords * 4)) { fclose(f); err = "file too short: " + name; return false; }
    if (header) {
        header->resize(skipWords);
        if (skipWords && fread(header->data(), 4, skipWords, f) != skipWords) { fclose(f); err = "short read " + name; return false; }
    } else fseek(f, (long)(skipWords * 4), SEEK_SET);
    const size_t words = ((size_t)bytes - skipWords * 4) / 4;
    out.resize(words);
    if (words && fread(out.data(), 4, words, f) != words) { fclose(f); err = "short read " + name; return false; }
    fclose(f);
    return true;
}

// cumulative 13-mer counts with the text padded by 12 'A' (2bwt-flex/LTConstruct.c:45-101)
void buildLookupTable(const std::vector<uint32_t> &pac, uint32_t n, bool reversed, std::vector<uint32_t> &table)
{
    const int K = 13;
    const uint32_t mask = (1u << (2 * K)) - 1u;
    table.assign((size_t)1 << (2 * K), 0);
    auto ch = [&](uint64_t p) -> uint32_t {
        if (p >= n) return 0u;
        const uint64_t q = reversed ? (uint64_t)n - 1 - p : p;
        return (pac[q >> 4] >> (2 * (15 - (q & 15)))) & 3u;
    };
    uint32_t w = 0;
    for (int k = 0; k < K - 1; k++) w = (w << 2) | ch((uint64_t)k);
    for (uint64_t p = 0; p < n; p++) {
        w = ((w << 2) | ch(p + K - 1)) & mask;
        table[w]++;
    }
    uint32_t acc = 0;
    for (size_t k = 0; k < table.size(); k++) { acc += table[k]; table[k] = acc; }
}

bool loadIndex(const std::string &prefix, HostIndex &ix, std::string &err)
{
    std::vector<uint32_t> h1, h2;
    if (!slurpWords(prefix + ".bwt", 5, ix.bwt, &h1, err)) return false;
    if (!slurpWords(prefix + ".fmv.gpu", 5, ix.occ, &h2, err)) return false;
    if (h1 != h2) { err = ".bwt and .fmv.gpu headers differ"; return false; }
    ix.inverseSa0 = h1[0];
    ix.cum[0] = 0;
    for (int k = 1; k <= 4; k++) ix.cum[k] = h1[k];
    ix.n = h1[4];
    if (!slurpWords(prefix + ".rev.bwt", 5, ix.rbwt, &h1, err)) return false;
    if (!slurpWords(prefix + ".rev.fmv.gpu", 5, ix.rocc, &h2, err)) return false;
    if (h1 != h2) { err = ".rev.bwt and .rev.fmv.gpu headers differ"; return false; }
    ix.rInverseSa0 = h1[0];
    ix.rcum[0] = 0;
    for (int k = 1; k <= 4; k++) ix.rcum[k] = h1[k];
    std::vector<uint32_t> hs;
    if (!slurpWords(prefix + ".sa", 6, ix.sa, &hs, err)) return false;
    if (hs[5] != 1) { err = "a full suffix array (SaValueFreq=1) is required"; return false; }
    if (ix.sa.size() < (size_t)ix.n + 1) { err = ".sa too short"; return false; }
    ix.sa.resize((size_t)ix.n + 1);
    ix.sa[0] = 0xFFFFFFFFu;                                            // BWT.c:286
    // packed text: bytes MSB first + trailing length byte -> words (TextConverter.c:666-720)
    {
        FILE *f = fopen((prefix + ".pac").c_str(), "rb");
        if (!f) { err = "cannot open " + prefix + ".pac"; return false; }
        fseek(f, 0, SEEK_END);
        const long bytes = ftell(f);
        fseek(f, 0, SEEK_SET);
        std::vector<unsigned char> raw((size_t)bytes);
        if (fread(raw.data(), 1, (size_t)bytes, f) != (size_t)bytes) { fclose(f); err = "short read .pac"; return false; }
        fclose(f);
        ix.dnaLength = (uint32_t)((bytes - 2) * 4 + raw[(size_t)bytes - 1]);
        if (ix.dnaLength != ix.n) { err = ".pac length does not match the BWT"; return false; }
        const size_t words = ((size_t)ix.n + 63) / 64 * 4 + 8;
        ix.pac.assign(words, 0);
        for (size_t b = 0; b + 1 < (size_t)bytes; b++) ix.pac[b >> 2] |= (uint32_t)raw[b] << (8 * (3 - (b & 3)));
    }
    // lookup tables: load if present (256 MiB each), otherwise rebuild from the text
    {
        std::vector<uint32_t> hdr;
        std::string e2;
        if (slurpWords(prefix + ".lkt", 1, ix.lkt, &hdr, e2) && hdr[0] == 13 && ix.lkt.size() == ((size_t)1 << 26)) {
            if (!slurpWords(prefix + ".rev.lkt", 1, ix.rlkt, &hdr, err)) return false;
        } else {
            buildLookupTable(ix.pac, ix.n, false, ix.lkt);
            buildLookupTable(ix.pac, ix.n, true, ix.rlkt);
        }
    }
    // annotation
    {
        FILE *f = fopen((prefix + ".ann").c_str(), "r");
        if (!f) { err = "cannot open " + prefix + ".ann"; return false; }
        uint32_t dl = 0, seed = 0;
        if (fscanf(f, "%u %d %u\n", &dl, &ix.numOfSeq, &seed) != 3) { fclose(f); err = "bad .ann"; return false; }
        char line[1024];
        for (int i = 0; i < ix.numOfSeq; i++) {
            uint32_t gi = 0, st = 0, len = 0;
            int amb = 0;
            if (fscanf(f, "%u ", &gi) != 1 || !fgets(line, sizeof line, f)) { fclose(f); err = "bad .ann entry"; return false; }
            if (fscanf(f, "%u %u %d\n", &st, &len, &amb) != 3) { fclose(f); err = "bad .ann entry"; return false; }
            std::string nm(line);
            size_t cut = nm.find_first_of(" \t\r\n");                    // SAMOutputHeaderConstruct, SAM.cpp:98-112
            if (cut != std::string::npos) nm.resize(cut);
            ix.seqName.push_back(nm);
            ix.seqStart.push_back(st);
            ix.seqEnd.push_back(st + len - 1);
        }
        fclose(f);
    }
    // translate + ambiguity grid (HSP.c:429-486)
    {
        FILE *f = fopen((prefix + ".tra").c_str(), "r");
        if (!f) { err = "cannot open " + prefix + ".tra"; return false; }
        uint32_t dl = 0, grid = 0;
        int ns = 0, removed = 0;
        if (fscanf(f, "%u %d %d %u\n", &dl, &ns, &removed, &grid) != 4 || ns != ix.numOfSeq) { fclose(f); err = "bad .tra"; return false; }
        ix.numOfRemovedSegment = (uint32_t)removed;
        ix.ambiguityMap.resize(grid);
        for (uint32_t j = 0; j < grid; j++) if (fscanf(f, "%u\n", &ix.ambiguityMap[j]) != 1) { fclose(f); err = "bad .tra grid"; return false; }
        ix.translate.resize((size_t)ns + removed);
        for (size_t j = 0; j < ix.translate.size(); j++)
            if (fscanf(f, "%u %u %u\n", &ix.translate[j].startPos, &ix.translate[j].chrID, &ix.translate[j].correction) != 3) { fclose(f); err = "bad .tra entry"; return false; }
        ix.actualStart.resize(ns);
        ix.actualEnd.resize(ns);
        for (int j = 0; j < ns; j++) {
            uint32_t a = 0, l = 0;
            if (fscanf(f, "%u %u\n", &a, &l) != 2) { a = ix.seqStart[j]; l = ix.seqEnd[j] - ix.seqStart[j]; }
            ix.actualStart[j] = a;
            ix.actualEnd[j] = a + l;
        }
        fclose(f);
    }
    return true;
}

s3_index_desc HostIndex::desc() const
{
    s3_index_desc d;
    memset(&d, 0, sizeof d);
    d.fwd.bwt = bwt.data(); d.fwd.bwtWords = bwt.size(); d.fwd.occ = occ.data(); d.fwd.occRecs = occ.size() / 4; d.fwd.inverseSa0 = inverseSa0;
    d.rev.bwt = rbwt.data(); d.rev.bwtWords = rbwt.size(); d.rev.occ = rocc.data(); d.rev.occRecs = rocc.size() / 4; d.rev.inverseSa0 = rInverseSa0;
    for (int k = 0; k < 5; k++) { d.fwd.cumFreq[k] = cum[k]; d.rev.cumFreq[k] = rcum[k]; }
    d.textLength = n;
    d.sa = sa.data();
    d.packedDNA = pac.data(); d.packedWords = pac.size();
    d.lkt = lkt.data(); d.revLkt = rlkt.data(); d.lktSize = 13;
    return d;
}

// ------------------------------------------------------------------------------------------------
// reads: FASTA / FASTQ, plain or gzip (behaviour of loadSingleReadsGz, QueryParser.cpp:996-1355)
// ------------------------------------------------------------------------------------------------
bool loadSingleReads(const std::string &file, uint32_t maxReadLength, int qualityConstant, int maxLenReadName, ReadSet &rs, std::string &err)
{
    gzFile gz = gzopen(file.c_str(), "r");
    if (!gz) { err = "cannot open " + file; return false; }
    unsigned char charMap[256];
    memset(charMap, 0, sizeof charMap);                                 // INDEXFillCharMap, IndexHandler.cpp:25-48
    const char *acgt = "ACGT";
    for (int i = 0; i < 4; i++) { charMap[(int)acgt[i]] = (unsigned char)i; charMap[(int)acgt[i] - 'A' + 'a'] = (unsigned char)i; }
    charMap['U'] = charMap['T']; charMap['N'] = charMap['G']; charMap['u'] = charMap['t']; charMap['n'] = charMap['g'];
    rs.maxReadLength = maxReadLength;
    uint32_t w = 1;
    while (w < maxReadLength) w *= 2;
    rs.wpq = w / 16 ? w / 16 : 1;
    std::vector<char> buf(1 << 20);
    std::string all;
    int got;
    while ((got = gzread(gz, buf.data(), (unsigned)buf.size())) > 0) all.append(buf.data(), (size_t)got);
    gzclose(gz);
    size_t p = 0;
    const size_t N = all.size();
    std::vector<std::vector<uint8_t>> codes;
    std::vector<std::string> quals;
    while (p < N) {
        while (p < N && all[p] != '>' && all[p] != '@') p++;
        if (p >= N) break;
        const bool fq = all[p] == '@';
        rs.isFastq = fq;
        p++;
        // name: up to the first blank, at most maxLenReadName-1 chars; a trailing "/1" or "/2" is cut
        std::string name;
        bool cropped = false;
        while (p < N && all[p] != '\n') {
            if (all[p] == ' ' || all[p] == '\t') cropped = true;
            if (!cropped && (int)name.size() < maxLenReadName - 1) name.push_back(all[p]);
            p++;
        }
        if (name.size() >= 2 && name[name.size() - 2] == '/') name.resize(name.size() - 2);
        if (p < N) p++;
        std::vector<uint8_t> seq;
        uint32_t nuc = 0;
        while (p < N && all[p] != '>' && all[p] != '@' && all[p] != '+') {
            if (all[p] != '\n') {
                if (nuc < maxReadLength) seq.push_back(charMap[(unsigned char)all[p]]);
                nuc++;
            }
            p++;
        }
        if (nuc == 0) { err = "a read of length 0 in the read file"; return false; }
        std::string q(seq.size(), (char)0);
        if (fq) {
            while (p < N && all[p] != '+') p++;
            while (p < N && all[p] != '\n') p++;
            if (p < N) p++;
            for (uint32_t i = 0; i < nuc && p < N; i++) {
                if (all[p] == '\n') break;
                if (i < q.size()) q[i] = (char)(all[p] - qualityConstant);
                p++;
            }
        }
        rs.names.push_back(name);
        codes.push_back(seq);
        quals.push_back(q);
    }
    rs.n = (uint32_t)codes.size();
    const size_t roundUp = ((size_t)rs.n + 31) / 32 * 32;
    rs.queries.assign(roundUp * rs.wpq, 0);
    rs.lengths.assign(rs.n, 0);
    rs.codes.assign((size_t)rs.n * maxReadLength, 0);
    rs.quals.assign((size_t)rs.n * maxReadLength, 0);
    for (uint32_t r = 0; r < rs.n; r++) {
        const std::vector<uint8_t> &s = codes[r];
        rs.lengths[r] = (uint32_t)s.size();
        uint32_t *dst = rs.queries.data() + (size_t)(r / 32) * 32 * rs.wpq + r % 32;
        for (size_t i = 0; i < s.size(); i++) {
            dst[(i / 16) * 32] |= (uint32_t)s[i] << (2 * (i % 16));
            rs.codes[(size_t)r * maxReadLength + i] = s[i];
            rs.quals[(size_t)r * maxReadLength + i] = quals[r][i];
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// SAM
// ------------------------------------------------------------------------------------------------
SamWriter::SamWriter(const HostIndex &ix_, const SamOptions &o, FILE *f) : ix(ix_), opt(o), out(f)
{
    for (int i = 1; i != 256; ++i) gLogN[i] = (int)(4.343 * log((double)i) + 0.5);      // bwase_initialize, CPUfunctions.cpp:3014
    gLogN[0] = 0;
}

void SamWriter::header()
{
    // SAMOutputHeaderConstruct (SAM.cpp:82-137); the @PG line names this program
    fprintf(out, "@HD\tVN:1.3\tSO:unsorted\n@RG\tID:%s\tSM:%s\t%s\n", opt.readGroup.c_str(), opt.sampleName.c_str(), opt.readGrpOption.c_str());
    for (int i = 0; i < ix.numOfSeq; i++) fprintf(out, "@SQ\tSN:%s\tLN:%u\n", ix.seqName[i].c_str(), ix.actualEnd[i] - ix.actualStart[i]);
    fprintf(out, "@PG\tID:SOAP3-dp\tPN:SOAP3-dp\tVN:%s\n", s3_version());
}

uint32_t SamWriter::chrAndPos(uint32_t ambPos, uint64_t &tp, uint32_t &chr) const
{
    uint32_t v = ix.ambiguityMap[ambPos >> 18];                         // GRID_SAMPLING_FACTOR_2_POWER
    while (ix.translate[v].startPos > ambPos) v--;
    tp = (uint32_t)(ambPos - ix.translate[v].correction);
    chr = ix.translate[v].chrID;
    // BGS-IO.cpp:1772-1773 compares against numOfRemovedSegment - 1 (unsigned); with zero removed segments the reference reads
    // past the table (undefined) -- there the evident intent is used: end of the next entry, or the text length for the last one.
    if (ix.numOfRemovedSegment == 0) return (size_t)v + 1 < ix.translate.size() ? ix.translate[v + 1].startPos - 1 : ix.dnaLength;
    return v < ix.numOfRemovedSegment - 1 ? ix.translate[v + 1].startPos - 1 : ix.dnaLength;
}

int SamWriter::chrAndPosChecked(uint32_t readLen, uint32_t ambPos, uint64_t &tp, uint32_t &chr, std::string *cigar) const
{
    uint32_t segEnd = chrAndPos(ambPos, tp, chr);
    const uint32_t chrEnd = ix.seqEnd[chr - 1];
    segEnd = chrEnd < segEnd ? chrEnd : segEnd;                          // BoundaryCheck, BGS-IO.cpp:1779-1806
    if (ambPos + readLen <= segEnd + 1) return 0;
    const int aligned = (int)(segEnd - ambPos + 1);
    char tmp[64];
    if (aligned >= ((int)readLen + 1) / 2) {
        if (cigar) { snprintf(tmp, sizeof tmp, "%dM%dS", aligned, (int)readLen - aligned); *cigar = tmp; }
        return -((int)readLen - aligned);                               // trim right; position unchanged
    }
    if (cigar) { snprintf(tmp, sizeof tmp, "%dS%dM", aligned, (int)readLen - aligned); *cigar = tmp; }
    chrAndPos(segEnd + 1, tp, chr);                                      // trimmed left: report the corrected position
    return aligned;
}

void SamWriter::seqAndQual(const ReadSet &rs, uint32_t r, bool negative, std::string &line) const
{
    const uint32_t L = rs.lengths[r];
    const uint8_t *c = &rs.codes[(size_t)r * rs.maxReadLength];
    const char *q = &rs.quals[(size_t)r * rs.maxReadLength];
    static const char dna[] = "ACGT";
    if (!negative) for (uint32_t i = 0; i < L; i++) line.push_back(dna[c[i]]);
    else for (uint32_t i = 0; i < L; i++) line.push_back(dna[3 - c[L - 1 - i]]);
    line.push_back('\t');
    if (L && (unsigned char)q[negative ? L - 1 : 0] == 0xff) line.push_back('*');
    else if (!negative) for (uint32_t i = 0; i < L; i++) line.push_back((char)(q[i] + 33));
    else for (uint32_t i = 0; i < L; i++) line.push_back((char)(q[L - 1 - i] + 33));
}

void SamWriter::unaligned(const ReadSet &rs, uint32_t r)
{
    std::string line = rs.names[r] + "\t4\t*\t0\t0\t*\t*\t0\t0\t";
    seqAndQual(rs, r, false, line);
    line += "\tRG:Z:" + opt.readGroup + "\n";
    fwrite(line.data(), 1, line.size(), out);
    nRecords++;
}

// MD string and average quality at the mismatching positions (getMdStr / mdStr, PE.cpp:209-284,374-420)
static int mdString(const HostIndex &ix, const uint8_t *query, const char *qual, uint32_t qlen, uint32_t pos, bool negative, int mismatchNum, int trim,
                    std::string &md, int &avgQual)
{
    if (!negative) {
        if (trim > 0) { query += trim; pos += (uint32_t)trim; qlen -= (uint32_t)trim; }
        else if (trim < 0) qlen -= (uint32_t)(-trim);
    } else {
        if (trim > 0) { pos += (uint32_t)trim; qlen -= (uint32_t)trim; }
        else if (trim < 0) { query += -trim; qlen -= (uint32_t)(-trim); }
    }
    avgQual = 40;                                                         // DEFAULT_QUAL_VALUE
    md.clear();
    if (mismatchNum == 0) { md = std::to_string(qlen); return (int)md.size(); }
    double sum = 0;
    int prev = -1;
    static const char dna[] = "ACGT";
    for (uint32_t p = 0; p < qlen; p++) {
        const uint32_t qc = negative ? 3u - query[qlen - 1 - p] : query[p];
        const uint64_t tp = (uint64_t)pos + p;
        const uint32_t tc = tp < ix.n ? (ix.pac[tp >> 4] >> (2 * (15 - (tp & 15)))) & 3u : 0u;
        if (qc != tc) {
            md += std::to_string((int)p - prev - 1);
            md.push_back(dna[tc]);
            prev = (int)p;
            sum += qual[p];
        }
    }
    md += std::to_string((int)qlen - prev - 1);
    avgQual = (int)(sum / mismatchNum);
    return (int)md.size();
}

// mismatches (0..5) x average mismatch quality class: fraction of MaxMAPQ (the reference's mapping_score table, BGS-IO.cpp:33)
static const double kMappingScore[6][2] = {{1.0, 1.0}, {0.875, 0.85}, {0.75, 0.7}, {0.625, 0.55}, {0.475, 0.4}, {0.325, 0.25}};

void SamWriter::singleEnd(const ReadSet &rs, uint32_t r, const uint32_t *pos, const uint8_t *strand, const uint8_t *mis, size_t nOcc, uint32_t checkLen)
{
    if (nOcc == 0) { unaligned(rs, r); return; }
    const uint32_t fullLen = rs.lengths[r];
    const uint32_t readLen = checkLen ? checkLen : fullLen;             // boundary check + XA CIGAR length
    // best hit = first hit with the fewest mismatches (BGS-IO.cpp:5590-5610)
    size_t best = 0;
    uint32_t bestMis = mis[0];
    int bestHitNum = 1;
    for (size_t k = 1; k < nOcc; k++) {
        if (mis[k] < bestMis) { best = k; bestMis = mis[k]; bestHitNum = 1; }
        else if (mis[k] == bestMis) bestHitNum++;
    }
    uint64_t bestTP = 0;
    uint32_t bestChr = 0;
    std::string newCigar;
    int boundTrim = chrAndPosChecked(readLen, pos[best], bestTP, bestChr, nullptr);
    if (boundTrim) {                                                      // cross-chromosome handling, :5614-5650
        const size_t stored = best;
        long found = -1;
        bestMis = 9999; bestHitNum = 0;
        for (size_t k = 0; k < nOcc; k++) {
            uint64_t tp; uint32_t ch;
            if (!chrAndPosChecked(readLen, pos[k], tp, ch, nullptr)) {
                if (mis[k] < bestMis) { found = (long)k; bestMis = mis[k]; bestHitNum = 1; }
                else if (mis[k] == bestMis) bestHitNum++;
            }
        }
        if (found < 0) { found = (long)stored; bestHitNum = 1; }
        best = (size_t)found;
        boundTrim = chrAndPosChecked(readLen, pos[best], bestTP, bestChr, &newCigar);
    }
    const bool negative = strand[best] == 2;
    int secBestHitNum = 0;
    std::string xa;
    if (nOcc > 1 && !boundTrim) {                                         // XA:Z, :5657-5695
        for (size_t k = 0; k < nOcc; k++) {
            if (k == best) continue;
            if (opt.alignmentType == 2 && mis[k] > bestMis) continue;   // OUTPUT_ALL_BEST
            uint64_t tp; uint32_t ch;
            if (chrAndPosChecked(readLen, pos[k], tp, ch, nullptr)) continue;
            xa += ix.seqName[ch - 1];
            xa.push_back(',');
            xa.push_back(strand[k] == 2 ? '-' : '+');
            xa += std::to_string((unsigned long long)tp);
            xa.push_back(',');
            xa += std::to_string(readLen);
            xa += "M,";
            xa += std::to_string((unsigned)mis[k]);
            xa.push_back(';');
            if (mis[k] > bestMis) secBestHitNum++;
        }
    }
    std::string md;
    int avgQual = 40;
    const uint8_t *codes = &rs.codes[(size_t)r * rs.maxReadLength];
    const char *quals = &rs.quals[(size_t)r * rs.maxReadLength];
    // bestMis == 9999 cannot reach here: a hit is always chosen
    const int mdLen = mdString(ix, codes, quals, fullLen, pos[best], negative, (int)mis[best], boundTrim, md, avgQual);
    uint32_t outMis = mis[best];
    if (boundTrim && outMis) {                                            // :5706-5712
        outMis = 0;
        for (char c : md) outMis += c > '9';
    }
    // MAPQ: getMapQualScoreSingle (BGS-IO.cpp:2329-2368)
    int mapq;
    if (opt.bwaLikeScore) {
        if (bestHitNum > 1) mapq = 0;
        else if (secBestHitNum == 0) mapq = 37;
        else { const int x1 = secBestHitNum > 255 ? 255 : secBestHitNum; const int nl = gLogN[x1]; mapq = 23 < nl ? 0 : 23 - nl; }
    } else {
        if (bestHitNum == 1 && secBestHitNum == 0) {
            int mi = (int)outMis > 5 ? 5 : (int)outMis;
            int qi = ((rs.isFastq ? avgQual : 20) - 1) / 20;
            qi = qi > 1 ? 1 : qi < 0 ? 0 : qi;
            mapq = (int)(opt.maxMAPQ * kMappingScore[mi][qi]);
            if (mapq < opt.minMAPQ) mapq = opt.minMAPQ;
        } else mapq = opt.minMAPQ;
    }
    (void)mdLen;
    std::string line = rs.names[r];
    line.push_back('\t');
    line += std::to_string(negative ? 16 : 0);
    line.push_back('\t');
    line += ix.seqName[bestChr - 1];
    line.push_back('\t');
    line += std::to_string((unsigned long long)bestTP);
    line.push_back('\t');
    line += std::to_string(mapq);
    line.push_back('\t');
    if (!newCigar.empty()) line += newCigar; else { line += std::to_string(fullLen); line.push_back('M'); }
    line += "\t*\t0\t0\t";
    seqAndQual(rs, r, negative, line);
    line += "\tRG:Z:" + opt.readGroup;
    if (opt.printMDNM) line += "\tNM:i:" + std::to_string(outMis);
    line += "\tX0:i:" + std::to_string(bestHitNum) + "\tX1:i:" + std::to_string(secBestHitNum) + "\tXM:i:" + std::to_string(outMis) + "\tXO:i:0\tXG:i:0";
    if (opt.printMDNM && !md.empty()) line += "\tMD:Z:" + md;
    if (nOcc > 1 && !xa.empty()) line += "\tXA:Z:" + xa;
    line.push_back('\n');
    fwrite(line.data(), 1, line.size(), out);
    nRecords++;
}

void SamWriter::singleAnswer(const ReadSet &rs, uint32_t r, uint32_t pos, uint8_t strand, uint8_t mis)
{
    const uint32_t readLen = rs.lengths[r];
    const bool negative = strand == 2;
    uint64_t tp = 0;
    uint32_t chr = 0;
    chrAndPos(pos, tp, chr);
    std::string md;
    int avgQual = 40;
    mdString(ix, &rs.codes[(size_t)r * rs.maxReadLength], &rs.quals[(size_t)r * rs.maxReadLength], readLen, pos, negative, (int)mis, 0, md, avgQual);
    // getMapQualScore with n = 1 (BGS-IO.cpp:2280): the mismatch x quality-class table, second-best hits not considered
    int mi = mis > 5 ? 5 : (int)mis;
    int qi = ((rs.isFastq ? avgQual : 20) - 1) / 20;
    qi = qi > 1 ? 1 : qi < 0 ? 0 : qi;
    int mapq = (int)(opt.maxMAPQ * kMappingScore[mi][qi]);
    if (mapq < opt.minMAPQ) mapq = opt.minMAPQ;
    std::string line = rs.names[r];
    line.push_back('\t');
    line += std::to_string(negative ? 16 : 0);
    line.push_back('\t');
    line += ix.seqName[chr - 1];
    line.push_back('\t');
    line += std::to_string((unsigned long long)tp);
    line.push_back('\t');
    line += std::to_string(mapq);
    line.push_back('\t');
    line += std::to_string(readLen);
    line += "M\t*\t0\t0\t";
    seqAndQual(rs, r, negative, line);
    line += "\tRG:Z:" + opt.readGroup;
    if (opt.printMDNM) line += "\tNM:i:" + std::to_string((unsigned)mis);
    line += "\tX0:i:1\tXM:i:" + std::to_string((unsigned)mis) + "\tXO:i:0\tXG:i:0";
    if (opt.printMDNM && !md.empty()) line += "\tMD:Z:" + md;
    line.push_back('\n');
    fwrite(line.data(), 1, line.size(), out);
    nRecords++;
}

}  // namespace s3host
