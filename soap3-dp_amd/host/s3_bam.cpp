// s3_bam.cpp -- SAM text -> BAM records -> BGZF blocks (see s3_bam.h)
#include "s3_bam.h"
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <map>
#include <mutex>
#include <unordered_map>

namespace s3host {

namespace {
void put32(std::string &s, uint32_t v) { char b[4] = {(char)v, (char)(v >> 8), (char)(v >> 16), (char)(v >> 24)}; s.append(b, 4); }
void put16(std::string &s, uint32_t v) { char b[2] = {(char)v, (char)(v >> 8)}; s.append(b, 2); }

// bam_reg2bin (bam.h:654)
inline int reg2bin(uint32_t beg, uint32_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return 4681 + (beg >> 14);
    if (beg >> 17 == end >> 17) return 585 + (beg >> 17);
    if (beg >> 20 == end >> 20) return 73 + (beg >> 20);
    if (beg >> 23 == end >> 23) return 9 + (beg >> 23);
    if (beg >> 26 == end >> 26) return 1 + (beg >> 26);
    return 0;
}

struct Field { const char *p; size_t n; };

inline bool nextField(const char *&cur, const char *end, Field &f)
{
    if (cur > end) return false;
    const char *q = (const char *)memchr(cur, '\t', (size_t)(end - cur));
    if (!q) q = end;
    f.p = cur; f.n = (size_t)(q - cur);
    cur = q + 1;
    return true;
}

inline long long toInt(const Field &f)
{
    char tmp[32];
    const size_t n = f.n < 31 ? f.n : 31;
    memcpy(tmp, f.p, n); tmp[n] = 0;
    return strtoll(tmp, nullptr, 10);
}

// "=ACMGRSVTWYHKDBN" (bam_nt16_table)
inline uint8_t nt16(char c)
{
    switch (c) {
    case '=': return 0; case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'M': case 'm': return 3; case 'G': case 'g': return 4;
    case 'R': case 'r': return 5; case 'S': case 's': return 6; case 'V': case 'v': return 7; case 'T': case 't': return 8; case 'W': case 'w': return 9;
    case 'Y': case 'y': return 10; case 'H': case 'h': return 11; case 'K': case 'k': return 12; case 'D': case 'd': return 13; case 'B': case 'b': return 14;
    default: return 15;
    }
}
}  // namespace

bool samTextToBam(const std::vector<std::string> &refNames, const char *text, size_t n, std::string &raw, std::string &err)
{
    // reference name -> index: the last answer first (consecutive records mostly name the same sequence), then the map -- looked up with a key kept
    // between records, so a lookup allocates nothing
    std::unordered_map<std::string, int> ref;
    for (size_t i = 0; i < refNames.size(); i++) ref.emplace(refNames[i], (int)i);
    int lastId = -1;
    std::string key;
    const char *cur = text, *const endAll = text + n;
    std::vector<uint32_t> cigar;
    while (cur < endAll) {
        const char *eol = (const char *)memchr(cur, '\n', (size_t)(endAll - cur));
        if (!eol) eol = endAll;
        if (eol == cur) { cur = eol + 1; continue; }
        if (*cur == '@') { cur = eol + 1; continue; }                     // header lines belong to the header block
        const char *p = cur;
        Field qname, flagF, rname, posF, mapqF, cigarF, rnext, pnextF, tlenF, seq, qual;
        if (!nextField(p, eol, qname) || !nextField(p, eol, flagF) || !nextField(p, eol, rname) || !nextField(p, eol, posF) || !nextField(p, eol, mapqF) ||
            !nextField(p, eol, cigarF) || !nextField(p, eol, rnext) || !nextField(p, eol, pnextF) || !nextField(p, eol, tlenF) || !nextField(p, eol, seq) ||
            !nextField(p, eol, qual)) {
            err = "not a SAM line: " + std::string(cur, (size_t)(eol - cur > 80 ? 80 : eol - cur));
            return false;
        }
        auto refId = [&](const Field &f) -> int {
            if (f.n == 1 && f.p[0] == '*') return -1;
            if (lastId >= 0 && refNames[(size_t)lastId].size() == f.n && memcmp(refNames[(size_t)lastId].data(), f.p, f.n) == 0) return lastId;
            key.assign(f.p, f.n);
            const auto it = ref.find(key);
            if (it == ref.end()) return -1;
            lastId = it->second;
            return lastId;
        };
        const int tid = refId(rname);
        const int mtid = (rnext.n == 1 && rnext.p[0] == '=') ? tid : refId(rnext);
        const int32_t pos = (int32_t)toInt(posF) - 1, mpos = (int32_t)toInt(pnextF) - 1;
        const uint32_t flag = (uint32_t)toInt(flagF), mapq = (uint32_t)toInt(mapqF);
        const int32_t tlen = (int32_t)toInt(tlenF);
        // CIGAR
        cigar.clear();
        uint32_t refLen = 0;
        if (!(cigarF.n == 1 && cigarF.p[0] == '*')) {
            uint32_t len = 0;
            for (size_t k = 0; k < cigarF.n; k++) {
                const char c = cigarF.p[k];
                if (c >= '0' && c <= '9') { len = len * 10 + (uint32_t)(c - '0'); continue; }
                static const char ops[] = "MIDNSHP=X";
                const char *o = strchr(ops, c);
                if (!o) { err = std::string("unknown CIGAR operation '") + c + "'"; return false; }
                const uint32_t op = (uint32_t)(o - ops);
                cigar.push_back(len << 4 | op);
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) refLen += len;
                len = 0;
            }
        }
        const bool noSeq = seq.n == 1 && seq.p[0] == '*';
        const uint32_t lseq = noSeq ? 0u : (uint32_t)seq.n;
        const uint32_t endPos = pos < 0 ? 0u : (uint32_t)pos + (refLen ? refLen : 1u);      // bam_calend; an unmapped read covers one base (sam_read1)
        const uint32_t bin = (uint32_t)reg2bin(pos < 0 ? (uint32_t)-1 : (uint32_t)pos, endPos);
        const size_t at = raw.size();
        put32(raw, 0);                                                      // block_size, patched below
        put32(raw, (uint32_t)tid);
        put32(raw, (uint32_t)pos);
        put32(raw, bin << 16 | (mapq & 0xFFu) << 8 | (uint32_t)((qname.n + 1) & 0xFFu));
        put32(raw, flag << 16 | (uint32_t)cigar.size());
        put32(raw, lseq);
        put32(raw, (uint32_t)mtid);
        put32(raw, (uint32_t)mpos);
        put32(raw, (uint32_t)tlen);
        raw.append(qname.p, qname.n);
        raw.push_back('\0');
        for (uint32_t c : cigar) put32(raw, c);
        {
            // sequence (two bases per byte) and qualities written in place
            const size_t sb = (lseq + 1) / 2, at2 = raw.size();
            raw.resize(at2 + sb + lseq);
            char *o = &raw[at2];
            for (uint32_t k = 0; k + 1 < lseq; k += 2) o[k >> 1] = (char)(nt16(seq.p[k]) << 4 | nt16(seq.p[k + 1]));
            if (lseq & 1u) o[lseq >> 1] = (char)(nt16(seq.p[lseq - 1]) << 4);
            o += sb;
            if (qual.n == 1 && qual.p[0] == '*' && lseq != 1) memset(o, 0xFF, lseq);
            else for (uint32_t k = 0; k < lseq; k++) o[k] = (char)((k < qual.n ? qual.p[k] : '!') - 33);
        }
        // optional fields TAG:TYPE:VALUE (sam_read1's rules for the integer width, bam_import.c)
        Field t;
        while (p <= eol && nextField(p, eol, t)) {
            if (t.n < 5 || t.p[2] != ':' || t.p[4] != ':') { if (t.n == 0) continue; err = "bad optional field: " + std::string(t.p, t.n); return false; }
            raw.append(t.p, 2);
            const char type = t.p[3];
            const Field v{t.p + 5, t.n - 5};
            if (type == 'A' || type == 'a' || type == 'c' || type == 'C') { raw.push_back('A'); raw.push_back(v.n ? v.p[0] : ' '); }
            else if (type == 'i' || type == 'I') {
                const long long x = toInt(v);
                if (x < 0) {
                    if (x >= -128) { raw.push_back('c'); raw.push_back((char)x); }
                    else if (x >= -32768) { raw.push_back('s'); put16(raw, (uint32_t)x); }
                    else { raw.push_back('i'); put32(raw, (uint32_t)x); }
                } else {
                    if (x <= 255) { raw.push_back('C'); raw.push_back((char)x); }
                    else if (x <= 65535) { raw.push_back('S'); put16(raw, (uint32_t)x); }
                    else { raw.push_back('I'); put32(raw, (uint32_t)x); }
                }
            } else if (type == 'f') {
                char tmp[64];
                const size_t m = v.n < 63 ? v.n : 63;
                memcpy(tmp, v.p, m); tmp[m] = 0;
                const float fl = (float)atof(tmp);
                uint32_t bits;
                memcpy(&bits, &fl, 4);
                raw.push_back('f'); put32(raw, bits);
            } else if (type == 'Z' || type == 'H') { raw.push_back(type); raw.append(v.p, v.n); raw.push_back('\0'); }
            else { err = std::string("unknown optional field type '") + type + "'"; return false; }
        }
        const uint32_t blockSize = (uint32_t)(raw.size() - at - 4);
        raw[at] = (char)blockSize; raw[at + 1] = (char)(blockSize >> 8); raw[at + 2] = (char)(blockSize >> 16); raw[at + 3] = (char)(blockSize >> 24);
        cur = eol + 1;
    }
    return true;
}

void bamHeaderBytes(const std::string &headerText, const std::vector<std::string> &refNames, const std::vector<uint32_t> &refLengths, std::string &raw)
{
    raw.append("BAM\1", 4);
    put32(raw, (uint32_t)headerText.size());
    raw += headerText;
    put32(raw, (uint32_t)refNames.size());
    for (size_t i = 0; i < refNames.size(); i++) {
        put32(raw, (uint32_t)refNames[i].size() + 1u);
        raw += refNames[i];
        raw.push_back('\0');
        put32(raw, i < refLengths.size() ? refLengths[i] : 0u);
    }
}

namespace {
// one deflate state per thread and level, reset per block: deflateInit2 allocates a quarter of a megabyte -- per 64 KB block and from every host
// thread at once that is an mmap / munmap pair on one address space for every block
struct Deflater {
    z_stream zs;
    int level = -100;
    bool live = false;
    ~Deflater() { if (live) deflateEnd(&zs); }
    bool ready(int lv)
    {
        if (live && lv == level) return deflateReset(&zs) == Z_OK;
        if (live) { deflateEnd(&zs); live = false; }
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, lv, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
        live = true; level = lv;
        return true;
    }
};
}  // namespace

// one block of at most 0xff00 bytes (len == 0: the empty block that marks the end of a BAM file -- never written in the middle: samtools stops there)
static bool bgzfBlock(const char *data, size_t len, std::string &out, int level)
{
    thread_local Deflater df;
    thread_local std::vector<unsigned char> bufV(65536 + 1024);
    unsigned char *buf = bufV.data();
    if (!df.ready(level)) return false;
    z_stream &zs = df.zs;
    zs.next_in = (Bytef *)data; zs.avail_in = (uInt)len;
    zs.next_out = buf + 18; zs.avail_out = (uInt)(bufV.size() - 18 - 8);
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    if (rc != Z_STREAM_END) return false;                               // (incompressible data of 0xff00 bytes still fits: deflate's stored blocks add 5 bytes per 64 KB)
    const size_t total = clen + 18 + 8;
    static const unsigned char hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(buf, hdr, 16);
    buf[16] = (unsigned char)((total - 1) & 0xFF); buf[17] = (unsigned char)((total - 1) >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)data, (uInt)len);
    unsigned char *tail = buf + 18 + clen;
    tail[0] = (unsigned char)crc; tail[1] = (unsigned char)(crc >> 8); tail[2] = (unsigned char)(crc >> 16); tail[3] = (unsigned char)(crc >> 24);
    tail[4] = (unsigned char)len; tail[5] = (unsigned char)(len >> 8); tail[6] = (unsigned char)(len >> 16); tail[7] = (unsigned char)(len >> 24);
    out.append((const char *)buf, total);
    return true;
}

bool bgzfCompress(const char *data, size_t n, std::string &out, int level)
{
    const size_t kBlock = 0xff00;                                           // bgzf.c: DEFAULT_BLOCK_SIZE 64 KB, filled to 0xff00
    for (size_t off = 0; off < n; off += kBlock)
        if (!bgzfBlock(data + off, n - off < kBlock ? n - off : kBlock, out, level)) return false;
    return true;                                                            // (nothing in, nothing out)
}

const std::string &bgzfEofBlock()
{
    static const std::string eof = [] { std::string s; bgzfBlock("", 0, s, -1); return s; }();
    return eof;
}

// ---- files ----------------------------------------------------------------------------------------------------------------------------
namespace {
std::mutex gMu;
std::map<FILE *, BamFile> gFiles;
}

BamFile *bamFileOf(FILE *f)
{
    std::lock_guard<std::mutex> g(gMu);
    const auto it = gFiles.find(f);
    return it == gFiles.end() ? nullptr : &it->second;
}

void bamRegister(FILE *f, const std::vector<std::string> &refNames, const std::vector<uint32_t> &refLengths, int level)
{
    std::lock_guard<std::mutex> g(gMu);
    BamFile &b = gFiles[f];
    b.f = f; b.refNames = refNames; b.refLengths = refLengths; b.level = level;
}

bool bamWriteHeader(FILE *f, const std::string &headerText)
{
    BamFile *b = bamFileOf(f);
    if (!b) return false;
    std::string raw, z;
    bamHeaderBytes(headerText, b->refNames, b->refLengths, raw);
    if (!bgzfCompress(raw.data(), raw.size(), z, b->level)) return false;
    return fwrite(z.data(), 1, z.size(), f) == z.size();
}

bool bamFlush(FILE *f)
{
    BamFile *b = bamFileOf(f);
    if (!b) return true;
    if (b->pending.empty()) return true;
    std::string z;
    if (!bgzfCompress(b->pending.data(), b->pending.size(), z, b->level)) return false;
    b->pending.clear();
    return fwrite(z.data(), 1, z.size(), f) == z.size();
}

bool bamWriteLines(FILE *f, const char *text, size_t n)
{
    BamFile *b = bamFileOf(f);
    if (!b) return false;
    std::string err;
    if (!samTextToBam(b->refNames, text, n, b->pending, err)) { fprintf(stderr, "BAM output: %s\n", err.c_str()); return false; }
    if (b->pending.size() >= (size_t)1 << 20) return bamFlush(f);
    return true;
}

bool bamClose(FILE *f)
{
    if (!bamFileOf(f)) return true;
    if (!bamFlush(f)) return false;
    const std::string &e = bgzfEofBlock();
    const bool ok = fwrite(e.data(), 1, e.size(), f) == e.size();
    std::lock_guard<std::mutex> g(gMu);
    gFiles.erase(f);
    return ok;
}

}  // namespace s3host
