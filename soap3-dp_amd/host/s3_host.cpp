// s3_host.cpp -- index files, read files and SAM records for the drop-in `soap3-dp` program (see s3_host.h).
#include "s3_host.h"
#include "s3_bam.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <fcntl.h>
#include <unistd.h>
#ifdef __SSSE3__
#include <tmmintrin.h>
#endif
#include <sys/stat.h>
#include <algorithm>
#include <atomic>
#include <sched.h>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>

namespace s3host {

// ------------------------------------------------------------------------------------------------
// index files (formats: SURVEY.md 5.9)
// ------------------------------------------------------------------------------------------------
// One index file into memory: the header words, then the payload read at its final place by all host threads (pread, 64 MiB and more per thread).  The
// suffix array of a 3.1 Gbp text is 12.4 GB: one thread's zero fill + fread took 3 of the 4 seconds of "Loading time".
template <class V>
static bool slurpWords(const std::string &name, size_t skipWords, V &out, std::vector<uint32_t> *header, std::string &err)
{
    const int fd = open(name.c_str(), O_RDONLY);
    if (fd < 0) { err = "cannot open " + name; return false; }
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); err = "cannot stat " + name; return false; }
    const size_t bytes = (size_t)st.st_size;
    if (bytes < skipWords * 4) { close(fd); err = "file too short: " + name; return false; }
    auto readAt = [&](void *dst, size_t n, size_t at) -> bool {
        size_t done = 0;
        while (done < n) {
            const ssize_t k = pread(fd, (char *)dst + done, n - done, (off_t)(at + done));
            if (k <= 0) return false;
            done += (size_t)k;
        }
        return true;
    };
    if (header) {
        header->resize(skipWords);
        if (skipWords && !readAt(header->data(), skipWords * 4, 0)) { close(fd); err = "short read " + name; return false; }
    }
    const size_t words = (bytes - skipWords * 4) / 4;
    out.resize(words);
    size_t piece = (size_t)64 << 20;
    if (const char *e = getenv("S3DP_LOAD_PIECE")) { const size_t v = (size_t)strtoull(e, nullptr, 10); if (v >= 4096) piece = v; }   // (tests: small files in many pieces)
    const size_t payload = words * 4;
    size_t T = (payload + piece - 1) / piece;
    if (T > (size_t)hostThreads()) T = (size_t)hostThreads();
    if (T < 1) T = 1;
    std::atomic<bool> bad(false);
    std::vector<std::thread> th;
    const size_t per = ((payload + T - 1) / T + 4095) / 4096 * 4096;
    for (size_t t = 0; t < T; t++) th.emplace_back([&, t] {
        const size_t lo = t * per, hi = lo + per < payload ? lo + per : payload;
        if (lo < hi && !readAt((char *)out.data() + lo, hi - lo, skipWords * 4 + lo)) bad = true;
    });
    for (auto &x : th) x.join();
    close(fd);
    if (bad) { err = "short read " + name; return false; }
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
    // SaValueFreq (README.md:55-65: 1 = full, 2 / 4 = sampled): the library expands a sampled array on the device at upload
    ix.saInterval = hs[5] ? hs[5] : 1;
    const size_t saEntries = ((size_t)ix.n + ix.saInterval) / ix.saInterval;       // BWT.c:255
    if (ix.sa.size() < saEntries) { err = ".sa too short"; return false; }
    ix.sa.resize(saEntries);
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
        {
            // (bytes MSB first into words: pieces of whole words, one per host thread)
            const size_t nb = (size_t)bytes - 1;
            const size_t T = (size_t)hostThreads(), per = ((nb + T - 1) / T + 3) / 4 * 4;
            std::vector<std::thread> th;
            for (size_t t = 0; t < T; t++) th.emplace_back([&, t] {
                const size_t lo = t * per, hi = lo + per < nb ? lo + per : nb;
                for (size_t b = lo; b < hi; b++) ix.pac[b >> 2] |= (uint32_t)raw[b] << (8 * (3 - (b & 3)));
            });
            for (auto &x : th) x.join();
        }
    }
    // lookup tables: load if present (256 MiB each), otherwise rebuild from the text
    {
        std::vector<uint32_t> hdr;
        std::string e2;
        if (slurpWords(prefix + ".lkt", 1, ix.lkt, &hdr, e2) && hdr[0] == 13 && ix.lkt.size() == ((size_t)1 << 26)) {
            if (!slurpWords(prefix + ".rev.lkt", 1, ix.rlkt, &hdr, err)) return false;
            if (hdr[0] != 13 || ix.rlkt.size() != ((size_t)1 << 26)) { err = prefix + ".rev.lkt is not a 13-mer table of 2^26 entries"; return false; }
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
    d.sa = sa.data(); d.saInterval = saInterval;
    d.packedDNA = pac.data(); d.packedWords = pac.size();
    d.lkt = lkt.data(); d.revLkt = rlkt.data(); d.lktSize = 13;
    return d;
}

// ------------------------------------------------------------------------------------------------
// reads: FASTA / FASTQ, plain or gzip (behaviour of loadSingleReadsGz, QueryParser.cpp:996-1355)
// ------------------------------------------------------------------------------------------------
// The whole file is inflated into memory, cut into one piece per host thread at record boundaries, and the pieces are parsed in
// parallel (count pass, then fill pass straight into the packed arrays).  S3DP_HOST_THREADS overrides the thread count.
// CPUs the process may actually use: the hardware's, cut to the control group's quota (a container that shows 256 processors and grants sixteen runs
// 256 threads at a sixteenth of the speed each -- and stalls all of them whenever the quota of a period is spent)
static int cpuAllowance()
{
    int hw = (int)std::thread::hardware_concurrency();
    if (hw < 1) hw = 1;
    auto fromQuota = [&](double quota, double period) { if (quota > 0 && period > 0) { const int c = (int)((quota + period - 1) / period); if (c >= 1 && c < hw) hw = c; } };
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                      // cgroup v2: "<quota|max> <period>"
        char q[64] = {0};
        double period = 0;
        if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0) fromQuota(atof(q), period);
        fclose(f);
    } else {
        double quota = -1, period = 0;                                          // cgroup v1
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &quota) != 1) quota = -1; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &period) != 1) period = 0; fclose(g); }
        fromQuota(quota, period);
    }
    return hw;
}

int hostThreads()
{
    static const int quota = cpuAllowance();
    // ... and to the CPUs the thread may run on NOW: the command line binds itself to the device's NUMA node after its first calls here, and a run under
    // taskset starts narrowed -- threads beyond the affinity mask only take turns
    int allowed = quota;
    cpu_set_t have;
    if (sched_getaffinity(0, sizeof have, &have) == 0) { const int c = CPU_COUNT(&have); if (c >= 1 && c < allowed) allowed = c; }
    const char *e = getenv("S3DP_HOST_THREADS");
    int t = e && *e ? atoi(e) : allowed;
    if (t < 1) t = 1;
    if (t > 256) t = 256;
    return t;
}

namespace {
struct Piece { size_t beg = 0, end = 0; uint32_t first = 0, count = 0; };

// one record starting at p (which points at '>' or '@'); returns the position after it.  Emits through `rec` when asked to.
// the whole read file in memory: a malloc'ed buffer when the file is plain (no value-initialisation: the slices are filled by the reading threads, which
// also take the page faults in parallel), a string when it had to be inflated
struct TextView {
    const char *p = nullptr;
    size_t n = 0;
    char operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
};

template <class F>
size_t parseRecord(const TextView &all, size_t p, bool fq, uint32_t maxReadLength, int maxLenReadName, const unsigned char *charMap, int qualityConstant, F &&rec,
                   bool fill, std::string &name, std::vector<uint8_t> &seq, std::string &q, bool &bad)
{
    const size_t N = all.size();
    p++;
    if (fill) name.clear();
    bool cropped = false;
    while (p < N && all[p] != '\n') {
        if (all[p] == ' ' || all[p] == '\t') cropped = true;
        if (fill && !cropped && (int)name.size() < maxLenReadName - 1) name.push_back(all[p]);
        p++;
    }
    if (fill && name.size() >= 2 && name[name.size() - 2] == '/') name.resize(name.size() - 2);
    if (p < N) p++;
    if (fill) seq.clear();
    uint32_t nuc = 0;
    while (p < N && all[p] != '>' && all[p] != '@' && all[p] != '+') {
        if (all[p] != '\n') {
            if (fill && nuc < maxReadLength) seq.push_back(charMap[(unsigned char)all[p]]);
            nuc++;
        }
        p++;
    }
    if (nuc == 0) bad = true;
    if (fill) q.assign(seq.size(), (char)0);
    if (fq) {
        while (p < N && all[p] != '+') p++;
        while (p < N && all[p] != '\n') p++;
        if (p < N) p++;
        for (uint32_t i = 0; i < nuc && p < N; i++) {
            if (all[p] == '\n') break;
            if (fill && i < q.size()) q[i] = (char)(all[p] - qualityConstant);
            p++;
        }
    }
    rec();
    return p;
}
}  // namespace

// sizes the arrays of `rs` for n reads.  zeroContent (the BAM path, whose records OR their bits into the packed words): the zeros that are content are
// written here, slice by slice with all threads; the text path writes every word and every byte of a read's rows itself (fillRead), which saved three
// passes over 230 MB per chunk of 1 Mi reads.
static void allocateReads(ReadSet &rs, uint32_t n, bool fq, uint32_t wpq, uint32_t maxReadLength, bool zeroContent)
{
    rs.maxReadLength = maxReadLength; rs.wpq = wpq; rs.isFastq = fq; rs.n = n;
    const size_t roundUp = ((size_t)n + 31) / 32 * 32;
    rs.queries.resize(roundUp * wpq);
    rs.lengths.resize(n);
    rs.names.resize(n);
    rs.codes.resize((size_t)n * maxReadLength);
    rs.quals.resize((size_t)n * maxReadLength);
    if (!zeroContent) {
        // the packed words of the reads that pad the last group of 32
        for (size_t r = n; r < roundUp; r++) for (uint32_t w = 0; w < wpq; w++) rs.queries[(r / 32) * 32 * wpq + r % 32 + (size_t)32 * w] = 0;
        return;
    }
    const int T = hostThreads();
    auto zero = [&](void *base, size_t bytes) {
        std::vector<std::thread> z;
        for (int t = 0; t < T; t++) z.emplace_back([=] {
            const size_t lo = bytes / (size_t)T * (size_t)t, hi = t + 1 == T ? bytes : bytes / (size_t)T * (size_t)(t + 1);
            memset((char *)base + lo, 0, hi - lo);
        });
        for (auto &x : z) x.join();
    };
    zero(rs.queries.data(), rs.queries.size() * 4);
    zero(rs.codes.data(), rs.codes.size());
    zero(rs.quals.data(), rs.quals.size());
}

// ---- the read files as a stream of chunks ------------------------------------------------------------------------------------------------------
// The reference reads its input batch by batch (a loader thread fills the next buffer while the current one is aligned, aio_thread.cpp:582,
// SOAP3-DP.cu:607-800).  Here a chunk is cut at a record boundary out of a text buffer that is refilled from the file (plain files: 16 threads pread
// their slices; gzip: zlib), parsed by all host threads straight into the packed layout, and handed on; only the unparsed tail stays in the buffer.
struct ReadChunker::File {
    std::string name;
    int fd = -1;
    gzFile gz = nullptr;
    uint64_t pos = 0, size = 0;           // plain files
    bool eof = false, fq = false, kindKnown = false;
    RawVec<char> buf;
    size_t used = 0, have = 0;            // buf[used, have) is text not yet handed out
    double bytesPerRead = 0;

    ~File() { if (fd >= 0) close(fd); if (gz) gzclose(gz); }

    bool open(const std::string &file, std::string &err)
    {
        name = file;
        gzFile g = gzopen(file.c_str(), "r");
        if (!g) { err = "cannot open " + file; return false; }
        if (gzdirect(g)) {
            gzclose(g);
            fd = ::open(file.c_str(), O_RDONLY);
            if (fd < 0) { err = "cannot open " + file; return false; }
            struct stat sb;
            if (fstat(fd, &sb) != 0) { err = "cannot stat " + file; return false; }
            size = (uint64_t)sb.st_size;
        } else { gz = g; gzbuffer(gz, 1 << 20); }
        return true;
    }

    // make buf[used, have) at least `want` bytes long (or reach the end of the file)
    bool fill(size_t want, std::string &err)
    {
        if (have - used >= want || eof) return true;
        if (used > 0) {                                   // drop what was handed out
            memmove(buf.data(), buf.data() + used, have - used);
            have -= used; used = 0;
        }
        if (buf.size() < want + 1) buf.resize(want + 1);
        const size_t need = want - have;
        if (fd >= 0) {
            const size_t got = (size_t)std::min<uint64_t>(need, size - pos);
            const int Tr = got < (8u << 20) ? 1 : (hostThreads() > 16 ? 16 : hostThreads());
            std::vector<std::thread> rd;
            std::atomic<bool> bad(false);
            char *dst = buf.data() + have;
            for (int t = 0; t < Tr; t++) rd.emplace_back([&, t] {
                const size_t lo = got / (size_t)Tr * (size_t)t, hi = t + 1 == Tr ? got : got / (size_t)Tr * (size_t)(t + 1);
                size_t done = lo;
                while (done < hi) {
                    const ssize_t k = pread(fd, dst + done, hi - done, (off_t)(pos + done));
                    if (k <= 0) { bad = true; return; }
                    done += (size_t)k;
                }
            });
            for (auto &x : rd) x.join();
            if (bad) { err = "cannot read " + name; return false; }
            have += got; pos += got;
            if (pos >= size) eof = true;
        } else {
            size_t got = 0;
            while (got < need) {
                const size_t ask = std::min<size_t>(need - got, 1u << 30);
                const int k = gzread(gz, buf.data() + have + got, (unsigned)ask);
                if (k < 0) { err = "cannot inflate " + name; return false; }
                if (k == 0) { eof = true; break; }
                got += (size_t)k;
            }
            have += got;
        }
        return true;
    }
};

namespace {
// ---- the record form every sequencer writes: header line, ONE sequence line, '+' line, quality line --------------------------------------------------
// parseRecord walks a record byte by byte (0.9 us per 100-base read on one core: the loader set the pace of the command line).  A record of the
// four-line form is taken line by line instead (memchr) and yields exactly what parseRecord yields -- same name, same characters, same position
// behind it, whatever the length of the quality line --; anything else (FASTA, wrapped sequences, a sequence line holding '>', '@' or '+', a record cut
// by the end of the buffer) is left to parseRecord.  S3DP_GENERIC_PARSER=1 switches the short cut off (tests compare the two).
struct FastRec { const char *name, *nameEnd, *seq, *qual; uint32_t nuc, nq; size_t next; };

static bool genericParserOnly()
{
    static const bool v = [] { const char *e = getenv("S3DP_GENERIC_PARSER"); return e && *e == '1'; }();
    return v;
}

static inline bool hasStopChar(const char *s, size_t n)
{
    size_t i = 0;
#ifdef __SSSE3__
    const __m128i gt = _mm_set1_epi8('>'), at = _mm_set1_epi8('@'), pl = _mm_set1_epi8('+');
    for (; i + 16 <= n; i += 16) {
        const __m128i x = _mm_loadu_si128((const __m128i *)(s + i));
        if (_mm_movemask_epi8(_mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(x, gt), _mm_cmpeq_epi8(x, at)), _mm_cmpeq_epi8(x, pl)))) return true;
    }
#endif
    for (; i < n; i++) if (s[i] == '>' || s[i] == '@' || s[i] == '+') return true;
    return false;
}

static inline bool fastqRecord(const TextView &all, size_t p, FastRec &R)
{
    const char *b = all.p, *end = all.p + all.size();
    const char *e1 = (const char *)memchr(b + p + 1, '\n', (size_t)(end - (b + p + 1)));
    if (!e1) return false;
    const char *s = e1 + 1;
    const char *e2 = (const char *)memchr(s, '\n', (size_t)(end - s));
    if (!e2 || e2 == s || e2 + 1 >= end || e2[1] != '+') return false;
    if ((size_t)(e2 - s) > 0xFFFFFFFFu || hasStopChar(s, (size_t)(e2 - s))) return false;
    const char *e3 = (const char *)memchr(e2 + 1, '\n', (size_t)(end - (e2 + 1)));
    if (!e3) return false;
    const char *qs = e3 + 1;
    R.name = b + p + 1; R.nameEnd = e1; R.seq = s; R.qual = qs;
    R.nuc = (uint32_t)(e2 - s);
    const size_t lim = std::min<size_t>(R.nuc, (size_t)(end - qs));
    const char *nl = lim ? (const char *)memchr(qs, '\n', lim) : nullptr;
    R.nq = (uint32_t)(nl ? (size_t)(nl - qs) : lim);
    R.next = (size_t)(qs - b) + R.nq;
    return true;
}

// the name as parseRecord takes it: up to the first blank, at most maxLenReadName - 1 characters, a trailing "/x" dropped
static inline void fastName(const FastRec &R, int maxLenReadName, std::string &name)
{
    const char *e = R.name;
    while (e < R.nameEnd && *e != ' ' && *e != '\t') e++;
    size_t n = (size_t)(e - R.name);
    const size_t cap = maxLenReadName > 1 ? (size_t)(maxLenReadName - 1) : 0;
    if (n > cap) n = cap;
    if (n >= 2 && R.name[n - 2] == '/') n -= 2;
    name.assign(R.name, n);
}

// one read into its rows and packed words; every byte of the rows and every one of the wpq words is written
static inline void fillRead(const char *seq, uint32_t len, const char *qual, uint32_t nq, uint32_t maxReadLength, uint32_t wpq, const unsigned char *charMap,
                            int qualityConstant, uint32_t *dst, uint8_t *cd, char *qd)
{
    uint32_t i0 = 0;
#ifdef __SSSE3__
    {
        // the character table of fillCharMap in registers: case folded by clearing bit 5 (a letter and its lower-case form are the only two bytes that fold
        // onto it), C -> 1, G / N -> 2, T / U -> 3, everything else 0
        const __m128i fold = _mm_set1_epi8((char)0xDF), one = _mm_set1_epi8(1), two = _mm_set1_epi8(2), three = _mm_set1_epi8(3);
        const __m128i cC = _mm_set1_epi8('C'), cG = _mm_set1_epi8('G'), cN = _mm_set1_epi8('N'), cT = _mm_set1_epi8('T'), cU = _mm_set1_epi8('U');
        for (; i0 + 16 <= len; i0 += 16) {
            const __m128i x = _mm_and_si128(_mm_loadu_si128((const __m128i *)(seq + i0)), fold);
            const __m128i c = _mm_and_si128(_mm_cmpeq_epi8(x, cC), one);
            const __m128i g = _mm_and_si128(_mm_or_si128(_mm_cmpeq_epi8(x, cG), _mm_cmpeq_epi8(x, cN)), two);
            const __m128i t = _mm_and_si128(_mm_or_si128(_mm_cmpeq_epi8(x, cT), _mm_cmpeq_epi8(x, cU)), three);
            _mm_storeu_si128((__m128i *)(cd + i0), _mm_or_si128(_mm_or_si128(c, g), t));
        }
    }
#endif
    for (uint32_t i = i0; i < len; i++) cd[i] = charMap[(unsigned char)seq[i]];
    memset(cd + len, 0, maxReadLength - len);
    const uint32_t k = nq < len ? nq : len;
    for (uint32_t i = 0; i < k; i++) qd[i] = (char)(qual[i] - qualityConstant);
    memset(qd + k, 0, maxReadLength - k);
    uint32_t w = 0;
#ifdef __SSSE3__
    {
        // sixteen 2-bit codes -> one word, first character in the lowest bits: c0 + 4 c1 per byte pair, + 16 (..) per pair of those, bytes gathered
        const __m128i m1 = _mm_set1_epi16(0x0401), m2 = _mm_set1_epi32(0x00100001);
        const __m128i gather = _mm_setr_epi8(0, 4, 8, 12, (char)0x80, (char)0x80, (char)0x80, (char)0x80, (char)0x80, (char)0x80, (char)0x80, (char)0x80, (char)0x80, (char)0x80,
                                             (char)0x80, (char)0x80);
        for (; w < wpq && 16u * w + 16u <= maxReadLength; w++) {
            const __m128i x = _mm_loadu_si128((const __m128i *)(cd + 16u * w));
            dst[(size_t)w * 32] = (uint32_t)_mm_cvtsi128_si32(_mm_shuffle_epi8(_mm_madd_epi16(_mm_maddubs_epi16(x, m1), m2), gather));
        }
    }
#endif
    for (; w < wpq; w++) {
        uint32_t word = 0;
        for (uint32_t j = 0, i = 16u * w; j < 16u && i < len; j++, i++) word |= (uint32_t)cd[i] << (2u * j);
        dst[(size_t)w * 32] = word;
    }
}

// what one file contributes to a chunk: the text region, its pieces (one per thread) with their record counts
struct Region {
    TextView all;
    bool fq = false;
    std::vector<Piece> pieces;
    uint32_t n = 0;
    size_t consumed = 0;                  // bytes of the region that hold the records taken
};

bool isRecordStart(const TextView &all, bool fq, size_t p)
{
    const size_t N = all.size();
    if (p >= N || all[p] != (fq ? '@' : '>')) return false;
    if (p > 0 && all[p - 1] != '\n') return false;
    if (!fq) return true;
    // FASTQ: the line after next must begin with '+' (a quality line may begin with '@')
    size_t q = p;
    for (int k = 0; k < 2; k++) { while (q < N && all[q] != '\n') q++; if (q < N) q++; }
    return q < N && all[q] == '+';
}

void fillCharMap(unsigned char *charMap)
{
    memset(charMap, 0, 256);                                            // INDEXFillCharMap, IndexHandler.cpp:25-48
    const char *acgt = "ACGT";
    for (int i = 0; i < 4; i++) { charMap[(int)acgt[i]] = (unsigned char)i; charMap[(int)acgt[i] - 'A' + 'a'] = (unsigned char)i; }
    charMap['U'] = charMap['T']; charMap['N'] = charMap['G']; charMap['u'] = charMap['t']; charMap['n'] = charMap['g'];
}

// pieces + counts of the complete records in `all`, at most maxReads of them (the rest stays for the next chunk)
bool scanRegion(Region &R, uint32_t maxReads, uint32_t maxReadLength, int maxLenReadName, int qualityConstant, const unsigned char *charMap, std::string &err)
{
    const TextView &all = R.all;
    const size_t N = all.size();
    size_t p0 = 0;
    while (p0 < N && !isRecordStart(all, R.fq, p0)) p0++;
    const int T = hostThreads();
    R.pieces.assign((size_t)T, Piece());
    for (int t = 0; t < T; t++) {
        size_t b = p0 + (N - p0) / (size_t)T * (size_t)t;
        if (t == 0) b = p0;
        else { while (b < N && !isRecordStart(all, R.fq, b)) b++; }
        R.pieces[(size_t)t].beg = b;
    }
    for (int t = 0; t < T; t++) R.pieces[(size_t)t].end = t + 1 < T ? R.pieces[(size_t)t + 1].beg : N;
    std::atomic<bool> anyBad(false);
    const bool fast = R.fq && !genericParserOnly();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t] {
        Piece &P = R.pieces[(size_t)t];
        std::string name, q; std::vector<uint8_t> seq; bool bad = false;
        uint32_t c = 0;
        size_t p = P.beg;
        while (p < P.end) {
            while (p < P.end && all[p] != '>' && all[p] != '@') p++;
            if (p >= P.end) break;
            FastRec fr;
            if (fast && fastqRecord(all, p, fr)) { c++; p = fr.next; continue; }
            p = parseRecord(all, p, R.fq, maxReadLength, maxLenReadName, charMap, qualityConstant, [&] { c++; }, false, name, seq, q, bad);
        }
        P.count = c;
        if (bad) anyBad = true;
    });
    for (auto &x : th) x.join();
    if (anyBad) { err = "a read of length 0 in the read file"; return false; }
    uint32_t n = 0;
    for (auto &P : R.pieces) { P.first = n; n += P.count; }
    R.n = n; R.consumed = N;
    (void)maxReads;
    return true;
}

// keep only the first m records of the region (m <= R.n): the piece that holds record m ends there, later pieces are empty
void cutRegion(Region &R, uint32_t m, uint32_t maxReadLength, int maxLenReadName, int qualityConstant, const unsigned char *charMap)
{
    if (m >= R.n) return;
    const TextView &all = R.all;
    bool cutSet = false;
    for (size_t t = 0; t < R.pieces.size(); t++) {
        Piece &P = R.pieces[t];
        if (P.first + P.count <= m) continue;
        if (P.first >= m) {
            if (!cutSet) { R.consumed = P.beg; cutSet = true; }          // the cut falls on a piece boundary
            P.end = P.beg; P.count = 0;
            continue;
        }
        // record m lies in this piece: walk m - first records
        std::string name, q; std::vector<uint8_t> seq; bool bad = false;
        uint32_t c = 0;
        size_t p = P.beg;
        while (p < P.end && c < m - P.first) {
            while (p < P.end && all[p] != '>' && all[p] != '@') p++;
            if (p >= P.end) break;
            FastRec fr;
            if (R.fq && !genericParserOnly() && fastqRecord(all, p, fr)) { c++; p = fr.next; continue; }
            p = parseRecord(all, p, R.fq, maxReadLength, maxLenReadName, charMap, qualityConstant, [&] { c++; }, false, name, seq, q, bad);
        }
        while (p < P.end && !isRecordStart(all, R.fq, p)) p++;
        P.end = p; P.count = c;
        R.consumed = p; cutSet = true;
    }
    R.n = m;
}

// fill pass: record k of the region becomes read stride * k + offset of rs.  32 consecutive reads share packed words (32-interleaved layout), so a
// piece boundary inside a group of 32 is written by two threads -- to different words (word index = group base + r % 32 + 32 w), never the same one
void fillRegion(const Region &R, ReadSet &rs, uint32_t stride, uint32_t offset, uint32_t maxReadLength, int maxLenReadName, int qualityConstant,
                const unsigned char *charMap)
{
    const TextView &all = R.all;
    const uint32_t wpq = rs.wpq;
    const bool fast = R.fq && !genericParserOnly();
    std::vector<std::thread> th;
    for (size_t t = 0; t < R.pieces.size(); t++) th.emplace_back([&, t] {
        const Piece &P = R.pieces[t];
        std::string name, q; std::vector<uint8_t> seq; bool bad = false;
        uint32_t r = P.first;
        size_t p = P.beg;
        while (p < P.end) {
            while (p < P.end && all[p] != '>' && all[p] != '@') p++;
            if (p >= P.end) break;
            FastRec fr;
            if (fast && fastqRecord(all, p, fr)) {
                if (r < R.n) {
                    const size_t rd = (size_t)stride * r + offset;
                    fastName(fr, maxLenReadName, rs.names[rd]);
                    const uint32_t len = fr.nuc < maxReadLength ? fr.nuc : maxReadLength;
                    rs.lengths[rd] = len;
                    fillRead(fr.seq, len, fr.qual, fr.nq, maxReadLength, wpq, charMap, qualityConstant, rs.queries.data() + (rd / 32) * 32 * wpq + rd % 32,
                             &rs.codes[rd * maxReadLength], &rs.quals[rd * maxReadLength]);
                }
                r++;
                p = fr.next;
                continue;
            }
            p = parseRecord(all, p, R.fq, maxReadLength, maxLenReadName, charMap, qualityConstant, [&] {
                if (r < R.n) {
                    const size_t rd = (size_t)stride * r + offset;
                    rs.names[rd] = name;
                    rs.lengths[rd] = (uint32_t)seq.size();
                    uint32_t *dst = rs.queries.data() + (rd / 32) * 32 * wpq + rd % 32;
                    uint8_t *cd = &rs.codes[rd * maxReadLength];
                    char *qd = &rs.quals[rd * maxReadLength];
                    for (uint32_t w = 0; w < wpq; w++) dst[(size_t)w * 32] = 0;
                    memset(cd, 0, maxReadLength);
                    memset(qd, 0, maxReadLength);
                    for (size_t i = 0; i < seq.size(); i++) {
                        dst[(i / 16) * 32] |= (uint32_t)seq[i] << (2 * (i % 16));
                        cd[i] = seq[i];
                        qd[i] = q[i];
                    }
                }
                r++;
            }, true, name, seq, q, bad);
        }
    });
    for (auto &x : th) x.join();
}
}  // namespace

ReadChunker::ReadChunker() {}
ReadChunker::~ReadChunker() {}

bool ReadChunker::open(const std::string &file1, const std::string &file2, uint32_t maxReadLength_, int qualityConstant_, int maxLenReadName_, std::string &err)
{
    maxReadLength = maxReadLength_; qualityConstant = qualityConstant_; maxLenReadName = maxLenReadName_;
    files.clear();
    files.emplace_back(new File());
    if (!files[0]->open(file1, err)) return false;
    if (!file2.empty()) {
        files.emplace_back(new File());
        if (!files[1]->open(file2, err)) return false;
    }
    return true;
}

bool ReadChunker::openBam(const std::string &file, bool paired, uint32_t maxReadLength_, int qualityConstant_, int maxLenReadName_, std::string &err)
{
    maxReadLength = maxReadLength_; qualityConstant = qualityConstant_; maxLenReadName = maxLenReadName_;
    bam = true; bamPaired = paired;
    files.clear();
    files.emplace_back(new File());
    File &X = *files[0];
    X.name = file;
    X.gz = gzopen(file.c_str(), "r");                                   // (BGZF is a series of gzip members: zlib reads across them)
    if (!X.gz) { err = "cannot open " + file; return false; }
    gzbuffer(X.gz, 1 << 20);
    auto rd = [&](void *p, unsigned n) { return gzread(X.gz, p, n) == (int)n; };
    char magic[4];
    int32_t lText = 0, nRef = 0;
    if (!rd(magic, 4) || memcmp(magic, "BAM\1", 4) != 0) { err = file + " is not a BAM file"; return false; }
    if (!rd(&lText, 4) || lText < 0) { err = "bad BAM header in " + file; return false; }
    std::vector<char> skip((size_t)lText + 1);
    if (lText && !rd(skip.data(), (unsigned)lText)) { err = "bad BAM header in " + file; return false; }
    if (!rd(&nRef, 4) || nRef < 0) { err = "bad BAM header in " + file; return false; }
    for (int i = 0; i < nRef; i++) {
        int32_t lName = 0, lRef = 0;
        if (!rd(&lName, 4) || lName < 0) { err = "bad BAM header in " + file; return false; }
        skip.resize((size_t)lName + 1);
        if ((lName && !rd(skip.data(), (unsigned)lName)) || !rd(&lRef, 4)) { err = "bad BAM header in " + file; return false; }
    }
    return true;
}

bool ReadChunker::nextBam(uint32_t maxReads, ReadSet &rs, std::string &err)
{
    File &X = *files[0];
    if (bamPaired) maxReads &= ~1u;
    if (maxReads == 0) { err = "ReadChunker::next: bad chunk size"; return false; }
    unsigned char charMap[256];
    fillCharMap(charMap);
    uint32_t w = 1;
    while (w < maxReadLength) w *= 2;
    const uint32_t wpq = w / 16 ? w / 16 : 1;
    // the records of the chunk, raw (their number is not known before they are read)
    std::vector<std::vector<unsigned char>> recs;
    while (recs.size() < maxReads && !X.eof) {
        int32_t blockSize = 0;
        const int got = gzread(X.gz, &blockSize, 4);
        if (got == 0) { X.eof = true; break; }
        if (got != 4 || blockSize < 32) { err = "truncated BAM record in " + X.name; return false; }
        std::vector<unsigned char> b((size_t)blockSize);
        if (gzread(X.gz, b.data(), (unsigned)blockSize) != blockSize) { err = "truncated BAM record in " + X.name; return false; }
        recs.push_back(std::move(b));
    }
    uint32_t n = (uint32_t)recs.size();
    if (bamPaired) n &= ~1u;                                            // (a last read without its mate is dropped, as a shorter second file would drop it)
    if (n == 0) return false;
    allocateReads(rs, n, true, wpq, maxReadLength, true);
    static const char nt16[] = "=ACMGRSVTWYHKDBN";
    const int T = hostThreads();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t] {
        for (uint32_t r = (uint32_t)t; r < n; r += (uint32_t)T) {
            const unsigned char *b = recs[r].data();
            const size_t size = recs[r].size();
            uint32_t v[8];
            memcpy(v, b, 32);
            const uint32_t lName = v[2] & 0xFFu, flag = v[3] >> 16, nCigar = v[3] & 0xFFFFu, lSeq = v[4];
            const size_t seqAt = 32 + (size_t)lName + 4 * (size_t)nCigar, qualAt = seqAt + (lSeq + 1) / 2;
            if (qualAt + lSeq > size) continue;                        // a broken record: an empty read
            const char *nm = (const char *)b + 32;
            size_t nl = lName ? lName - 1 : 0;
            if (nl > (size_t)maxLenReadName - 1) nl = (size_t)maxLenReadName - 1;
            rs.names[r].assign(nm, nl);
            const bool fwd = !(flag & 16u);
            const uint32_t L = lSeq > maxReadLength ? maxReadLength : lSeq;
            rs.lengths[r] = L;
            uint32_t *dst = rs.queries.data() + ((size_t)r / 32) * 32 * wpq + r % 32;
            uint8_t *cd = &rs.codes[(size_t)r * maxReadLength];
            char *qd = &rs.quals[(size_t)r * maxReadLength];
            for (uint32_t i = 0; i < L; i++) {
                const uint32_t k = fwd ? i : lSeq - 1 - i;              // the read as sequenced: a reverse-strand record is turned back
                const unsigned char ch = (unsigned char)nt16[(b[seqAt + k / 2] >> (k & 1u ? 0 : 4)) & 15u];
                // (the reference indexes its four-entry complement table with the CHARACTER here, QueryParser.cpp:132 -- out of bounds; the evident intent:
                //  the complement of A / C / G / T, anything else stays what it is, N -> G as everywhere)
                const bool acgt = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
                const uint8_t code = fwd || !acgt ? charMap[ch] : (uint8_t)(3 - charMap[ch]);
                dst[(i / 16) * 32] |= (uint32_t)code << (2 * (i % 16));
                cd[i] = code;
                const unsigned char q = b[qualAt + k];
                qd[i] = q == 0xFF ? (char)0xFF : (char)((int)q + 33 - qualityConstant);
            }
        }
    });
    for (auto &x : th) x.join();
    return true;
}

bool ReadChunker::next(uint32_t maxReads, ReadSet &rs, std::string &err)
{
    err.clear();
    rs.n = 0;                                                            // (a recycled set keeps its arrays: allocateReads sizes and zeroes them again)
    if (bam) return nextBam(maxReads, rs, err);
    const size_t F = files.size();
    if (F == 2) maxReads &= ~1u;
    const uint32_t perFile = F == 2 ? maxReads / 2 : maxReads;
    if (perFile == 0) { err = "ReadChunker::next: bad chunk size"; return false; }
    unsigned char charMap[256];
    fillCharMap(charMap);
    uint32_t w = 1;
    while (w < maxReadLength) w *= 2;
    const uint32_t wpq = w / 16 ? w / 16 : 1;
    std::vector<Region> regs(F);
    std::vector<bool> ok(F, true);
    std::vector<std::string> errs(F);
    // both files of a pair are refilled and scanned at the same time
    auto prepare = [&](size_t f) {
        File &X = *files[f];
        double est = X.bytesPerRead > 0 ? X.bytesPerRead : 2.0 * maxReadLength + 96;
        size_t want = (size_t)std::min<double>((double)perFile * est * 1.05 + 65536.0, 3.5e10);
        // never ask for more than the file still holds (plain files), and approach an unknown size step by step (gzip)
        if (X.fd >= 0) want = (size_t)std::min<uint64_t>(want, (uint64_t)(X.have - X.used) + (X.size - X.pos));
        else if (X.bytesPerRead == 0) want = std::min<size_t>(want, (size_t)1 << 28);
        if (want < 4096) want = 4096;
        for (;;) {
            if (!X.fill(want, errs[f])) { ok[f] = false; return; }
            TextView all{X.buf.data() + X.used, X.have - X.used};
            if (!X.kindKnown) {
                size_t p0 = 0;
                while (p0 < all.size() && all[p0] != '>' && all[p0] != '@') p0++;
                if (p0 < all.size()) { X.fq = all[p0] == '@'; X.kindKnown = true; }
            }
            // the last record of the buffer may be incomplete unless the file has ended: the region stops at its start
            size_t end = all.size();
            if (!X.eof) {
                size_t p = end;
                while (p > 0) {
                    p--;
                    if ((p == 0 || all[p - 1] == '\n') && isRecordStart(all, X.fq, p)) break;
                }
                end = p;
            }
            regs[f].all = TextView{all.p, end};
            regs[f].fq = X.fq;
            if (!scanRegion(regs[f], perFile, maxReadLength, maxLenReadName, qualityConstant, charMap, errs[f])) { ok[f] = false; return; }
            if (regs[f].n >= perFile || X.eof) return;
            // fewer complete records than asked for and more file to come: read more (the estimate was low)
            want = (size_t)((double)want * 1.5) + (1u << 20);
            if (X.fd >= 0) want = (size_t)std::min<uint64_t>(want, (uint64_t)(X.have - X.used) + (X.size - X.pos));
        }
    };
    if (F == 2) {
        std::thread second([&] { prepare(1); });
        prepare(0);
        second.join();
    } else prepare(0);
    for (size_t f = 0; f < F; f++) if (!ok[f]) { err = errs[f]; return false; }
    uint32_t m = perFile;
    for (size_t f = 0; f < F; f++) m = std::min(m, regs[f].n);
    if (m == 0) return false;                                           // end of the input (the reference stops at the shorter file of a pair)
    for (size_t f = 0; f < F; f++) cutRegion(regs[f], m, maxReadLength, maxLenReadName, qualityConstant, charMap);
    allocateReads(rs, (uint32_t)F * m, regs[0].fq, wpq, maxReadLength, false);
    if (F == 2) {
        std::thread second([&] { fillRegion(regs[1], rs, 2, 1, maxReadLength, maxLenReadName, qualityConstant, charMap); });
        fillRegion(regs[0], rs, 2, 0, maxReadLength, maxLenReadName, qualityConstant, charMap);
        second.join();
    } else fillRegion(regs[0], rs, 1, 0, maxReadLength, maxLenReadName, qualityConstant, charMap);
    for (size_t f = 0; f < F; f++) {
        File &X = *files[f];
        X.bytesPerRead = (double)regs[f].consumed / (double)m;
        X.used += regs[f].consumed;
    }
    return true;
}

// the whole input as one chunk (the test tools; the command line streams)
bool loadSingleReads(const std::string &file, uint32_t maxReadLength, int qualityConstant, int maxLenReadName, ReadSet &rs, std::string &err)
{
    ReadChunker c;
    if (!c.open(file, "", maxReadLength, qualityConstant, maxLenReadName, err)) return false;
    if (!c.next(0xFFFFFFE0u, rs, err)) {
        if (!err.empty()) return false;
        uint32_t w = 1;
        while (w < maxReadLength) w *= 2;
        allocateReads(rs, 0, false, w / 16 ? w / 16 : 1, maxReadLength, true);
    }
    return true;
}

// two files, mates interleaved: read 2p from file 1, read 2p+1 from file 2 (loadPairReadsGz2, QueryParser.cpp:177)
bool loadPairReads(const std::string &file1, const std::string &file2, uint32_t maxReadLength, int qualityConstant, int maxLenReadName, ReadSet &rs,
                   std::string &err)
{
    ReadChunker c;
    if (!c.open(file1, file2, maxReadLength, qualityConstant, maxLenReadName, err)) return false;
    if (!c.next(0xFFFFFFE0u, rs, err)) {
        if (!err.empty()) return false;
        uint32_t w = 1;
        while (w < maxReadLength) w *= 2;
        allocateReads(rs, 0, false, w / 16 ? w / 16 : 1, maxReadLength, true);
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

SamWriter::SamWriter(const HostIndex &ix_, const SamOptions &o, std::string *sk) : ix(ix_), opt(o), out(nullptr), sink(sk)
{
    for (int i = 1; i != 256; ++i) gLogN[i] = (int)(4.343 * log((double)i) + 0.5);
    gLogN[0] = 0;
}

void SamWriter::put(const std::string &line)
{
    if (sink) sink->append(line);
    else if (bamFileOf(out)) { if (!bamWriteLines(out, line.data(), line.size())) { fprintf(stderr, "write error on the output file\n"); fflush(stdout); fflush(stderr); _exit(1); } }
    else fwrite(line.data(), 1, line.size(), out);
    nRecords++;
}

void SamWriter::header()
{
    // SAMOutputHeaderConstruct (SAM.cpp:82-137); the @PG line names this program
    std::string h = "@HD\tVN:1.3\tSO:unsorted\n@RG\tID:" + opt.readGroup + "\tSM:" + opt.sampleName + "\t" + opt.readGrpOption + "\n";
    for (int i = 0; i < ix.numOfSeq; i++) h += "@SQ\tSN:" + ix.seqName[i] + "\tLN:" + std::to_string(ix.actualEnd[i] - ix.actualStart[i]) + "\n";
    h += std::string("@PG\tID:SOAP3-dp\tPN:SOAP3-dp\tVN:") + s3_version() + "\n";
    if (bamFileOf(out)) { if (!bamWriteHeader(out, h)) { fprintf(stderr, "write error on the output file\n"); fflush(stdout); fflush(stderr); _exit(1); } }
    else fwrite(h.data(), 1, h.size(), out);
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

// decimal digits appended without a temporary string (the records are made of a dozen small numbers each)
static inline void appendUInt(std::string &s, unsigned long long v)
{
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    const size_t at = s.size();
    s.resize(at + (size_t)n);
    for (int k = 0; k < n; k++) s[at + (size_t)k] = tmp[n - 1 - k];
}

// sequence and quality columns: 2-bit codes -> letters (reverse-complemented for the reverse strand), qualities + 33 (reversed likewise).  Sixteen
// characters per step where the CPU has byte shuffles: these two columns are two thirds of a record, and the formatter shares sixteen CPUs with the loader.
void SamWriter::seqAndQual(const ReadSet &rs, uint32_t r, bool negative, std::string &line) const
{
    const uint32_t L = rs.lengths[r];
    const uint8_t *c = &rs.codes[(size_t)r * rs.maxReadLength];
    const char *q = &rs.quals[(size_t)r * rs.maxReadLength];
    static const char dna[] = "ACGT";
    const bool noQual = L && (unsigned char)q[negative ? L - 1 : 0] == 0xff;
    const size_t at = line.size();
    line.resize(at + L + 1 + (noQual ? 1 : L));
    char *o = &line[at];
    uint32_t i = 0;
#ifdef __SSSE3__
    {
        const __m128i lut = _mm_setr_epi8('A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T');
        const __m128i lutRc = _mm_setr_epi8('T', 'G', 'C', 'A', 'T', 'G', 'C', 'A', 'T', 'G', 'C', 'A', 'T', 'G', 'C', 'A');
        const __m128i rev = _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
        const __m128i three = _mm_set1_epi8(3);
        if (!negative) for (; i + 16 <= L; i += 16) _mm_storeu_si128((__m128i *)(o + i), _mm_shuffle_epi8(lut, _mm_and_si128(_mm_loadu_si128((const __m128i *)(c + i)), three)));
        else for (; i + 16 <= L; i += 16)
            _mm_storeu_si128((__m128i *)(o + i), _mm_shuffle_epi8(lutRc, _mm_shuffle_epi8(_mm_and_si128(_mm_loadu_si128((const __m128i *)(c + L - 16 - i)), three), rev)));
    }
#endif
    if (!negative) for (; i < L; i++) o[i] = dna[c[i]];
    else for (; i < L; i++) o[i] = dna[3 - c[L - 1 - i]];
    o[L] = '\t';
    o += L + 1;
    if (noQual) { o[0] = '*'; return; }
    i = 0;
#ifdef __SSSE3__
    {
        const __m128i rev = _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
        const __m128i add = _mm_set1_epi8(33);
        if (!negative) for (; i + 16 <= L; i += 16) _mm_storeu_si128((__m128i *)(o + i), _mm_add_epi8(_mm_loadu_si128((const __m128i *)(q + i)), add));
        else for (; i + 16 <= L; i += 16) _mm_storeu_si128((__m128i *)(o + i), _mm_add_epi8(_mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(q + L - 16 - i)), rev), add));
    }
#endif
    if (!negative) for (; i < L; i++) o[i] = (char)(q[i] + 33);
    else for (; i < L; i++) o[i] = (char)(q[L - 1 - i] + 33);
}

void SamWriter::unaligned(const ReadSet &rs, uint32_t r)
{
    std::string line = rs.names[r] + "\t4\t*\t0\t0\t*\t*\t0\t0\t";
    seqAndQual(rs, r, false, line);
    line += "\tRG:Z:" + opt.readGroup + "\n";
    put(line);
}

// MD string and average quality at the mismatching positions (getMdStr / mdStr, PE.cpp:209-284,374-420)
int mdString(const HostIndex &ix, const uint8_t *query, const char *qual, uint32_t qlen, uint32_t pos, bool negative, int mismatchNum, int trim,
             std::string &md, int &avgQual)
{
    if (!negative) {
        if (trim > 0) { query += trim; pos += (uint32_t)trim; qlen -= (uint32_t)trim; }
        else if (trim < 0) qlen -= (uint32_t)(-trim);
    } else {
        if (trim > 0) { pos += (uint32_t)trim; qlen -= (uint32_t)trim; }
        else if (trim < 0) { query += -trim; qlen -= (uint32_t)(-trim); }
    }
    avgQual = 20;                                                         // DEFAULT_QUAL_VALUE, PE.h:27
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
            appendUInt(xa, tp);
            xa.push_back(',');
            appendUInt(xa, readLen);
            xa += "M,";
            appendUInt(xa, mis[k]);
            xa.push_back(';');
            if (mis[k] > bestMis) secBestHitNum++;
        }
    }
    std::string md;
    int avgQual = 20;
    const uint8_t *codes = &rs.codes[(size_t)r * rs.maxReadLength];
    const char *quals = &rs.quals[(size_t)r * rs.maxReadLength];
    // bestMis == 9999 cannot reach here: a hit is always chosen
    // (the MD string: printed with -p, its mismatch letters counted for a trimmed read, its qualities averaged for the non-BWA MAPQ -- otherwise nobody
    //  looks at it, and it costs a walk over the read against the text)
    int mdLen = 0;
    if (opt.printMDNM || !opt.bwaLikeScore || (boundTrim && mis[best]))
        mdLen = mdString(ix, codes, quals, fullLen, pos[best], negative, (int)mis[best], boundTrim, md, avgQual);
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
    // (the record is built where it goes: at the end of the thread's buffer when there is one, in a kept string otherwise -- no allocation, no copy)
    std::string &line = sink ? *sink : scratch;
    if (!sink) line.clear();
    line += rs.names[r];
    line += negative ? "\t16\t" : "\t0\t";
    line += ix.seqName[bestChr - 1];
    line.push_back('\t');
    appendUInt(line, bestTP);
    line.push_back('\t');
    appendUInt(line, (unsigned)mapq);
    line.push_back('\t');
    if (!newCigar.empty()) line += newCigar; else { appendUInt(line, fullLen); line.push_back('M'); }
    line += "\t*\t0\t0\t";
    seqAndQual(rs, r, negative, line);
    line += "\tRG:Z:";
    line += opt.readGroup;
    if (opt.printMDNM) { line += "\tNM:i:"; appendUInt(line, outMis); }
    line += "\tX0:i:"; appendUInt(line, (unsigned)bestHitNum);
    line += "\tX1:i:"; appendUInt(line, (unsigned)secBestHitNum);
    line += "\tXM:i:"; appendUInt(line, outMis);
    line += "\tXO:i:0\tXG:i:0";
    if (opt.printMDNM && !md.empty()) { line += "\tMD:Z:"; line += md; }
    if (nOcc > 1 && !xa.empty()) { line += "\tXA:Z:"; line += xa; }
    line.push_back('\n');
    if (sink) nRecords++; else put(line);
}

void SamWriter::singleAnswer(const ReadSet &rs, uint32_t r, uint32_t pos, uint8_t strand, uint8_t mis)
{
    const uint32_t readLen = rs.lengths[r];
    const bool negative = strand == 2;
    uint64_t tp = 0;
    uint32_t chr = 0;
    chrAndPos(pos, tp, chr);
    std::string md;
    int avgQual = 20;
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
    put(line);
}

// ------------------------------------------------------------------------------------------------
// paired-end records
// ------------------------------------------------------------------------------------------------
// one SAM line the way samtools prints a bam1_t: RNAME/RNEXT from tid/mtid ("=" when equal), POS/PNEXT one-based

void SamWriter::emit(const ReadSet &rs, const Rec &c)
{
    // (built where it goes -- at the end of the thread's buffer, or in the kept string -- without temporaries: as singleEnd builds its records)
    std::string &line = sink ? *sink : scratch;
    if (!sink) line.clear();
    line += rs.names[c.r];
    line.push_back('\t'); appendUInt(line, (unsigned)c.flag);
    line.push_back('\t'); if (c.tid < 0) line.push_back('*'); else line += ix.seqName[c.tid];
    line.push_back('\t'); appendUInt(line, (unsigned long long)(c.pos0 + 1));
    line.push_back('\t'); appendUInt(line, (unsigned)c.mapq);
    line.push_back('\t'); if (c.cigar.empty()) line.push_back('*'); else line += c.cigar;
    line.push_back('\t');
    if (c.mtid < 0) line.push_back('*');
    else if (c.tid >= 0 && c.mtid == c.tid) line.push_back('=');
    else line += ix.seqName[c.mtid];
    line.push_back('\t'); appendUInt(line, (unsigned long long)(c.mpos0 + 1));
    line.push_back('\t');
    if (c.isize < 0) { line.push_back('-'); appendUInt(line, (unsigned long long)(-c.isize)); } else appendUInt(line, (unsigned long long)c.isize);
    line.push_back('\t');
    seqAndQual(rs, c.r, c.negative, line);
    line += "\tRG:Z:";
    line += opt.readGroup;
    line += c.tags;
    line.push_back('\n');
    if (sink) nRecords++; else put(line);
}

int bwaLikeSingle(int x0, int x1, const int *gLogN)                // bwaLikeSingleQualScore, BGS-IO.cpp:2310
{
    if (x0 > 1) return 0;
    if (x1 == 0) return 37;
    x1 = x1 > 255 ? 255 : x1;
    const int n = gLogN[x1];
    return 23 < n ? 0 : 23 - n;
}

int tableScore(int mismatchNum, int avgQual, int maxMAPQ, int minMAPQ)
{
    int mi = mismatchNum > 5 ? 5 : mismatchNum;
    int qi = (avgQual - 1) / 20;
    qi = qi > 1 ? 1 : qi < 0 ? 0 : qi;
    int v = (int)(maxMAPQ * kMappingScore[mi][qi]);
    return v < minMAPQ ? minMAPQ : v;
}

void SamWriter::properPair(const ReadSet &rs, uint32_t r1, uint32_t r2, const s3_pair_result &P, const s3_pair_xa *xa, size_t nXa)
{
    const int type = opt.alignmentType;
    const bool mapqMode = type == 1 || type == 2;
    // unique best with several optimal pairings: both ends are written as unmapped (CPUfunctions.cpp:2366-2377, BGS-IO.cpp:3767-3804)
    if (type == 3 && P.numMin != 1) {
        Rec a; a.r = r1; a.flag = 1 | 0x40 | 0x4 | 0x8;
        Rec b; b.r = r2; b.flag = 1 | 0x80 | 0x4 | 0x8;
        emit(rs, a); emit(rs, b);
        return;
    }
    const uint32_t len1 = rs.lengths[r1], len2 = rs.lengths[r2];
    uint64_t tp1 = 0, tp2 = 0;
    uint32_t chr1 = 0, chr2 = 0;
    std::string cig1, cig2, md1, md2;
    const int trim1 = chrAndPosChecked(len1, P.bestA1, tp1, chr1, &cig1);
    const int trim2 = chrAndPosChecked(len2, P.bestA2, tp2, chr2, &cig2);
    const bool neg1 = P.bestS1 == 2, neg2 = P.bestS2 == 2;
    int avg1 = 20, avg2 = 20;
    // (the MD string: printed with -p, its mismatch letters counted for a trimmed read, its qualities averaged for the non-BWA MAPQ -- otherwise nobody looks at
    //  it, and it costs a walk over the read against the text; as in singleEnd)
    if (opt.printMDNM || !opt.bwaLikeScore || (trim1 && P.bestM1))
        mdString(ix, &rs.codes[(size_t)r1 * rs.maxReadLength], &rs.quals[(size_t)r1 * rs.maxReadLength], len1, P.bestA1, neg1, P.bestM1, trim1, md1, avg1);
    if (opt.printMDNM || !opt.bwaLikeScore || (trim2 && P.bestM2))
        mdString(ix, &rs.codes[(size_t)r2 * rs.maxReadLength], &rs.quals[(size_t)r2 * rs.maxReadLength], len2, P.bestA2, neg2, P.bestM2, trim2, md2, avg2);
    int mis1 = P.bestM1, mis2 = P.bestM2;
    if (trim1 && mis1) { mis1 = 0; for (char c : md1) mis1 += c > '9'; }
    if (trim2 && mis2) { mis2 = 0; for (char c : md2) mis2 += c > '9'; }
    int mapq1 = 255, mapq2 = 255;                                       // SAM_MAPQ_UNAVAILABLE for unique / random best
    int x0a = -1, x0b = -1, x1a = -1, x1b = -1;
    if (type == 3) { x0a = x0b = 1; }
    if (mapqMode) {
        x0a = (int)P.x0_1; x1a = (int)P.x1_1; x0b = (int)P.x0_2; x1b = (int)P.x1_2;
        if (opt.bwaLikeScore) {                                         // bwaLikePairQualScore, BGS-IO.cpp:2415-2463
            int op = ((int)(len1 + len2) - (int)P.minTotal) * opt.dpMatchScore + (int)P.minTotal * opt.dpMismatchScore;
            int sub = ((int)(len1 + len2) - (int)(signed char)P.secMinTotal) * opt.dpMatchScore + (int)(signed char)P.secMinTotal * opt.dpMismatchScore;
            int subNum = (int)P.numPairs - (int)P.numMin > 0 ? (int)P.numPairs - (int)P.numMin : 0;
            int q0 = bwaLikeSingle(x0a, x1a, gLogN), q1 = bwaLikeSingle(x0b, x1b, gLogN), qp = 0;
            op *= 10; sub *= 10;
            if (q0 > 0 && q1 > 0) { qp = q0 + q1; if (qp > 60) qp = 60; q0 = q1 = qp; }
            else {
                if (P.numMin == 1) {
                    if (subNum == 0) qp = 29;
                    else if (op - sub > 0.3 * ((int)(len1 + len2) / 2)) qp = 23;
                    else { subNum = subNum > 255 ? 255 : subNum; qp = (op - sub) / 2 - gLogN[subNum]; if (qp < 0) qp = 0; }
                }
                if (q0 == 0) q0 = qp + 7 < q1 ? qp + 7 : q1;
                if (q1 == 0) q1 = qp + 7 < q0 ? qp + 7 : q0;
            }
            mapq1 = q0; mapq2 = q1;
        } else {                                                        // getMapQualScore2 + getMapQualScoreForPair, BGS-IO.cpp:2465,2577
            auto one = [&](int mis, int avg, int x0, int x1, bool isBest) {
                if (x0 == 1 && P.numPairs == 1) { if (!isBest && x1 > 1) return opt.minMAPQ; return tableScore(mis, rs.isFastq ? avg : 20, opt.maxMAPQ, opt.minMAPQ); }
                return opt.minMAPQ;
            };
            const int s1 = one(P.bestM1, avg1, x0a, x1a, P.min1 == P.bestM1), s2 = one(P.bestM2, avg2, x0b, x1b, P.min2 == P.bestM2);
            mapq1 = mapq2 = s1 > s2 ? (int)(s1 * 0.2 + s2 * 0.8) : (int)(s1 * 0.8 + s2 * 0.2);
        }
        if (trim1) mapq1 = 0;
        if (trim2) mapq2 = 0;
    }
    auto xaString = [&](bool first) {
        std::string s;
        if (!mapqMode) return s;
        for (size_t k = 0; k < nXa; k++) {
            uint64_t tp; uint32_t ch;
            chrAndPos(first ? xa[k].a1 : xa[k].a2, tp, ch);
            s += ix.seqName[ch - 1];
            s.push_back(',');
            s.push_back((first ? xa[k].s1 : xa[k].s2) == 2 ? '-' : '+');
            s += std::to_string((unsigned long long)tp);
            s.push_back(',');
            s += std::to_string(first ? len1 : len2);
            s += "M,";
            s += std::to_string((unsigned)(first ? xa[k].m1 : xa[k].m2));
            s.push_back(';');
        }
        return s;
    };
    auto tags = [&](int mis, int x0, int x1, const std::string &md, const std::string &xaz) {
        std::string t;
        if (opt.printMDNM) t += "\tNM:i:" + std::to_string(mis);
        if (x0 >= 0) t += "\tX0:i:" + std::to_string(x0);
        if (x1 >= 0) t += "\tX1:i:" + std::to_string(x1);
        t += "\tXM:i:" + std::to_string(mis) + "\tXO:i:0\tXG:i:0";
        if (opt.printMDNM && !md.empty()) t += "\tMD:Z:" + md;
        if (!xaz.empty()) t += "\tXA:Z:" + xaz;
        return t;
    };
    Rec a, b;
    a.r = r1; a.negative = neg1; a.flag = 1 | 2 | 0x40 | (neg1 ? 0x10 : 0) | (neg2 ? 0x20 : 0);
    a.tid = (int)chr1 - 1; a.pos0 = (long long)tp1 - 1; a.mtid = (int)chr2 - 1; a.mpos0 = (long long)tp2 - 1; a.mapq = mapq1;
    a.cigar = cig1.empty() ? std::to_string(len1) + "M" : cig1;
    if (chr1 == chr2) a.isize = tp2 > tp1 ? (long long)(tp2 + len2 - tp1) : -(long long)(tp1 + len1 - tp2);
    a.tags = tags(mis1, x0a, x1a, md1, xaString(true));
    b.r = r2; b.negative = neg2; b.flag = 1 | 2 | 0x80 | (neg2 ? 0x10 : 0) | (neg1 ? 0x20 : 0);
    b.tid = (int)chr2 - 1; b.pos0 = (long long)tp2 - 1; b.mtid = (int)chr1 - 1; b.mpos0 = (long long)tp1 - 1; b.mapq = mapq2;
    b.cigar = cig2.empty() ? std::to_string(len2) + "M" : cig2;
    if (chr1 == chr2) b.isize = tp1 > tp2 ? (long long)(tp1 + len1 - tp2) : -(long long)(tp2 + len2 - tp1);
    b.tags = tags(mis2, x0b, x1b, md2, xaString(false));
    emit(rs, a); emit(rs, b);
}

void SamWriter::improperPair(const ReadSet &rs, uint32_t r1, uint32_t r2, const uint32_t *pos1, const uint8_t *strand1, const uint8_t *mis1, size_t n1,
                             const uint32_t *pos2, const uint8_t *strand2, const uint8_t *mis2, size_t n2, uint32_t xaLimit)
{
    const int type = opt.alignmentType;
    struct End {
        long best = -1; int bestScore = 0, bestNum = 0, secNum = 0; uint64_t tp = 0; uint32_t chr = 0; int mapq = 0; std::string md, xaz; bool mapped = false;
    } e[2];
    const uint32_t rr[2] = {r1, r2};
    const uint32_t *pos[2] = {pos1, pos2};
    const uint8_t *strand[2] = {strand1, strand2}, *mis[2] = {mis1, mis2};
    const size_t nn[2] = {n1, n2};
    for (int k = 0; k < 2; k++) {
        End &E = e[k];
        if (nn[k] > 0) {                                                 // best = first with the fewest mismatches; X1 counts best + 1 (BGS-IO.cpp:2628-2700)
            E.best = 0; E.bestScore = mis[k][0]; E.bestNum = 1;
            for (size_t i = 1; i < nn[k]; i++) {
                const int m = mis[k][i];
                if (m < E.bestScore) { E.secNum = E.bestScore == m + 1 ? E.bestNum : 0; E.best = (long)i; E.bestScore = m; E.bestNum = 1; }
                else if (m == E.bestScore) E.bestNum++;
                else if (m == E.bestScore + 1) E.secNum++;
            }
        }
        if (E.best >= 0 && (type != 3 || E.bestNum == 1)) {
            E.mapped = true;
            chrAndPos(pos[k][E.best], E.tp, E.chr);
            int avg = 20;
            const uint32_t len = rs.lengths[rr[k]];
            mdString(ix, &rs.codes[(size_t)rr[k] * rs.maxReadLength], &rs.quals[(size_t)rr[k] * rs.maxReadLength], len, pos[k][E.best], strand[k][E.best] == 2,
                     mis[k][E.best], 0, E.md, avg);
            if (type >= 3) E.mapq = 255;
            else {                                                       // getMapQualScoreSingle halved, floored at MinMAPQ (BGS-IO.cpp:2714-2722)
                int q;
                if (opt.bwaLikeScore) q = bwaLikeSingle(E.bestNum, E.secNum, gLogN);
                else if (E.bestNum == 1) q = E.secNum > 0 ? opt.minMAPQ : tableScore(mis[k][E.best], rs.isFastq ? avg : 20, opt.maxMAPQ, opt.minMAPQ);
                else q = opt.minMAPQ;
                q >>= 1;
                E.mapq = q < opt.minMAPQ ? opt.minMAPQ : q;
            }
        } else E.best = -1;
    }
    if (type == 1 || type == 2) {
        for (int k = 0; k < 2; k++) {
            uint32_t total = 1;
            const uint32_t len = rs.lengths[rr[k]];
            for (size_t i = 0; i < nn[k] && total < xaLimit; i++) {
                if ((long)i == e[k].best) continue;
                if (type == 2 && (int)mis[k][i] > e[k].bestScore) continue;
                uint64_t tp; uint32_t ch;
                chrAndPos(pos[k][i], tp, ch);
                e[k].xaz += ix.seqName[ch - 1];
                e[k].xaz.push_back(',');
                e[k].xaz.push_back(strand[k][i] == 2 ? '-' : '+');
                e[k].xaz += std::to_string((unsigned long long)tp) + "," + std::to_string(len) + "M," + std::to_string((unsigned)mis[k][i]) + ";";
                total++;
            }
        }
    }
    for (int k = 0; k < 2; k++) {
        const End &me = e[k], &mate = e[1 - k];
        Rec c;
        c.r = rr[k];
        c.flag = 1 | (k == 0 ? 0x40 : 0x80) | (me.mapped ? 0 : 0x4) | (mate.mapped ? 0 : 0x8);
        const bool negMe = me.mapped && strand[k][me.best] == 2, negMate = mate.mapped && strand[1 - k][mate.best] == 2;
        if (negMe) c.flag |= 0x10;
        if (negMate) c.flag |= 0x20;
        c.negative = negMe;
        // an unmapped end takes the coordinates of its mate (BGS-IO.cpp:2819-2822)
        c.tid = me.chr ? (int)me.chr - 1 : (mate.chr ? (int)mate.chr - 1 : -1);
        c.pos0 = me.tp ? (long long)me.tp - 1 : (mate.tp ? (long long)mate.tp - 1 : -1);
        c.mtid = mate.chr ? (int)mate.chr - 1 : (me.chr ? (int)me.chr - 1 : -1);
        c.mpos0 = mate.tp ? (long long)mate.tp - 1 : (me.tp ? (long long)me.tp - 1 : -1);
        const uint32_t len1 = rs.lengths[r1], len2 = rs.lengths[r2];
        if (e[0].chr > 0 && e[0].chr == e[1].chr) {
            const uint64_t t1 = e[0].tp, t2 = e[1].tp;
            if (k == 0) c.isize = t2 > t1 ? (long long)(t2 + len2 - t1) : -(long long)(t1 + len1 - t2);
            else c.isize = t1 > t2 ? (long long)(t1 + len1 - t2) : -(long long)(t2 + len2 - t1);
        }
        if (me.mapped) {
            const int m = mis[k][me.best];
            c.mapq = me.mapq;
            c.cigar = std::to_string(rs.lengths[rr[k]]) + "M";
            if (opt.printMDNM) c.tags += "\tNM:i:" + std::to_string(m);
            if (type != 4) c.tags += "\tX0:i:" + std::to_string(me.bestNum);
            if (type == 1 || type == 2) c.tags += "\tX1:i:" + std::to_string(me.secNum);
            c.tags += "\tXM:i:" + std::to_string(m) + "\tXO:i:0\tXG:i:0";
            if (opt.printMDNM && !me.md.empty()) c.tags += "\tMD:Z:" + me.md;
            if (!me.xaz.empty()) c.tags += "\tXA:Z:" + me.xaz;
        }
        emit(rs, c);
    }
}

}  // namespace s3host
