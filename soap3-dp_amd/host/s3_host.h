// s3_host.h -- host side of the drop-in `soap3-dp` program above the C-ABI (include/s3dp.h):
// index files, read files, SAM records.  Mirrors the reference's host interfaces for this path:
//   IndexHandler.cpp (INDEXLoad), QueryParser.cpp (loadSingleReadsGz), BGS-IO.cpp (OCCOutputSAMAPI,
//   noAnsOutputSAMAPI, getChrAndPos, BoundaryCheck, MAPQ), SAM.cpp (SAMOutputHeaderConstruct).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <memory>
#include <utility>
#include "../../include/s3dp.h"

namespace s3host {

struct Translate { uint32_t startPos, chrID, correction; };

// vector whose resize() leaves new elements uninitialised: the big per-read arrays are sized once and then filled (and, where zeros are part of the
// content, zeroed) by all host threads, which also spreads the page faults
template <class T> struct DefaultInit : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInit<U>; };
    template <class U, class... A> void construct(U *p, A &&...a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
using BigWords = std::vector<uint32_t, DefaultInit<uint32_t>>;   // file payloads of gigabytes: sized, then filled by all host threads (no zero fill first)

struct HostIndex {
    // search structures (file payloads, headers stripped)
    BigWords bwt, occ, rbwt, rocc, sa;
    std::vector<uint32_t> pac, lkt, rlkt;
    uint32_t inverseSa0 = 0, rInverseSa0 = 0, cum[5] = {0, 0, 0, 0, 0}, rcum[5] = {0, 0, 0, 0, 0};
    uint32_t n = 0;
    uint32_t saInterval = 1;                             // SaValueFreq of the .sa file (1: full suffix array)
    // annotation (HSPLoad, 2bwt-lib/HSP.c:256-531)
    uint32_t dnaLength = 0;
    int numOfSeq = 0;
    std::vector<std::string> seqName;
    std::vector<uint32_t> seqStart, seqEnd;              // seqOffset[].startPos / endPos (inclusive)
    std::vector<uint32_t> actualStart, actualEnd;        // seqActualOffset[]
    uint32_t numOfRemovedSegment = 0;
    std::vector<uint32_t> ambiguityMap;                  // grid: position >> 18 -> translate entry
    std::vector<Translate> translate;
    s3_index_desc desc() const;
};

bool loadIndex(const std::string &prefix, HostIndex &ix, std::string &err);
int hostThreads();                                       // S3DP_HOST_THREADS or the hardware concurrency (1..256)
void buildLookupTable(const std::vector<uint32_t> &pacWords, uint32_t n, bool reversed, std::vector<uint32_t> &table);

template <class T> using RawVec = std::vector<T, DefaultInit<T>>;

struct ReadSet {
    RawVec<uint32_t> queries;                            // reference packed layout
    std::vector<uint32_t> lengths;
    std::vector<std::string> names;
    RawVec<uint8_t> codes;                               // n x maxReadLength, 2-bit codes
    RawVec<char> quals;                                  // n x maxReadLength, Phred (char - 33 / 64)
    uint32_t n = 0, wpq = 0, maxReadLength = 0;
    bool isFastq = false;
};

// The read files as a stream of chunks (the reference's loader thread fills the next batch while the current one is aligned, aio_thread.cpp:582,
// SOAP3-DP.cu:607-800): next() returns the next at most maxReads reads (pairs: reads 2p / 2p+1 from file 1 / file 2, maxReads rounded down to even),
// parsed by all host threads; false with an empty err at the end of the input (for pairs: of the shorter file).
class ReadChunker {
public:
    ReadChunker();
    ~ReadChunker();
    bool open(const std::string &file1, const std::string &file2 /* empty: single reads */, uint32_t maxReadLength, int qualityConstant, int maxLenReadName,
              std::string &err);
    // reads from ONE BAM file (the reference's -bam: loadBAMReads, QueryParser.cpp:27-171): every record is a read -- for pairs the mates follow each
    // other --, a record with the reverse-strand flag is taken back to the read as sequenced (reverse complement, qualities reversed)
    bool openBam(const std::string &file, bool paired, uint32_t maxReadLength, int qualityConstant, int maxLenReadName, std::string &err);
    bool next(uint32_t maxReads, ReadSet &rs, std::string &err);
private:
    struct File;
    std::vector<std::unique_ptr<File>> files;
    bool bam = false, bamPaired = false;
    bool nextBam(uint32_t maxReads, ReadSet &rs, std::string &err);
    uint32_t maxReadLength = 0;
    int qualityConstant = 33, maxLenReadName = 64;
};

bool loadSingleReads(const std::string &file, uint32_t maxReadLength, int qualityConstant, int maxLenReadName, ReadSet &rs, std::string &err);
// two files, mates interleaved: read 2p from file 1, read 2p+1 from file 2 (loadPairReadsGz2, QueryParser.cpp:177)
bool loadPairReads(const std::string &file1, const std::string &file2, uint32_t maxReadLength, int qualityConstant, int maxLenReadName, ReadSet &rs,
                   std::string &err);

struct SamOptions {
    std::string readGroup, sampleName, readGrpOption;
    int alignmentType = 2;                               // OUTPUT_ALL_BEST
    int minMAPQ = 1, maxMAPQ = 40, bwaLikeScore = 1;
    int dpMatchScore = 1, dpMismatchScore = -2;          // only for the BWA-like pair MAPQ (BGS-IO.cpp:3550-3554)
    bool printMDNM = false;
};

// one DP result of a read, as the DP engines return it (SingleAlgnmtResult, PEAlgnmt.h:434); cigar in the engine's dialect ('m' = mismatch run)
// fromDP = false: a hit of the BWT stage carried into the pair leftover records (inputSoap3AnsToArray, PEAlgnmt.cpp:1180: score = L x match + mismatches x
// mismatch score, editdist = mismatches, cigar "<L>M")
struct DpHit { uint32_t pos; uint8_t strand; int score, editdist; std::string cigar; bool fromDP = true; };

// one result of the default DP for a pair (AlgnmtDPResult, PEAlgnmt.h:384): one end from the BWT stage (score = mismatches), the other from the
// DP (score = DP score; cigar / editdist / numSameScore belong to it); whichFromDP 0: first end, 1: second end, 2: the DP found nothing
struct PairDpHit { uint32_t pos1, pos2; uint8_t strand1, strand2, whichFromDP; int score1, score2, editdist, insertSize, numSameScore; std::string cigar; };
// one result of the deep DP (DeepDPAlignResult, PEAlgnmt.h:409): both ends from the DP
struct DeepDpHit { uint32_t pos1, pos2; uint8_t strand1, strand2; int score1, score2, editdist1, editdist2, insertSize, numSame1, numSame2; std::string cigar1, cigar2; };
// OutputBuffer::arrayCopyNRemoveDuplicate (OutputDPResult.cpp; called by OutputBuffer::ready()): a pair's deep-DP results sorted by the first end's position
// ALONE, the first result of every position kept.  The reference sorts with std::sort, and so does this: among more than sixteen results the survivor of
// several with equal first positions is whatever libstdc++'s introsort leaves in front (up to sixteen: insertion sort, i.e. the first in input order), and
// X0 / X1 / XA follow from it.  Same library, same comparator, same input order: same survivor.  tests/test_host_dp_sam_cpu.py pins the survivors of a
// 40-result case to a golden list, so that a toolchain whose sort permutes differently shows up in the CPU suite.
void keepFirstResultPerPosition(std::vector<DeepDpHit> &hits);
// what the BWT stage left for a read in hspaux->x0_array / x1_array / mismatch_array (CPUfunctions.cpp:2099-2103): X0, X1, fewest mismatches; zeros without a hit
struct BwtEndStats { int x0 = 0, x1 = 0, mismatch = 0; };

class SamWriter {
public:
    SamWriter(const HostIndex &ix, const SamOptions &opt, FILE *out);
    // records are appended to *sink instead of a file: one writer + buffer per formatting thread, buffers written in read order
    SamWriter(const HostIndex &ix, const SamOptions &opt, std::string *sink);
    void header();
    // one read of the single-end path: occurrence list in the CPU path's final order (OCCOutputSAMAPI, BGS-IO.cpp:5556)
    // checkLen: the length the reference hands to the boundary check and writes into the XA:Z CIGARs -- the SEARCHED length, i.e. the
    // seed length in long-read mode (hostKernel passes its local readLength, CPUfunctions.cpp:1684,1938); 0 = the read length
    void singleEnd(const ReadSet &rs, uint32_t r, const uint32_t *pos, const uint8_t *strand, const uint8_t *mis, size_t nOcc, uint32_t checkLen = 0);
    // unique-best / random-best record: one hit, X0 = 1, no X1, no XA, no boundary check (SingleAnsOutputSAMAPI, BGS-IO.cpp:5774)
    void singleAnswer(const ReadSet &rs, uint32_t r, uint32_t pos, uint8_t strand, uint8_t mis);
    void unaligned(const ReadSet &rs, uint32_t r);       // noAnsOutputSAMAPI, BGS-IO.cpp:5829
    // the DP results of one single read (SingleDPOutputSAMAPI, BGS-IO.cpp:5857); hits[0].pos == 0xFFFFFFFF or n == 0: unaligned record
    void singleDP(const ReadSet &rs, uint32_t r, const DpHit *hits, size_t n, int singleDPcutoffThreshold);
    // the default-DP results of one pair, in the engine's order after OutputBuffer::ready: choice of the reported result (outputRead2,
    // OutputDPResult.cpp:217-447) and the two records (pairDPOutputSAMAPI, BGS-IO.cpp:4504-5554).  r1 = first read of the pair
    void pairDP(const ReadSet &rs, uint32_t r1, const PairDpHit *hits, size_t n, const BwtEndStats st[2]);
    // the deep-DP results of one pair (outputDeepDPResult2, OutputDPResult.cpp:564-845; pairDeepDPOutputSAMAPI, BGS-IO.cpp:3824-4502)
    void pairDeepDP(const ReadSet &rs, uint32_t r1, const DeepDpHit *hits, size_t n, const BwtEndStats st[2]);
    // a pair no stage could pair: each end with the hits it has on its own -- from the single-end DP or from the BWT stage -- scored like single
    // reads (outputSingleResultForPairEnds, OutputDPResult.cpp:1062; unproperlypairDPOutputSAMAPI, BGS-IO.cpp:2932-3390)
    void unpairedDP(const ReadSet &rs, uint32_t r1, const DpHit *hits1, size_t n1, const DpHit *hits2, size_t n2, int singleDPcutoffThreshold);
    static std::string convertCigar(const std::string &engineCigar, int *deletedEnd = nullptr);                             // convertToCigarStr, PE.cpp:421
    // a pair with a proper pairing (pairOutputSAMAPI, BGS-IO.cpp:3478): reads r1 = 2p, r2 = 2p+1; xa = the other optimal pairings
    void properPair(const ReadSet &rs, uint32_t r1, uint32_t r2, const s3_pair_result &P, const s3_pair_xa *xa, size_t nXa);
    // a pair without one (unproperlypairOutputSAMAPI, BGS-IO.cpp:2582): the occurrence lists of both ends; xaLimit = the hit limit per end
    void improperPair(const ReadSet &rs, uint32_t r1, uint32_t r2, const uint32_t *pos1, const uint8_t *strand1, const uint8_t *mis1, size_t n1,
                      const uint32_t *pos2, const uint8_t *strand2, const uint8_t *mis2, size_t n2, uint32_t xaLimit);
    uint64_t records() const { return nRecords; }
    void flush() { if (out) fflush(out); }

private:
    uint32_t chrAndPos(uint32_t ambPos, uint64_t &tp, uint32_t &chr) const;                    // getChrAndPos, BGS-IO.cpp:1746
    int chrAndPosChecked(uint32_t readLen, uint32_t ambPos, uint64_t &tp, uint32_t &chr, std::string *cigar) const;   // :1981
    void seqAndQual(const ReadSet &rs, uint32_t r, bool negative, std::string &line) const;
    int boundaryCheckDP(uint32_t readLen, uint32_t pacPos, const std::string &cigar, uint64_t &tp, uint32_t &chr, std::string *newCigar) const;
    int misInfoDP(const ReadSet &rs, uint32_t r, uint32_t pos, bool negative, const std::string &engineCigar, int trim, std::string &md, int &numMis,
                  int &gapOpen, int &gapExt, int &avgQual) const;
    struct Rec {
        uint32_t r = 0;
        int flag = 0, tid = -1, mtid = -1, mapq = 0;
        long long pos0 = -1, mpos0 = -1, isize = 0;
        std::string cigar;                               // empty = "*"
        bool negative = false;
        std::string tags;                                // after RG:Z
    };
    struct EndView;
    int mapqSingleDP(int maxDP, int avgMismatchQual, int x0, int x1t1, int x1t2, int bestDP, int secDP, int cutoff) const;
    void pairRecords(const ReadSet &rs, uint32_t r1, EndView *e, int bestInsert, const std::string xa[2]);
    void emit(const ReadSet &rs, const Rec &rec);
    const HostIndex &ix;
    SamOptions opt;
    FILE *out;
    std::string *sink = nullptr;
    std::string scratch;                                 // the record being built (kept: no allocation per record)
    void put(const std::string &line);
    int gLogN[256];
    uint64_t nRecords = 0;
};

// shared by the record writers (s3_host.cpp)
int mdString(const HostIndex &ix, const uint8_t *query, const char *qual, uint32_t qlen, uint32_t pos, bool negative, int mismatchNum, int trim, std::string &md,
             int &avgQual);                                  // getMdStr, PE.cpp
int bwaLikeSingle(int x0, int x1, const int *gLogN);         // bwaLikeSingleQualScore, BGS-IO.cpp:2310
int tableScore(int index, int avgQual, int maxMAPQ, int minMAPQ);   // maxMAPQ * mapping_score[min(index, 5)][quality class], floor minMAPQ

}  // namespace s3host
