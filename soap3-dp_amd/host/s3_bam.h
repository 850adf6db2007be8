// s3_bam.h -- BAM output of the command line (`-b 3`): the reference hands every record to samtools (samopen(.., "wb"), SOAP3-DP.cu:414; samwrite), which
// writes BAM instead of SAM text.  Here the records exist as SAM text (s3_host.cpp::SamWriter); this file turns SAM lines into BAM records and BAM
// bytes into BGZF blocks (SAM/BAM specification 1.3, as samtools-0.1.18 implements it: bam.h, bgzf.c).  BGZF blocks are independent, so every host
// thread compresses its own part of a chunk and the parts are written behind each other.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

namespace s3host {

// SAM lines [text, text + n) -> BAM records appended to `raw` (reference names in header order).  Returns false on a line that is not SAM.
bool samTextToBam(const std::vector<std::string> &refNames, const char *text, size_t n, std::string &raw, std::string &err);
// BAM header block (magic, header text, reference dictionary), uncompressed
void bamHeaderBytes(const std::string &headerText, const std::vector<std::string> &refNames, const std::vector<uint32_t> &refLengths, std::string &raw);
// bytes -> BGZF blocks (at most 0xff00 bytes each) appended to `out`; level: zlib's (-1 = default, as samtools)
bool bgzfCompress(const char *data, size_t n, std::string &out, int level = -1);
// the empty block that ends a BAM file (bgzf.c: 28 bytes)
const std::string &bgzfEofBlock();

// FILE-side bookkeeping: a file registered here gets BAM instead of text from SamWriter (header(), put()) and from the command line's parallel formatter
struct BamFile {
    FILE *f = nullptr;
    std::vector<std::string> refNames;
    std::vector<uint32_t> refLengths;
    std::string pending;                 // BAM bytes not yet compressed (records written one at a time)
    int level = -1;
};
BamFile *bamFileOf(FILE *f);                                             // nullptr: the file takes SAM text
void bamRegister(FILE *f, const std::vector<std::string> &refNames, const std::vector<uint32_t> &refLengths, int level);
bool bamWriteHeader(FILE *f, const std::string &headerText);
bool bamWriteLines(FILE *f, const char *text, size_t n);                 // buffered: compressed and written in whole blocks
bool bamFlush(FILE *f);                                                  // what is pending goes out (before somebody writes behind it)
bool bamClose(FILE *f);                                                  // flush + end-of-file block; the caller closes the FILE

}  // namespace s3host
