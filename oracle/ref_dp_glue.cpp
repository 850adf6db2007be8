// ref_dp_glue.cpp -- TEST INFRASTRUCTURE ONLY (parity oracle), never shipped, never on the product path.
//
// Our own glue that lets the reference's DP ENGINES -- the host halves of DV-DPfunctions.cu (window geometry and packers, result decode,
// CIGAR encoder, seeding, candidate merges, output filters; :742-3911) and their callers DV-SemiDP.cu, DV-DPForBothUnalign.cu,
// DV-DPForSingleReads.cu -- run in this container, compiled with g++ from where they lie (oracle/ref.mk: the line edits are listed there;
// no stand-in headers, no copies).  What those files need from the CUDA side is exactly three things, and this file supplies them as a
// driver would:
//
//   SemiGlobalAligner::{decideConfiguration,init,performAlignment,freeMemory}   DV-DPfunctions.cu:520-741 (cudaMalloc / memcpy / two kernel
//        launches).  Here: the same argument contract on host memory; the two kernels are oracle/s3_oracle_dp.c, the restatement that
//        tests/test_gpu_dp_reference.py holds bit-identical to the reference's own device code (oracle/_ref/libdpref.so) on the MI355X.
//   GPUINDEXUpload                                                             alignment.cu:27 -- nothing to upload.
//   single_1_mismatch_alignment2 (+ single_all_valid_seed_alignment)           alignment.cu:1470-1894 -- the seed search.  Same batching,
//        thread buckets and hostKernelSingle hand-over as the reference; where it launches the BWT kernel, every (seed, case) is handed to
//        the reference's CPU twin instead (ProcessReadDoubleStrand3 through hostKernelThreadWrapperSingle: answer word 0xFFFFFFFE with
//        round 2 skipped; 0xFFFFFFFD for a case without a hit, so that "no alignment" stays honest).
//
// With REFDP_DUMP=<file> every DP problem the engines submit is written out (inputs as the packers built them, outputs as the decode
// stage receives them) -- the vectors of tests/golden/dp3.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <vector>

#include "DV-DPfunctions.h"          // the filtered header (oracle/ref.mk)
#include "CPUfunctions.h"

extern "C" {
    typedef struct { int match, mismatch, gapOpen, gapExtend; } s3o_dp_scores;
    int s3o_dp_align ( const uint32_t * packedDNA, const uint32_t * dnaLengths, unsigned maxDNALength, unsigned maxDPTableLength,
                       const uint32_t * packedReads, const uint32_t * readLengths, unsigned maxReadLength, unsigned n,
                       const uint32_t * clipLt, uint32_t * clipRt, const uint32_t * anchorLeft, const uint32_t * anchorRight,
                       const s3o_dp_scores * sc, const int32_t * cutoffs, int scheme,
                       int32_t * scores, uint32_t * hitLocs, uint32_t * startOffsets, uint32_t * maxScoreCounts, uint8_t * pattern, unsigned patternLen );
}

unsigned long long ProcessReadDoubleStrand3 ( SRAQueryInput * qInput, SRAQueryInput * qInput_neg, SRAModel * aModel, SRAModel * aModel_neg,
        int whichCase, SingleAlgnResult * alignResult );

// ---------------------------------------------------------------------------------------------------------------------------------------
// SemiGlobalAligner on host memory
static pthread_mutex_t dumpMutex = PTHREAD_MUTEX_INITIALIZER;
static FILE * dumpFile = NULL;
static int dumpTried = 0;

SemiGlobalAligner::SemiGlobalAligner ()
{
    n_conf = 0;
    alignmentScheme = 1;
}

void SemiGlobalAligner::decideConfiguration ( int maxReadLength, int maxDNALength, int & maxDPTableLength, int & numOfBlocks,
        int & patternLength, DPParameters & dpPara )
{
    // what the reference decides on a device with room for 64 blocks of full tables (DV-DPfunctions.cu:586-598): scheme 1.
    // REFDP_SCHEME=2 takes the branch of a small device (:600-612) where the window is long enough for it.
    const char * e = getenv ( "REFDP_SCHEME" );
    alignmentScheme = 1;
    numOfBlocks = 64;
    maxDPTableLength = maxDNALength;

    if ( e && atoi ( e ) == 2 && maxDNALength >= maxReadLength * 3 )
    {
        alignmentScheme = 2;
        maxDPTableLength = maxReadLength + maxReadLength / 2 + 8;
    }

    this->maxReadLength = maxReadLength;
    this->maxDPTableLength = maxDPTableLength;
    patternLength = maxReadLength + maxDPTableLength;
}

void SemiGlobalAligner::init ( int batchSize, int maxReadLength, int maxDNALength, int maxDPTableLength, DPParameters & dpPara )
{
    this->batchSize = batchSize;
    this->maxReadLength = maxReadLength;
    this->maxDNALength = maxDNALength;
    this->maxDPTableLength = maxDPTableLength;
    this->dpPara = dpPara;
}

void SemiGlobalAligner::performAlignment ( uint * packedDNASequence, uint * DNALengths, uint * packedReadSequence, uint * readLengths,
        int * cutoffThresholds, int * scores, uint * hitLocs, uint * maxScoreCounts, uchar * pattern, int numOfThreads,
        uint * clipLtSizes, uint * clipRtSizes, uint * anchorLeftLocs, uint * anchorRightLocs )
{
    const unsigned patLen = maxReadLength + maxDPTableLength;
    s3o_dp_scores sc = { dpPara.matchScore, dpPara.mismatchScore, dpPara.openGapScore, dpPara.extendGapScore };
    std::vector<uint32_t> clipRt ( numOfThreads, 0 ), startOffsets ( numOfThreads, 0 );

    // the right-clip array is written by the kernel on the device only; the host copy stays as the packer set it (:308-309, :716-719)
    if ( clipRtSizes ) { memcpy ( clipRt.data (), clipRtSizes, numOfThreads * sizeof ( uint32_t ) ); }

    s3o_dp_align ( packedDNASequence, DNALengths, maxDNALength, maxDPTableLength, packedReadSequence, readLengths, maxReadLength, numOfThreads,
                   clipLtSizes, clipRtSizes ? clipRt.data () : NULL, anchorLeftLocs, anchorRightLocs, &sc, cutoffThresholds, alignmentScheme,
                   scores, hitLocs, startOffsets.data (), maxScoreCounts, pattern, patLen );

    if ( !dumpTried )
    {
        pthread_mutex_lock ( &dumpMutex );

        if ( !dumpTried ) { const char * p = getenv ( "REFDP_DUMP" ); dumpFile = p ? fopen ( p, "w" ) : NULL; dumpTried = 1; }

        pthread_mutex_unlock ( &dumpMutex );
    }

    if ( dumpFile )
    {
        pthread_mutex_lock ( &dumpMutex );

        for ( int t = 0; t < numOfThreads; t++ )
        {
            // P <readLen> <windowLen> <clipLt> <clipRt> <anchorL> <anchorR> <cutoff> <read> <window> | <score> <hitLoc> <ties> <ops>
            // sequences as letters (1-based characters of the packed words, DV-DPfunctions.cu:57-59); ops as the kernel emits them, 'V'<n> runs as (n)
            const unsigned rbase = ( t >> 5 ) * ( ( ( maxReadLength + 15 ) / 16 ) << 5 ) + ( t & 31 ), dbase = ( t >> 5 ) * ( ( ( maxDNALength + 15 ) / 16 ) << 5 ) + ( t & 31 );
            fprintf ( dumpFile, "P %u %u %u %u %u %u %d ", readLengths[t], DNALengths[t], clipLtSizes ? clipLtSizes[t] : 0, clipRtSizes ? clipRtSizes[t] : 0,
                      anchorLeftLocs ? anchorLeftLocs[t] : ( uint ) maxDNALength, anchorRightLocs ? anchorRightLocs[t] : 0, cutoffThresholds[t] );

            for ( unsigned i = 1; i <= readLengths[t]; i++ ) { fputc ( "ACGT"[ ( packedReadSequence[rbase + ( ( i >> 4 ) << 5 )] >> ( ( 15 - ( i & 15 ) ) << 1 ) ) & 3], dumpFile ); }

            fputc ( ' ', dumpFile );

            for ( unsigned i = 1; i <= DNALengths[t]; i++ ) { fputc ( "ACGT"[ ( packedDNASequence[dbase + ( ( i >> 4 ) << 5 )] >> ( ( 15 - ( i & 15 ) ) << 1 ) ) & 3], dumpFile ); }

            fprintf ( dumpFile, " | %d %u %u ", scores[t], hitLocs[t], maxScoreCounts[t] );

            if ( scores[t] >= cutoffThresholds[t] )
            {
                const uchar * p = pattern + ( size_t ) t * patLen;

                for ( unsigned k = 0; p[k]; k++ )
                {
                    if ( p[k] == 'V' ) { fprintf ( dumpFile, "(%u)", ( unsigned ) p[++k] ); if ( p[k] == 0 ) { break; } }
                    else { fputc ( p[k], dumpFile ); }
                }
            }
            else { fputc ( '*', dumpFile ); }

            fputc ( '\n', dumpFile );
        }

        fflush ( dumpFile );
        pthread_mutex_unlock ( &dumpMutex );
    }
}

void SemiGlobalAligner::freeMemory () {}

void refDpDumpNote ( const char * line )
{
    // stage separators in the dump (written by the harness between the DP stages)
    if ( !dumpTried ) { const char * p = getenv ( "REFDP_DUMP" ); dumpFile = p ? fopen ( p, "w" ) : NULL; dumpTried = 1; }

    if ( dumpFile ) { fprintf ( dumpFile, "# %s\n", line ); fflush ( dumpFile ); }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
void GPUINDEXUpload ( Soap3Index * index, uint ** _bwt, uint ** _occ, uint ** _revBwt, uint ** _revOcc )
{
    *_bwt = *_occ = *_revBwt = *_revOcc = NULL;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// the seed search: single_all_valid_seed_alignment (alignment.cu:1470-1836) with the kernel launches replaced by the CPU twin
static FILE * seedDump = NULL;
static int seedDumpTried = 0;

static void seedAlignment ( unsigned int * queries, unsigned int * readLengths, uint numMismatch, unsigned int maxReadLength, unsigned int wordPerQuery,
                            unsigned int maxBatchSize, unsigned int numQueries, Soap3Index * index, unsigned long long & numOfAnswer,
                            unsigned int & numOfAlignedRead, unsigned int cpuNumThreads, SingleAlgnResultArray * alignResultArray, unsigned int maxHitNum,
                            unsigned char outNoAlign, unsigned char * noAlignment, unsigned int * readIDs )
{
    numOfAnswer = 0;
    numOfAlignedRead = 0;

    if ( numQueries == 0 ) { return; }

    uint numCases, sa1, sa2, wpa, wpa2;
    char skip2;
    getParametersForThisMismatch2 ( numMismatch, numCases, sa1, sa2, skip2, wpa, wpa2 );
    static HostKernelArguements hk[MAX_NUM_CPU_THREADS];
    pthread_t threads[MAX_NUM_CPU_THREADS];
    static SRAModel * M[MAX_READ_LENGTH + 1], * Mn[MAX_READ_LENGTH + 1];
    setHostKernelArguments2 ( hk, threads, index, maxReadLength, wordPerQuery, wpa, cpuNumThreads );

    for ( uint t = 0; t < cpuNumThreads; t++ )
    {
        hk[t].SRAMismatchModel = M; hk[t].SRAMismatchModel_neg = Mn;
        hk[t].readLengths = readLengths;
        hk[t].maxHitNum = maxHitNum;
        hk[t].readIDs = readIDs;
        hk[t].sraQuerySettings.MaxError = numMismatch;
        hk[t].sraQuerySettings.MaxNBMismatch = 0;
        hk[t].numCases = numCases;
        hk[t].sa_range_allowed_1 = sa1;
        hk[t].sa_range_allowed_2 = sa2;
        hk[t].maxNumMismatch = numMismatch;
        hk[t].alignResult = alignResultArray->array[t];
    }

    memset ( M, 0, sizeof ( M ) ); memset ( Mn, 0, sizeof ( Mn ) );

    for ( uint r = 0; r < numQueries; r++ )
    {
        uint L = readLengths[r];

        if ( M[L] == NULL )
        {
            M[L] = SRAModelConstruct ( L, QUERY_POS_STRAND, &hk[0].sraQuerySettings, index->sraIndex, SRA_MODEL_16G );
            Mn[L] = SRAModelConstruct ( L, QUERY_NEG_STRAND, &hk[0].sraQuerySettings, index->sraIndex, SRA_MODEL_16G );
        }
    }

    size_t roundUp = ( ( size_t ) maxBatchSize + 31 ) / 32 * 32;
    uint * answers[MAX_NUM_CASES];

    for ( uint c = 0; c < numCases; c++ ) { answers[c] = ( uint * ) malloc ( roundUp * wpa * sizeof ( uint ) ); }

    uint zeros[MAX_NUM_CASES];
    memset ( zeros, 0, sizeof ( zeros ) );
    // scratch for the dry run that tells "hit" from "no hit" per (seed, case)
    SingleAlgnResult * scratch = constructSingleAlgnResult ();
    SRAQueryResultCount rOut;
    memset ( &rOut, 0, sizeof ( rOut ) );
    char dummyQuality[MAX_READ_LENGTH];
    memset ( dummyQuality, 1, sizeof ( dummyQuality ) );
    SRAQueryInfo qiP, qiN;
    SRAQueryInput qP, qN;
    memset ( &qiP, 0, sizeof ( qiP ) ); memset ( &qiN, 0, sizeof ( qiN ) );
    qiP.ReadStrand = QUERY_POS_STRAND; qiP.ReadQuality = dummyQuality;
    qiN.ReadStrand = QUERY_NEG_STRAND; qiN.ReadQuality = dummyQuality;
    qP.QueryInfo = &qiP; qP.QuerySetting = &hk[0].sraQuerySettings; qP.AlgnmtIndex = &hk[0].sraIndex; qP.QueryOutput = &rOut;
    qN.QueryInfo = &qiN; qN.QuerySetting = &hk[0].sraQuerySettings; qN.AlgnmtIndex = &hk[0].sraIndex; qN.QueryOutput = &rOut;
    unsigned char fwd[MAX_READ_LENGTH], rev[MAX_READ_LENGTH];
    unsigned long long queriesLeft = numQueries;

    while ( queriesLeft > 0 )
    {
        uint processedQuery = numQueries - queriesLeft;
        unsigned long long batchSize = queriesLeft > maxBatchSize ? maxBatchSize : queriesLeft;
        uint bucket[MAX_NUM_CPU_THREADS];
        uint rough = batchSize / cpuNumThreads, left = batchSize;

        for ( uint t = 0; t + 1 < cpuNumThreads; t++ ) { bucket[t] = rough; left -= rough; }

        bucket[cpuNumThreads - 1] = left;

        for ( uint c = 0; c < numCases; c++ ) { memset ( answers[c], 0xFF, roundUp * wpa * sizeof ( uint ) ); }

        for ( unsigned long long j = 0; j < batchSize; j++ )
        {
            unsigned long long readId = processedQuery + j;
            int L = readLengths[readId];
            retrieve2BWTQueryFormat ( queries, readId, wordPerQuery, fwd );

            for ( int i = 0; i < L; i++ ) { rev[i] = soap3DnaComplement[fwd[L - i - 1]]; }

            qiP.ReadCode = fwd; qiP.ReadLength = L;
            qiN.ReadCode = rev; qiN.ReadLength = L;
            char isNoAlign = 1;

            for ( uint c = 0; c < numCases; c++ )
            {
                resetSingleAlgnResult ( scratch );
                rOut.TotalOccurrences = 0;
                unsigned long long got = ProcessReadDoubleStrand3 ( &qP, &qN, M[L], Mn[L], c, scratch );
                uint * ans = answers[c] + j / 32 * 32 * wpa + j % 32;
                ans[0] = got ? 0xFFFFFFFE : 0xFFFFFFFD;

                if ( got ) { isNoAlign = 0; }
            }

            if ( outNoAlign == 1 && isNoAlign ) { noAlignment[readId] = 1; }
        }

        uint done = 0;

        for ( uint t = 0; t < cpuNumThreads; t++ )
        {
            hk[t].batchFirstReadId = processedQuery;
            hk[t].skipFirst = done;
            hk[t].numQueries = bucket[t];
            hk[t].queries = queries;
            hk[t].word_per_query = wordPerQuery;
            hk[t].answers = answers;
            hk[t].alignedOcc = 0;
            hk[t].alignedReads = 0;
            hk[t].badReadIndices = NULL;
            hk[t].badAnswers = NULL;
            hk[t].badStartOffset = zeros;
            hk[t].badCountOffset = zeros;
            hk[t].skip_round_2 = 1;

            if ( pthread_create ( &threads[t], NULL, hostKernelThreadWrapperSingle, ( void * ) &hk[t] ) ) { fprintf ( stderr, "pthread_create failed\n" ); exit ( 1 ); }

            done += bucket[t];
        }

        for ( uint t = 0; t < cpuNumThreads; t++ )
        {
            pthread_join ( threads[t], NULL );
            numOfAnswer += hk[t].alignedOcc;
            numOfAlignedRead += hk[t].alignedReads;
        }

        queriesLeft -= batchSize;
    }

    freeSingleAlgnResult ( scratch );

    for ( int i = 0; i <= MAX_READ_LENGTH; i++ )
    {
        if ( M[i] ) { SRAModelFree ( M[i] ); M[i] = NULL; }

        if ( Mn[i] ) { SRAModelFree ( Mn[i] ); Mn[i] = NULL; }
    }

    for ( uint t = 0; t < cpuNumThreads; t++ ) { OCCFree ( hk[t].occ ); }

    for ( uint c = 0; c < numCases; c++ ) { free ( answers[c] ); }
}

// alignment.cu:1839-1894
void single_1_mismatch_alignment2 ( unsigned int * queries, unsigned int * readLengths, unsigned int maxReadLength, unsigned int wordPerQuery,
                                    unsigned int maxBatchSize, unsigned int numQueries, Soap3Index * index, uint * _bwt, unsigned int * _revBwt,
                                    unsigned int * _occ, unsigned int * _revOcc, unsigned long long & numOfAnswer, unsigned int & numOfAlignedRead,
                                    unsigned int cpuNumThreads, SingleAlgnResultArray * alignResultArray, unsigned int maxHitNum )
{
    for ( unsigned int t = 0; t < cpuNumThreads; ++t ) { resetSingleAlgnResult ( alignResultArray->array[t] ); }

    unsigned char * noAlignment = ( unsigned char * ) calloc ( numQueries + 1, 1 );
    seedAlignment ( queries, readLengths, 0, maxReadLength, wordPerQuery, maxBatchSize, numQueries, index, numOfAnswer, numOfAlignedRead, cpuNumThreads,
                    alignResultArray, maxHitNum, 1, noAlignment, NULL );
    unsigned long long roundUp = ( ( unsigned long long ) numQueries + 31 ) / 32 * 32;
    unsigned int * queries2 = ( unsigned int * ) malloc ( sizeof ( unsigned int ) * roundUp * wordPerQuery );
    unsigned int * readLengths2 = ( unsigned int * ) malloc ( sizeof ( unsigned int ) * roundUp );
    memcpy ( queries2, queries, sizeof ( unsigned int ) * roundUp * wordPerQuery );
    memcpy ( readLengths2, readLengths, sizeof ( unsigned int ) * numQueries );
    unsigned int numUnAligned;
    unsigned int * readIDs2 = packReads2 ( queries2, readLengths2, noAlignment, wordPerQuery, numQueries, numUnAligned );
    unsigned int aligned2 = 0;
    unsigned long long answer2 = 0;
    seedAlignment ( queries2, readLengths2, 1, maxReadLength, wordPerQuery, maxBatchSize, numUnAligned, index, answer2, aligned2, cpuNumThreads,
                    alignResultArray, maxHitNum, 0, noAlignment, readIDs2 );
    numOfAnswer += answer2;
    numOfAlignedRead += aligned2;

    if ( !seedDumpTried ) { const char * p = getenv ( "REFDP_SEED_DUMP" ); seedDump = p ? fopen ( p, "w" ) : NULL; seedDumpTried = 1; }

    if ( seedDump )
    {
        // one line per seeding call: how many seeds, how many found a place exactly / with one mismatch
        fprintf ( seedDump, "S %u seeds, %u without an exact hit, %u aligned, %llu hits\n", numQueries, numUnAligned, numOfAlignedRead, numOfAnswer );
        fflush ( seedDump );
    }

    free ( queries2 );
    free ( readLengths2 );
    free ( readIDs2 );
    free ( noAlignment );
}
