// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (parity oracle), never shipped, never on the product path.
//
// Our own driver around the *reference's* CPU alignment path, linked against the reference's
// objects compiled where they lie (see oracle/ref.mk).  It stands in for main()/alignment.cu
// (CUDA translation units that cannot be built here) and routes every read through the
// reference's CPU twin of the GPU kernel:
//      ProcessReadDoubleStrand2            CPUfunctions.cpp:555
//      BWTModelSwitch*2 / *_Lookup2 / _CE2 BGS-HostAlgnmtAlgo2.cpp:36-1728
//      hostKernelThreadWrapper             CPUfunctions.cpp:3022   (mode "sam")
//
// Modes
//   models <out>                                  dump SRAModel case/step tables (SRA2BWTMdl.c:1171,4411)
//   search <ini> <index> <reads> <L> <mm> <out>   per read / per case SA ranges + CE hits, reference order
//   sam    <ini> single <index> <reads> [soap3-dp options]   all-best / all-valid single-end SAM
//          (mirrors all_best_single_alignment, alignment.cu:1422-1465, and all_valid_alignment,
//           alignment.cu:534-1115, with every (read,case) answer marked "resolve on CPU")
//          With the DP stages on (no -s) the reference's own DP engines run behind the ladder (soap3_dp_pair_align /
//          soap3_dp_single_align, alignment.cu:1896-2700): new default DP, default DP, deep DP, single-end DP of the leftovers;
//          they are the reference's host code compiled by oracle/ref.mk, with oracle/ref_dp_glue.cpp in the device's place.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>
#include <zlib.h>
#include <vector>
#include <algorithm>
#include <string>

#include "CPUfunctions.h"
#include "QueryParser.h"
#include "IndexHandler.h"
#include "SAM.h"
#include "OutputDPResult.h"
#include "2bwt-lib/Timing.h"
#include "DV-SemiDP.h"
#include "DV-DPForBothUnalign.h"
#include "DV-DPForSingleReads.h"

void refDpDumpNote ( const char * line );      // oracle/ref_dp_glue.cpp

// defined (non-static) in CPUfunctions.cpp but not declared in its header
unsigned long long ProcessReadDoubleStrand2 ( SRAQueryInput * qInput, SRAQueryInput * qInput_neg, SRAModel * aModel, SRAModel * aModel_neg,
        int whichCase, SAList * sa_list, OCCList * occ_list );

static double g_hostKernelSeconds = 0;
static unsigned long long g_hostKernelReads = 0, g_hostKernelPasses = 0;

static void dumpModels ( const char * outName )
{
    FILE * f = fopen ( outName, "w" );
    SRASetting st;
    memset ( &st, 0, sizeof ( st ) );
    st.ErrorType = SRA_STEP_ERROR_TYPE_MISMATCH_ONLY;
    st.OutputType = SRA_REPORT_ALL;
    st.MaxNBMismatch = 0;
    SRAIndex idx;
    memset ( &idx, 0, sizeof ( idx ) );
    SRAModel * m = ( SRAModel * ) malloc ( sizeof ( SRAModel ) );

    for ( int L = 25; L <= 1024; L++ )
    {
        for ( int e = 0; e <= 4; e++ )
        {
            st.MaxError = e;
            SRAModelPopulate ( m, L, QUERY_POS_STRAND, &st, &idx, SRA_MODEL_16G );

            for ( int c = 0; c < MAX_NUM_OF_SRA_CASES; c++ )
            {
                if ( m->cases[c].type != SRA_CASE_TYPE_ALIGNMENT ) { break; }

                fprintf ( f, "L %d E %d C %d id %d :", L, e, c, m->cases[c].id );

                for ( int s = 0; s < MAX_NUM_OF_SRA_STEPS; s++ )
                {
                    SRAStep * p = &m->cases[c].steps[s];

                    if ( p->type == SRA_STEP_TYPE_COMPLETE ) { break; }

                    fprintf ( f, " [%d %d %d ce %d %d %d err %d %d %d]", ( int ) p->type, p->start, p->end, p->ceThreshold, p->ceStart, p->ceEnd,
                              ( int ) p->ErrorType, ( int ) p->MinError, ( int ) p->MaxError );
                }

                // leftMostAligned is derived data; a checksum over the positions the case visits pins it
                unsigned long long h = 1469598103934665603ULL;

                for ( int s = 0; s < MAX_NUM_OF_SRA_STEPS; s++ )
                {
                    SRAStep * p = &m->cases[c].steps[s];

                    if ( p->type == SRA_STEP_TYPE_COMPLETE ) { break; }

                    for ( int q = 0; q < p->len; q++ )
                    {
                        int pos = p->start + p->step * q;
                        h = ( h ^ ( unsigned long long ) ( unsigned int ) m->cases[c].leftMostAligned[pos] ) * 1099511628211ULL;
                    }
                }

                fprintf ( f, " lma %016llx\n", h );
            }
        }
    }

    free ( m );
    fclose ( f );
}

struct Reads
{
    uint * queries;
    uint * readLengths;
    uint * readIDs;
    char * upkdQualities;
    char * upkdQueryNames;
    uint numQueries;
    uint wordPerQuery;
    uint maxReadLength;
    char isFastq;
};

static Reads loadReads ( const char * fileName, unsigned char * charMap, uint maxReadLength, int maxLenReadName, uint maxNumQueries )
{
    Reads R;
    R.maxReadLength = maxReadLength;
    uint wpq = 1;

    while ( wpq < maxReadLength ) { wpq *= 2; }

    R.wordPerQuery = wpq / CHAR_PER_WORD;
    size_t roundUp = ( ( size_t ) maxNumQueries + 31 ) / 32 * 32;
    R.queries = ( uint * ) calloc ( roundUp * R.wordPerQuery, sizeof ( uint ) );
    R.readLengths = ( uint * ) calloc ( roundUp, sizeof ( uint ) );
    R.readIDs = ( uint * ) calloc ( roundUp, sizeof ( uint ) );
    R.upkdQualities = ( char * ) calloc ( roundUp * maxReadLength, 1 );
    R.upkdQueryNames = ( char * ) calloc ( roundUp * maxLenReadName, 1 );
    gzFile gz = gzopen ( fileName, "r" );

    if ( gz == NULL ) { fprintf ( stderr, "cannot open %s\n", fileName ); exit ( 1 ); }

    char * buf = ( char * ) malloc ( INPUT_BUFFER_SIZE );
    size_t bufferSize = gzread ( gz, buf, INPUT_BUFFER_SIZE );
    uint bufferIndex = 0;
    char queryChar = buf[bufferIndex++];
    R.isFastq = 0;
    R.numQueries = loadSingleReadsGz ( gz, buf, charMap, R.queries, R.readLengths, R.readIDs, R.upkdQualities, R.upkdQueryNames,
                                       maxReadLength, maxNumQueries, bufferSize, queryChar, bufferIndex, 0, R.wordPerQuery,
                                       DEFAULT_QUAL_CONST, R.isFastq, maxLenReadName );
    gzclose ( gz );
    free ( buf );
    return R;
}

// paired-end input: mates interleaved (read 2k from file 1, read 2k+1 from file 2), as SOAP3-DP.cu feeds loadPairReadsGz2
static Reads loadPairs ( const char * fileName, const char * fileName2, unsigned char * charMap, uint maxReadLength, int maxLenReadName, uint maxNumQueries )
{
    Reads R;
    R.maxReadLength = maxReadLength;
    uint wpq = 1;

    while ( wpq < maxReadLength ) { wpq *= 2; }

    R.wordPerQuery = wpq / CHAR_PER_WORD;
    size_t roundUp = ( ( size_t ) maxNumQueries + 31 ) / 32 * 32;
    R.queries = ( uint * ) calloc ( roundUp * R.wordPerQuery, sizeof ( uint ) );
    R.readLengths = ( uint * ) calloc ( roundUp, sizeof ( uint ) );
    R.readIDs = ( uint * ) calloc ( roundUp, sizeof ( uint ) );
    R.upkdQualities = ( char * ) calloc ( roundUp * maxReadLength, 1 );
    R.upkdQueryNames = ( char * ) calloc ( roundUp * maxLenReadName, 1 );
    gzFile gz = gzopen ( fileName, "r" ), gz2 = gzopen ( fileName2, "r" );

    if ( gz == NULL || gz2 == NULL ) { fprintf ( stderr, "cannot open %s / %s\n", fileName, fileName2 ); exit ( 1 ); }

    char * buf = ( char * ) malloc ( INPUT_BUFFER_SIZE ), * buf2 = ( char * ) malloc ( INPUT_BUFFER_SIZE );
    size_t bufferSize = gzread ( gz, buf, INPUT_BUFFER_SIZE ), bufferSize2 = gzread ( gz2, buf2, INPUT_BUFFER_SIZE );
    uint bufferIndex = 0, bufferIndex2 = 0;
    char queryChar = buf[bufferIndex++], queryChar2 = buf2[bufferIndex2++];
    R.isFastq = 0;
    R.numQueries = loadPairReadsGz2 ( gz, buf, gz2, buf2, charMap, R.queries, R.readLengths, R.readIDs, R.upkdQualities, R.upkdQueryNames,
                                      maxReadLength, maxNumQueries, bufferSize, queryChar, bufferIndex, bufferSize2, queryChar2, bufferIndex2,
                                      0, R.wordPerQuery, DEFAULT_QUAL_CONST, R.isFastq, maxLenReadName );
    gzclose ( gz ); gzclose ( gz2 );
    free ( buf ); free ( buf2 );
    return R;
}

static void initHspAux ( Soap3Index * index, IniParams & ini, InputOptions & opt, size_t roundUp )
{
    HSPAux * hspaux = index->sraIndex->hspaux;
    memset ( hspaux, 0, sizeof ( HSPAux ) );
    hspaux->dpMatchScore = ini.Ini_MatchScore;
    hspaux->dpMisMatchScore = ini.Ini_MismatchScore;
    hspaux->alignmentType = opt.alignmentType;
    hspaux->readType = opt.readType;
    hspaux->peMaxOutputPerRead = ini.Ini_PEMaxOutputPerPair;
    hspaux->ProceedDPForTooManyHits = ini.Ini_proceedDPForTooManyHits;
    hspaux->x0_array = ( int * ) malloc ( roundUp * sizeof ( int ) );
    hspaux->x1_array = ( int * ) malloc ( roundUp * sizeof ( int ) );
    hspaux->mismatch_array = ( int * ) malloc ( roundUp * sizeof ( int ) );
    hspaux->minMAPQ = ini.Ini_minMAPQ;
    hspaux->maxMAPQ = ini.Ini_maxMAPQ;
    hspaux->bwaLikeScore = ini.Ini_bwaLikeScore;
    hspaux->maxLenReadName = ini.Ini_maxReadNameLen;

    if ( hspaux->bwaLikeScore ) { bwase_initialize ( hspaux->g_log_n ); }

    hspaux->soap3AnsArray = malloc ( roundUp * sizeof ( void * ) );
    hspaux->sa_start = ( unsigned int * ) malloc ( roundUp * sizeof ( unsigned int ) );
    hspaux->occ_start = ( unsigned int * ) malloc ( roundUp * sizeof ( unsigned int ) );
    hspaux->sa_num = ( unsigned int * ) malloc ( roundUp * sizeof ( unsigned int ) );
    hspaux->occ_num = ( unsigned int * ) malloc ( roundUp * sizeof ( unsigned int ) );
    hspaux->isPrintMDNM = opt.isPrintMDNM;
    hspaux->readGroup = opt.readGroup;

    if ( hspaux->readGroup == NULL || strlen ( hspaux->readGroup ) == 0 ) { hspaux->readGroup = opt.queryFileName; }

    hspaux->sampleName = opt.sampleName;

    if ( hspaux->sampleName == NULL || strlen ( hspaux->sampleName ) == 0 ) { hspaux->sampleName = ( char * ) DEFAULT_SAMPLE_NAME; }

    hspaux->readGrpOption = opt.readGrpOption;
    hspaux->readsIDForSingleDP = ( BothUnalignedPairsArrays * ) constructBothUnalignedPairsArrays ( 1 );
    hspaux->allHits = ( AllHits * ) constructAllHits ();
}

// ---------------------------------------------------------------------------------------------
// mode "search": dump what the CPU twin finds, case by case, in its own emission order
static int modeSearch ( int argc, char ** argv )
{
    if ( argc < 8 ) { fprintf ( stderr, "search <ini> <index> <reads> <L> <mm> <out>\n" ); return 1; }

    IniParams ini;

    if ( ParseIniFile ( argv[2], ini ) != 0 ) { fprintf ( stderr, "bad ini\n" ); return 1; }

    char * indexName = argv[3];
    uint maxReadLength = atoi ( argv[5] );
    int mm = atoi ( argv[6] );
    Soap3Index * index = INDEXLoad ( &ini, indexName, 0 );
    InputOptions opt;
    memset ( &opt, 0, sizeof ( opt ) );
    opt.alignmentType = OUTPUT_ALL_VALID;
    opt.readType = SINGLE_READ;
    opt.queryFileName = argv[4];
    initHspAux ( index, ini, opt, 1024 );
    Reads R = loadReads ( argv[4], index->charMap, maxReadLength, ini.Ini_maxReadNameLen, 1 << 22 );
    fprintf ( stderr, "[harness] %u reads loaded\n", R.numQueries );
    SRASetting st;
    memset ( &st, 0, sizeof ( st ) );
    st.occ = OCCConstruct ();
    st.OutFileFormat = SRA_OUTPUT_FORMAT_SAM_API;
    st.MaxOutputPerRead = ini.Ini_MaxOutputPerRead;
    st.ReadStrand = QUERY_BOTH_STRAND;
    st.ErrorType = SRA_STEP_ERROR_TYPE_MISMATCH_ONLY;
    st.OutputType = SRA_REPORT_ALL;
    st.MaxError = mm;
    st.MaxNBMismatch = 0;
    SRAQueryResultCount rOut;
    memset ( &rOut, 0, sizeof ( rOut ) );
    SRAModel * models[MAX_READ_LENGTH + 1];
    memset ( models, 0, sizeof ( models ) );
    int numCases = GetNumCases ( mm );
    char dummyQuality[MAX_READ_LENGTH];
    memset ( dummyQuality, 1, sizeof ( dummyQuality ) );
    SRAQueryInfo qiP, qiN;
    SRAQueryInput qP, qN;
    memset ( &qiP, 0, sizeof ( qiP ) );
    memset ( &qiN, 0, sizeof ( qiN ) );
    qiP.ReadStrand = QUERY_POS_STRAND;
    qiP.ReadQuality = dummyQuality;
    qiN.ReadStrand = QUERY_NEG_STRAND;
    qiN.ReadQuality = dummyQuality;
    qP.QueryInfo = &qiP; qP.QuerySetting = &st; qP.AlgnmtIndex = index->sraIndex; qP.QueryOutput = &rOut;
    qN.QueryInfo = &qiN; qN.QuerySetting = &st; qN.AlgnmtIndex = index->sraIndex; qN.QueryOutput = &rOut;
    SAList * sa_list = SAListConstruct ();
    OCCList * occ_list = OCCListConstruct ();
    FILE * f = fopen ( argv[7], "w" );
    unsigned char fwd[MAX_READ_LENGTH], rev[MAX_READ_LENGTH];
    double t0 = setStartTime ();

    for ( uint r = 0; r < R.numQueries; r++ )
    {
        int L = R.readLengths[r];

        if ( models[L] == NULL ) { models[L] = SRAModelConstruct ( L, QUERY_POS_STRAND, &st, index->sraIndex, ini.Ini_HostAlignmentModel ); }

        retrieve2BWTQueryFormat ( R.queries, r, R.wordPerQuery, fwd );

        for ( int i = 0; i < L; i++ ) { rev[i] = soap3DnaComplement[fwd[L - i - 1]]; }

        qiP.ReadCode = fwd; qiP.ReadLength = L; qiP.ReadId = r + 1;
        qiN.ReadCode = rev; qiN.ReadLength = L; qiN.ReadId = r + 1;
        // per read reset, as collect_all_answers does (CPUfunctions.cpp:1246-1247)
        rOut.TotalOccurrences = 0;
        memset ( rOut.WithError, 0, sizeof ( rOut.WithError ) );
        SAListReset ( sa_list );
        OCCListReset ( occ_list );

        for ( int c = 0; c < numCases; c++ )
        {
            unsigned int sa0 = sa_list->curr_size, oc0 = occ_list->curr_size;
            ProcessReadDoubleStrand2 ( &qP, &qN, models[L], models[L], c, sa_list, occ_list );

            for ( unsigned int k = sa0; k < sa_list->curr_size; k++ )
            {
                fprintf ( f, "%u %d S %u %u %d %d\n", r, c, sa_list->sa[k].saIndexLeft, sa_list->sa[k].saIndexRight,
                          ( int ) sa_list->sa[k].strand, ( int ) sa_list->sa[k].mismatchCount );
            }

            for ( unsigned int k = oc0; k < occ_list->curr_size; k++ )
            {
                fprintf ( f, "%u %d O %u %u %d %d\n", r, c, ( unsigned int ) occ_list->occ[k].ambPosition, ( unsigned int ) occ_list->occ[k].ambPosition,
                          ( int ) occ_list->occ[k].strand, ( int ) occ_list->occ[k].mismatchCount );
            }
        }

        fprintf ( f, "%u T %u", r, rOut.TotalOccurrences );

        for ( int e = 0; e <= 4; e++ ) { fprintf ( f, " %u", rOut.WithError[e] ); }

        fprintf ( f, "\n" );
    }

    double dt = getElapsedTime ( t0 );
    fprintf ( stderr, "[harness] search mm=%d: %u reads in %.3f s (%.1f reads/s, 1 thread)\n", mm, R.numQueries, dt, R.numQueries / dt );
    fclose ( f );
    return 0;
}

// ---------------------------------------------------------------------------------------------
// mode "sam": reference host pipeline, single-end, with every answer resolved by the CPU twin
struct Ctx
{
    Soap3Index * index;
    IniParams ini;
    InputOptions opt;
    samfile_t * samOut;
    char * outName;
    // pairs with the DP stages on: where hostKernel's needProceedDP branch (CPUfunctions.cpp:2153-2262, 2444-2543) sends what it cannot pair
    ReadInputForDP * dpInput, * dpInputNew, * otherSoap3;
    BothUnalignedPairs * bothUnaligned;
    BothUnalignedPairsArrays * bothUnalignedArrays;
};

static void allValid ( Ctx & C, Reads & R, uint * queries, uint * readLengths, uint * seedLengths, uint * readIDs, uint numQueries, uint numMismatch,
                       uint * unAlignedReads, uint & numOfUnAligned, unsigned long long & numOfAnswer, uint & numOfAlignedRead,
                       uint8_t isTerminalCase )
{
    // mirrors all_valid_alignment (alignment.cu:534-1115) with one CPU thread and one batch
    uint numCases, sa1, sa2, wpa, wpa2;
    char skip2;
    getParametersForThisMismatch ( numMismatch, numCases, sa1, sa2, skip2, wpa, wpa2 );
    bool needOutputMAPQ = ( C.opt.alignmentType == OUTPUT_ALL_VALID || C.opt.alignmentType == OUTPUT_ALL_BEST ) &&
                          ( C.opt.outputFormat == SRA_OUTPUT_FORMAT_SAM_API );
    HostKernelArguements * hk = ( HostKernelArguements * ) calloc ( MAX_NUM_CPU_THREADS, sizeof ( HostKernelArguements ) );
    pthread_t threads[MAX_NUM_CPU_THREADS];
    IniParams ini1 = C.ini;
    ini1.Ini_NumOfCpuThreads = 1;
    setHostKernelArguments ( hk, threads, ini1, C.index, R.maxReadLength, R.wordPerQuery, wpa, &C.opt );
    static SRAModel * M[MAX_READ_LENGTH + 1], * M2[MAX_READ_LENGTH + 1], * Mn[MAX_READ_LENGTH + 1], * M2n[MAX_READ_LENGTH + 1];
    memset ( M, 0, sizeof ( M ) ); memset ( M2, 0, sizeof ( M2 ) ); memset ( Mn, 0, sizeof ( Mn ) ); memset ( M2n, 0, sizeof ( M2n ) );
    hk[0].upkdQualities = R.upkdQualities;
    hk[0].upkdQueryNames = R.upkdQueryNames;
    hk[0].SRAMismatchModel = M; hk[0].SRAMismatchModel2 = M2; hk[0].SRAMismatchModel_neg = Mn; hk[0].SRAMismatchModel2_neg = M2n;
    hk[0].outputFileName = C.outName;
    hk[0].readLengths = readLengths;
    hk[0].seedLengths = seedLengths;
    hk[0].readIDs = readIDs;
    hk[0].accumReadNum = 0;
    uint * unAlignedId = ( uint * ) malloc ( ( ( size_t ) numQueries + 32 ) * sizeof ( uint ) );
    hk[0].unAlignedIDs = unAlignedId;
    hk[0].unAlignedOcc = 0;
    hk[0].sraQuerySettings.MaxError = numMismatch;
    hk[0].sraQuerySettings.MaxNBMismatch = 0;
    hk[0].numCases = numCases;
    hk[0].numCases2 = GetNumCases ( numMismatch + 1 );
    hk[0].sa_range_allowed_1 = sa1;
    hk[0].sa_range_allowed_2 = sa2;
    hk[0].maxNumMismatch = numMismatch;
    hk[0].bothUnalignedPairs = ( ( BothUnalignedPairsArrays * ) C.index->sraIndex->hspaux->readsIDForSingleDP )->array[0];

    if ( C.dpInput )
    {
        hk[0].readInput = C.dpInput; hk[0].readInputForNewDefault = C.dpInputNew; hk[0].otherSoap3Result = C.otherSoap3;
    }

    if ( C.bothUnaligned ) { hk[0].bothUnalignedPairs = C.bothUnaligned; }

    hk[0].sraQuerySettings.SAMOutFilePtr = C.samOut;
    uint * lens = seedLengths ? seedLengths : readLengths;

    for ( uint r = 0; r < numQueries; r++ )
    {
        uint L = lens[r];

        if ( M[L] == NULL )
        {
            M[L] = SRAModelConstruct ( L, QUERY_POS_STRAND, &hk[0].sraQuerySettings, C.index->sraIndex, C.ini.Ini_HostAlignmentModel );
            Mn[L] = SRAModelConstruct ( L, QUERY_NEG_STRAND, &hk[0].sraQuerySettings, C.index->sraIndex, C.ini.Ini_HostAlignmentModel );
        }
    }

    if ( needOutputMAPQ && numMismatch < 4 )
    {
        SRASetting tmp;
        memcpy ( &tmp, &hk[0].sraQuerySettings, sizeof ( SRASetting ) );
        tmp.MaxError = numMismatch + 1;

        for ( uint r = 0; r < numQueries; r++ )
        {
            uint L = lens[r];

            if ( M2[L] == NULL )
            {
                M2[L] = SRAModelConstruct ( L, QUERY_POS_STRAND, &tmp, C.index->sraIndex, C.ini.Ini_HostAlignmentModel );
                M2n[L] = SRAModelConstruct ( L, QUERY_NEG_STRAND, &tmp, C.index->sraIndex, C.ini.Ini_HostAlignmentModel );
            }
        }
    }

    // answers: every (read, case) is "too many SA ranges => resolve on CPU" (0xFFFFFFFE), except
    // that a case with no hit must say 0xFFFFFFFD so that `nohit` stays honest (CPUfunctions.cpp:1750-1752)
    size_t roundUp = ( ( size_t ) numQueries + 31 ) / 32 * 32;
    uint * answers[MAX_NUM_CASES];
    {
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
        SAList * sl = SAListConstruct ();
        OCCList * ol = OCCListConstruct ();
        unsigned char fwd[MAX_READ_LENGTH], rev[MAX_READ_LENGTH];

        for ( uint c = 0; c < numCases; c++ )
        {
            answers[c] = ( uint * ) malloc ( roundUp * wpa * sizeof ( uint ) );
            memset ( answers[c], 0xFF, roundUp * wpa * sizeof ( uint ) );
        }

        for ( uint r = 0; r < numQueries; r++ )
        {
            int L = lens[r];
            retrieve2BWTQueryFormat ( queries, r, R.wordPerQuery, fwd );

            for ( int i = 0; i < L; i++ ) { rev[i] = soap3DnaComplement[fwd[L - i - 1]]; }

            qiP.ReadCode = fwd; qiP.ReadLength = L;
            qiN.ReadCode = rev; qiN.ReadLength = L;

            for ( uint c = 0; c < numCases; c++ )
            {
                rOut.TotalOccurrences = 0;
                SAListReset ( sl );
                OCCListReset ( ol );
                ProcessReadDoubleStrand2 ( &qP, &qN, M[L], Mn[L], c, sl, ol );
                bool hit = ( sl->curr_size > 0 || ol->curr_size > 0 );
                uint * ans = answers[c] + ( size_t ) r / 32 * 32 * wpa + r % 32;
                ans[0] = hit ? 0xFFFFFFFE : 0xFFFFFFFD;

                // unique-best decides on `isMoreThanSA1` (CPUfunctions.cpp:1262-1266), which the all-0xFFFFFFFE shortcut above
                // would make true for every aligned read.  For that mode a case with at most sa_range_allowed_1 ranges is handed
                // over the way the device does it: (l, r-l | mismatches << 24 | (strand-1) << 27) per range (CPUfunctions.cpp:1276-1295)
                if ( hit && C.opt.alignmentType == OUTPUT_UNIQUE_BEST && sl->curr_size + ol->curr_size <= sa1 )
                {
                    static std::vector<uint> invSa;
                    BWT * bwt = hk[0].sraIndex.bwt;

                    if ( invSa.empty () )
                    {
                        invSa.resize ( ( size_t ) bwt->textLength + 1 );

                        for ( unsigned long long i = 0; i <= bwt->textLength; i++ )
                        {
                            unsigned int v = ( *bwt->_bwtSaValue ) ( bwt, i );      // SA[0] is stored as -1

                            if ( v <= bwt->textLength ) { invSa[v] = ( uint ) i; }
                        }
                    }

                    uint k = 0;

                    for ( unsigned int x = 0; x < sl->curr_size; x++, k++ )
                    {
                        ans[ ( k * 2 ) * 32] = sl->sa[x].saIndexLeft;
                        ans[ ( k * 2 + 1 ) * 32] = ( sl->sa[x].saIndexRight - sl->sa[x].saIndexLeft ) | ( ( uint ) sl->sa[x].mismatchCount << BGS_GPU_ANSWER_OFFSET_LENGTH ) |
                                                   ( ( uint ) ( sl->sa[x].strand - 1 ) << ( BGS_GPU_ANSWER_OFFSET_LENGTH + 3 ) );
                    }

                    for ( unsigned int x = 0; x < ol->curr_size; x++, k++ )
                    {
                        ans[ ( k * 2 ) * 32] = invSa[ol->occ[x].ambPosition];
                        ans[ ( k * 2 + 1 ) * 32] = ( ( uint ) ol->occ[x].mismatchCount << BGS_GPU_ANSWER_OFFSET_LENGTH ) |
                                                   ( ( uint ) ( ol->occ[x].strand - 1 ) << ( BGS_GPU_ANSWER_OFFSET_LENGTH + 3 ) );
                    }
                }
            }
        }

        SAListFree ( sl );
        OCCListFree ( ol );
    }
    uint zeros[MAX_NUM_CASES];
    memset ( zeros, 0, sizeof ( zeros ) );
    hk[0].batchFirstReadId = 0;
    hk[0].skipFirst = 0;
    hk[0].numQueries = numQueries;
    hk[0].word_per_query = R.wordPerQuery;
    hk[0].queries = queries;
    hk[0].answers = answers;
    hk[0].alignedOcc = 0;
    hk[0].alignedReads = 0;
    hk[0].badReadIndices = NULL;
    hk[0].badAnswers = NULL;
    hk[0].badStartOffset = zeros;
    hk[0].badCountOffset = zeros;
    hk[0].outputGoodReads = TRUE;
    hk[0].skip_round_2 = 1;
    hk[0].isTerminalCase = isTerminalCase;
    pthread_t th;
    // REF_HARNESS_TIMING (bench.py's cpu_baseline, kind "reference"): wall clock inside the reference's own host kernel -- every (read, case) resolved by its
    // CPU path, occurrences decoded, records written -- summed over the passes of the ladder and reported by main() on stderr
    struct timespec hkT0, hkT1;
    clock_gettime ( CLOCK_MONOTONIC, &hkT0 );
    pthread_create ( &th, NULL, hostKernelThreadWrapper, ( void * ) &hk[0] );
    pthread_join ( th, NULL );
    clock_gettime ( CLOCK_MONOTONIC, &hkT1 );
    g_hostKernelSeconds += ( double ) ( hkT1.tv_sec - hkT0.tv_sec ) + 1e-9 * ( double ) ( hkT1.tv_nsec - hkT0.tv_nsec );
    g_hostKernelReads += numQueries;
    g_hostKernelPasses++;
    numOfAnswer = hk[0].alignedOcc;
    numOfAlignedRead = hk[0].alignedReads;
    numOfUnAligned = hk[0].unAlignedOcc;
    memcpy ( unAlignedReads, unAlignedId, ( size_t ) numOfUnAligned * sizeof ( uint ) );

    for ( uint c = 0; c < numCases; c++ ) { free ( answers[c] ); }

    for ( int i = 0; i <= MAX_READ_LENGTH; i++ )
    {
        if ( M[i] ) { SRAModelFree ( M[i] ); }

        if ( M2[i] ) { SRAModelFree ( M2[i] ); }

        if ( Mn[i] ) { SRAModelFree ( Mn[i] ); }

        if ( M2n[i] ) { SRAModelFree ( M2n[i] ); }
    }

    OCCFree ( hk[0].occ );
    free ( unAlignedId );
    free ( hk );
}

static int modeSam ( int argc, char ** argv )
{
    // argv: ref_harness sam <ini> single <index> <reads> [options]
    if ( argc < 6 ) { fprintf ( stderr, "sam <ini> single <index> <reads> [soap3-dp options]\n" ); return 1; }

    Ctx C;

    if ( ParseIniFile ( argv[2], C.ini ) != 0 ) { fprintf ( stderr, "bad ini\n" ); return 1; }

    memset ( &C.opt, 0, sizeof ( C.opt ) );
    argv[2] = argv[0];

    if ( !parseInputArgs ( argc - 2, argv + 2, C.opt ) ) { return 1; }

    const bool paired = C.opt.readType == PAIR_END_READ;

    if ( C.opt.enableDP == 1 ) { C.opt.numMismatch = C.ini.Ini_Soap3MisMatchAllow; }

    C.index = INDEXLoad ( &C.ini, C.opt.indexName, 0 );
    uint maxNumQueries = 1 << 22;
    size_t roundUp = ( ( size_t ) maxNumQueries + 31 ) / 32 * 32;
    initHspAux ( C.index, C.ini, C.opt, roundUp );
    HSP * hsp = C.index->sraIndex->hsp;
    HSPAux * hspaux = C.index->sraIndex->hspaux;
    Reads R = paired ? loadPairs ( C.opt.queryFileName, C.opt.queryFileName2, C.index->charMap, C.opt.maxReadLength, C.ini.Ini_maxReadNameLen, maxNumQueries )
              : loadReads ( C.opt.queryFileName, C.index->charMap, C.opt.maxReadLength, C.ini.Ini_maxReadNameLen, maxNumQueries );
    hspaux->isFastq = R.isFastq;
    fprintf ( stderr, "[harness] %u reads loaded, numMismatch=%d alignmentType=%d enableDP=%d outputFormat=%d\n", R.numQueries, C.opt.numMismatch,
              C.opt.alignmentType, ( int ) C.opt.enableDP, C.opt.outputFormat );
    C.outName = ( char * ) malloc ( strlen ( C.opt.outputPrefix ) + 16 );
    sprintf ( C.outName, "%s.gout.1", C.opt.outputPrefix );
    bam_header_t samHeader;
    SAMOutputHeaderConstruct ( &samHeader, hsp, hspaux, C.opt.maxReadLength );
    C.samOut = samopen ( C.outName, "wh", &samHeader );
    uint * seedLengths = NULL;

    if ( C.opt.maxReadLength > LONG_READ_LEN )
    {
        seedLengths = ( uint * ) malloc ( roundUp * sizeof ( uint ) );

        for ( uint i = 0; i < R.numQueries; i++ ) { seedLengths[i] = R.readLengths[i] > LONG_READ_LEN ? SOAP3_SEED_LEN : R.readLengths[i]; }
    }

    C.dpInput = C.dpInputNew = C.otherSoap3 = NULL;
    C.bothUnaligned = NULL;
    C.bothUnalignedArrays = NULL;
    std::vector<uint> origID ( R.readIDs, R.readIDs + R.numQueries );      // packUnPairedReads renumbers the reads between phases
    uint d1 = 0, d2 = 0;
    bool skipSoap3 = false;

    // SOAP3-DP.cu:620-686: what the detected read lengths switch
    if ( paired ) { d1 = GetReadLength ( R.readLengths, R.numQueries, 2 ); d2 = GetReadLength ( R.readLengths + 1, R.numQueries, 2 ); }
    else { d1 = GetReadLength ( R.readLengths, R.numQueries, 1 ); }

    if ( C.opt.enableDP == 1 && ( d1 < MIN_READ_LEN_FOR_DP || ( paired && d2 < MIN_READ_LEN_FOR_DP ) ) ) { C.opt.enableDP = 0; }
    else if ( C.opt.enableDP == 1 && paired && ( d1 > 150 || d2 > 150 ) ) { skipSoap3 = true; }

    if ( C.opt.enableDP == 0 && C.opt.numMismatch == -1 ) { C.opt.numMismatch = getDefaultMismatchNum ( d1 ); }

    if ( !paired && C.opt.enableDP == 1 )
    {
        C.bothUnalignedArrays = constructBothUnalignedPairsArrays ( 1 );
        C.bothUnaligned = C.bothUnalignedArrays->array[0];
    }

    if ( paired && C.opt.enableDP == 1 )
    {
        if ( C.opt.insert_low < ( int ) d2 ) { C.opt.insert_low = d2; }     // SOAP3-DP.cu:632-634

        if ( d1 <= 50 || d2 <= 50 ) { C.opt.numMismatch = 1; }

        C.opt.maxHitNum = getMaxHitNumForDefaultDP ( d1 );
        C.opt.maxHitNum2 = getMaxHitNumForDefaultDP ( d2 );
        C.dpInput = constructReadInputForDP ( 1 ); C.dpInputNew = constructReadInputForDP ( 1 ); C.otherSoap3 = constructReadInputForDP ( 1 );
        C.bothUnalignedArrays = constructBothUnalignedPairsArrays ( 2 );    // [0]: the host kernel's thread, [1]: what the default DP stages pass on
        C.bothUnaligned = C.bothUnalignedArrays->array[0];
        memset ( hspaux->x0_array, 0, roundUp * sizeof ( int ) ); memset ( hspaux->x1_array, 0, roundUp * sizeof ( int ) );
        memset ( hspaux->mismatch_array, 0, roundUp * sizeof ( int ) );
        memset ( hspaux->sa_num, 0, roundUp * sizeof ( unsigned int ) ); memset ( hspaux->occ_num, 0, roundUp * sizeof ( unsigned int ) );
        memset ( hspaux->soap3AnsArray, 0, roundUp * sizeof ( void * ) );
    }

    uint * unAligned = ( uint * ) malloc ( roundUp * sizeof ( uint ) );
    uint numUn = 0, numQueries = R.numQueries;
    unsigned long long totalAns = 0;
    uint totalAligned = 0;
    double t0 = setStartTime ();

    if ( skipSoap3 )
    {
        // alignment.cu:1929-1933: paired reads above 150 bases skip the BWT stage, every pair goes to the deep DP
        addAllFirstReadIDToBothUnalignedPairs ( C.bothUnaligned, R.numQueries );
    }
    else if ( paired )
    {
        // soap3_dp_pair_align with DP disabled (alignment.cu:1965-2013): all-best + SAM = all_best_alignment (phases 1, 2, k; :1236-1376),
        // unique / random best = four_phases_alignment (phases 0, 1, 2, k; :1119-1234), all valid = one pass at k
        uint nm = C.opt.numMismatch;
        std::vector<uint> phases;

        if ( C.opt.alignmentType == OUTPUT_ALL_BEST ) { phases.push_back ( 1 ); if ( nm > 1 ) { phases.push_back ( 2 ); } if ( nm > 2 ) { phases.push_back ( nm ); } }
        else if ( C.opt.alignmentType == OUTPUT_ALL_VALID ) { phases.push_back ( nm ); }
        else { phases.push_back ( 0 ); if ( nm > 0 ) { phases.push_back ( 1 ); } if ( nm > 1 ) { phases.push_back ( 2 ); } if ( nm > 2 ) { phases.push_back ( nm ); } }

        for ( size_t ph = 0; ph < phases.size (); ph++ )
        {
            unsigned long long a = 0;
            uint b = 0;
            bool last = ph + 1 == phases.size ();
            // the terminal flags of the reference: phase 1 is terminal when numMismatch <= 1 (all best), phase p when numMismatch == p
            allValid ( C, R, R.queries, R.readLengths, NULL, R.readIDs, numQueries, phases[ph], unAligned, numUn, a, b, last );
            totalAns += a;
            totalAligned += b;

            if ( last || b >= numQueries ) { break; }

            packUnPairedReads ( R.queries, R.readIDs, R.readLengths, unAligned, R.wordPerQuery, numUn, 1 << 20 );
            numQueries = numUn;
        }
    }
    else if ( C.opt.alignmentType == OUTPUT_ALL_BEST )
    {
        // all_best_single_alignment, alignment.cu:1422-1465
        uint nm = C.opt.numMismatch;

        for ( uint mm = 1; mm <= ( nm > 0 ? nm : 1 ); mm++ )
        {
            unsigned long long a = 0;
            uint b = 0;
            allValid ( C, R, R.queries, R.readLengths, seedLengths, R.readIDs, numQueries, mm, unAligned, numUn, a, b, mm >= nm );
            totalAns += a;
            totalAligned += b;

            if ( mm < nm )
            {
                packUnPairedReads ( R.queries, R.readIDs, R.readLengths, unAligned, R.wordPerQuery, numUn, 1 << 20 );

                if ( seedLengths )
                {
                    for ( uint i = 0; i < numUn; i++ ) { seedLengths[i] = R.readLengths[i] > LONG_READ_LEN ? SOAP3_SEED_LEN : R.readLengths[i]; }
                }

                numQueries = numUn;
            }
        }
    }
    else if ( C.opt.alignmentType == OUTPUT_ALL_VALID )
    {
        unsigned long long a = 0;
        uint b = 0;
        allValid ( C, R, R.queries, R.readLengths, seedLengths, R.readIDs, numQueries, C.opt.numMismatch, unAligned, numUn, a, b, 1 );
        totalAns += a;
        totalAligned += b;
    }
    else
    {
        // best_single_alignment, alignment.cu:1377-1420
        for ( uint mm = 0; mm <= ( uint ) C.opt.numMismatch; mm++ )
        {
            unsigned long long a = 0;
            uint b = 0;
            allValid ( C, R, R.queries, R.readLengths, seedLengths, R.readIDs, numQueries, mm, unAligned, numUn, a, b, mm == ( uint ) C.opt.numMismatch );
            totalAns += a;
            totalAligned += b;

            if ( mm < ( uint ) C.opt.numMismatch )
            {
                packUnPairedReads ( R.queries, R.readIDs, R.readLengths, unAligned, R.wordPerQuery, numUn, 1 << 20 );
                numQueries = numUn;
            }
        }
    }

    double dt = getElapsedTime ( t0 );
    samclose ( C.samOut );
    fprintf ( stderr, "[harness] aligned %u of %u reads, %llu alignments, %.3f s (%.1f reads/s, 1 thread)\n", totalAligned, R.numQueries, totalAns, dt,
              R.numQueries / dt );

    if ( C.dpInput )
    {
        // <prefix>.routing: one line per hit handed on, in the order of the lists (per read: SA ranges expanded, then occurrences):
        //   D|N|O <read> <strand> <mismatches> <position>      default DP / new default DP / kept for the unpaired output;   U <first read of the pair>
        // read = index in the input (the ladder packs the unpaired reads to the front between phases: readIDs[] maps back)
        std::string rn = std::string ( C.opt.outputPrefix ) + ".routing";
        FILE * f = fopen ( rn.c_str (), "w" );
        BWT * bwt = C.index->sraIndex->bwt;
        ReadInputForDP * lists[3] = { C.dpInput, C.dpInputNew, C.otherSoap3 };
        const char tag[3] = { 'D', 'N', 'O' };

        for ( int l = 0; l < 3; l++ )
        {
            ReadInputForDP * in = lists[l];
            std::vector<std::pair<uint, std::string> > rows;     // keyed by read so that SA ranges stand before the occurrences of the same read

            for ( unsigned int i = 0; i < in->saRangeTotalNum; i++ )
                for ( unsigned int k = in->sa_list[i].saIndexLeft; k <= in->sa_list[i].saIndexRight; k++ )
                {
                    char buf[128];
                    uint rd = R.readIDs[in->sa_list[i].readID] - 1;
                    sprintf ( buf, "%c %u %d %d %u", tag[l], rd, ( int ) in->sa_list[i].strand, ( int ) in->sa_list[i].mismatchCount, ( uint ) ( *bwt->_bwtSaValue ) ( bwt, k ) );
                    rows.push_back ( std::make_pair ( rd, std::string ( buf ) ) );
                }

            for ( unsigned int i = 0; i < in->occTotalNum; i++ )
            {
                char buf[128];
                uint rd = R.readIDs[in->occ_list[i].readID] - 1;
                sprintf ( buf, "%c %u %d %d %u", tag[l], rd, ( int ) in->occ_list[i].strand, ( int ) in->occ_list[i].mismatchCount, ( uint ) in->occ_list[i].ambPosition );
                rows.push_back ( std::make_pair ( rd, std::string ( buf ) ) );
            }

            std::stable_sort ( rows.begin (), rows.end (), [] ( const std::pair<uint, std::string> & a, const std::pair<uint, std::string> & b ) { return a.first < b.first; } );

            for ( size_t i = 0; i < rows.size (); i++ ) { fprintf ( f, "%s\n", rows[i].second.c_str () ); }
        }

        std::vector<uint> un;

        for ( unsigned int i = 0; i < C.bothUnaligned->totalNum; i++ ) { un.push_back ( R.readIDs[C.bothUnaligned->readIDs[i]] - 1 ); }

        std::sort ( un.begin (), un.end () );

        for ( size_t i = 0; i < un.size (); i++ ) { fprintf ( f, "U %u\n", un[i] ); }

        fclose ( f );
    }

    // -----------------------------------------------------------------------------------------------------------------------------
    // the DP stages behind the ladder: the sequencing of soap3_dp_pair_align (alignment.cu:2014-2446) and soap3_dp_single_align
    // (alignment.cu:2558-2690), calling the reference's own engines (REF_HARNESS_NO_DP=1 stops after the routing dump)
    if ( C.opt.enableDP == 1 && getenv ( "REF_HARNESS_NO_DP" ) == NULL )
    {
        std::string dpName = std::string ( C.opt.outputPrefix ) + ".dpout.1", unName = std::string ( C.opt.outputPrefix ) + ".unpair";
        samfile_t * samDP = samopen ( dpName.c_str (), "wh", &samHeader );
        InputOptions o = C.opt;

        if ( o.alignmentType == OUTPUT_ALL_BEST && o.outputFormat == SRA_OUTPUT_FORMAT_SAM_API ) { o.alignmentType = OUTPUT_ALL_VALID; }

        IniParams ini = C.ini;
        const char * th = getenv ( "REF_HARNESS_DP_THREADS" );
        ini.Ini_NumOfCpuThreads = th ? atoi ( th ) : 1;
        DPParameters dp;
        memset ( &dp, 0, sizeof ( dp ) );
        getParameterForAllDP ( dp, o, ini );
        unsigned int nRead = 0, nAlg = 0;

        if ( paired )
        {
            samfile_t * samUn = samopen ( unName.c_str (), "wh", &samHeader );
            ReadInputForDPArrays arrNew, arrDef;
            arrNew.inputArrays = &C.dpInputNew; arrNew.numArrays = 1;
            arrDef.inputArrays = &C.dpInput; arrDef.numArrays = 1;

            if ( !skipSoap3 )
            {
                getParameterForNewDefaultDP ( dp, o, ini, d1, d2 );
                refDpDumpNote ( "new default DP" );

                if ( C.dpInputNew->readNum > 0 )
                {
                    newSemiGlobalDP ( &arrNew, o.insert_high, o.insert_low, R.queries, R.readLengths, R.readIDs, R.upkdQueryNames, R.upkdQualities,
                                      C.opt.maxReadLength, C.ini.Ini_PEStrandLeftLeg, C.ini.Ini_PEStrandRightLeg, C.index, NULL, NULL, NULL, NULL,
                                      o.alignmentType, &dp, nRead, nAlg, 0, o.outputFormat, NULL, samDP, C.bothUnalignedArrays->array[1] );
                    fprintf ( stderr, "[harness] new default DP: %u pairs in, %u aligned\n", C.dpInputNew->readNum, nRead );
                }
            }

            getParameterForDefaultDP ( dp, o, ini, d1, d2 );
            refDpDumpNote ( "default DP" );

            if ( C.dpInput->readNum > 0 )
            {
                nRead = nAlg = 0;
                semiGlobalDP2 ( &arrDef, o.insert_high, o.insert_low, R.queries, R.readLengths, R.readIDs, R.upkdQueryNames, R.upkdQualities,
                                C.opt.maxReadLength, C.ini.Ini_PEStrandLeftLeg, C.ini.Ini_PEStrandRightLeg, C.index, NULL, NULL, NULL, NULL,
                                o.alignmentType, &dp, nRead, nAlg, 0, o.outputFormat, NULL, samDP, C.bothUnalignedArrays->array[1] );
                fprintf ( stderr, "[harness] default DP: %u pairs in, %u aligned\n", C.dpInput->readNum, nRead );
            }

            getParameterForDeepDP ( dp, o, ini, d1, d2 );
            refDpDumpNote ( "deep DP" );
            unsigned int toDeep = C.bothUnalignedArrays->array[0]->totalNum + C.bothUnalignedArrays->array[1]->totalNum;

            if ( toDeep > 0 )
            {
                nRead = nAlg = 0;
                DPForUnalignPairs2 ( C.bothUnalignedArrays, o.insert_high, o.insert_low, R.queries, R.readLengths, R.readIDs, R.upkdQueryNames, R.upkdQualities,
                                     C.opt.maxReadLength, C.ini.Ini_PEStrandLeftLeg, C.ini.Ini_PEStrandRightLeg, C.index, NULL, NULL, NULL, NULL,
                                     o.alignmentType, &dp, nRead, nAlg, 0, o.outputFormat, NULL, samDP );
                fprintf ( stderr, "[harness] deep DP: %u pairs in, %u aligned\n", toDeep, nRead );
            }

            // single-end DP of the ends of the pairs no stage could pair (alignment.cu:2368-2444)
            if ( o.outputFormat == SRA_OUTPUT_FORMAT_SAM_API )
            {
                o.readType = SINGLE_READ;
                getParameterForSingleDP ( dp, o, ini, d1 );
                hspaux->singleDPcutoffThreshold = dp.paramRead[0].cutoffThreshold;
                UnalignedSinglesArrays * us = ( UnalignedSinglesArrays * ) hspaux->readsIDForSingleDP;
                refDpDumpNote ( "single-end DP of leftover ends" );
                fprintf ( stderr, "[harness] single-end DP: %u ends in\n", us->array[0]->totalNum );

                if ( us->array[0]->totalNum > 0 )
                {
                    nRead = nAlg = 0;
                    DPForUnalignSingle2 ( us, R.queries, R.readLengths, R.readIDs, R.upkdQueryNames, R.upkdQualities, C.opt.maxReadLength, C.index,
                                          NULL, NULL, NULL, NULL, o.alignmentType, &dp, nRead, nAlg, 0, o.outputFormat, NULL, samUn );
                    sortReadPtrs ( ( AllHits * ) hspaux->allHits );
                    outputSingleResultForPairEnds ( ( AllHits * ) hspaux->allHits, R.queries, R.readLengths, R.readIDs, R.upkdQueryNames, R.upkdQualities,
                                                    C.opt.maxReadLength, 0, o.outputFormat, samUn, C.index );
                }
            }

            samclose ( samUn );
        }
        else
        {
            getParameterForSingleDP ( dp, o, ini, d1 );
            hspaux->singleDPcutoffThreshold = dp.paramRead[0].cutoffThreshold;
            refDpDumpNote ( "single-end DP" );
            fprintf ( stderr, "[harness] single-end DP: %u reads in\n", C.bothUnaligned->totalNum );

            if ( C.bothUnaligned->totalNum > 0 )
            {
                DPForUnalignSingle2 ( C.bothUnalignedArrays, R.queries, R.readLengths, R.readIDs, R.upkdQueryNames, R.upkdQualities, C.opt.maxReadLength, C.index,
                                      NULL, NULL, NULL, NULL, o.alignmentType, &dp, nRead, nAlg, 0, o.outputFormat, NULL, samDP );
                fprintf ( stderr, "[harness] single-end DP: %u aligned\n", nRead );
            }
        }

        samclose ( samDP );
    }

    return 0;
}

// ---------------------------------------------------------------------------------------------
// mode "dpparams": the DP stage parameters (getParameterFor*DP, CPUfunctions.cpp:46-260) and the seed tables (getSeedPositions,
// definitions.h:323-442) for every read length, as the reference computes them with the ini given
static int modeDpParams ( int argc, char ** argv )
{
    // argv: ref_harness dpparams <ini> <out>
    IniParams ini;

    if ( ParseIniFile ( argv[2], ini ) != 0 ) { fprintf ( stderr, "bad ini\n" ); return 1; }

    InputOptions opt;
    memset ( &opt, 0, sizeof ( opt ) );
    FILE * f = fopen ( argv[3], "w" );

    for ( uint L = 25; L <= 1024; L++ )
    {
        DPParameters d;
        memset ( &d, 0, sizeof ( d ) );
        getParameterForAllDP ( d, opt, ini );
        fprintf ( f, "L %u all %d %d %d %d\n", L, d.matchScore, d.mismatchScore, d.openGapScore, d.extendGapScore );
        memset ( &d, 0, sizeof ( d ) );
        getParameterForDefaultDP ( d, opt, ini, L, L );
        fprintf ( f, "L %u default cutoff %d clip %d %d maxhit %d\n", L, d.paramRead[0].cutoffThreshold, d.softClipLeft, d.softClipRight, d.paramRead[0].maxHitNum );
        memset ( &d, 0, sizeof ( d ) );
        getParameterForNewDefaultDP ( d, opt, ini, L, L );
        fprintf ( f, "L %u newdefault cutoff %d maxhit %d seedlen %d\n", L, d.paramRead[0].cutoffThreshold, d.paramRead[0].maxHitNum, d.paramRead[0].seedLength );
        memset ( &d, 0, sizeof ( d ) );
        getParameterForDeepDP ( d, opt, ini, L, L );
        fprintf ( f, "L %u deep cutoff %d maxhit %d seedlen %d sample %d tailtrim %d\n", L, d.paramRead[0].cutoffThreshold, d.paramRead[0].maxHitNum,
                  d.paramRead[0].seedLength, d.paramRead[0].sampleDist, d.tailTrimLen );
        memset ( &d, 0, sizeof ( d ) );
        getParameterForSingleDP ( d, opt, ini, L );
        fprintf ( f, "L %u single cutoff %d maxhit %d seedlen %d seednum %d\n", L, d.paramRead[0].cutoffThreshold, d.paramRead[0].maxHitNum, d.paramRead[0].seedLength,
                  d.singleDPSeedNum );
        fprintf ( f, "L %u defaultmaxhit %d\n", L, getMaxHitNumForDefaultDP ( L ) );

        for ( int stage = 1; stage <= 5; stage++ )
        {
            if ( stage == STAGE_DEFAULT_DP ) { continue; }

            int seedLength = 0, seedNum = 0, pos[256];
            getSeedPositions ( stage, ( int ) L, &seedLength, pos, &seedNum );
            fprintf ( f, "L %u seeds %d len %d n %d :", L, stage, seedLength, seedNum );

            for ( int i = 0; i < seedNum; i++ ) { fprintf ( f, " %d", pos[i] ); }

            fprintf ( f, "\n" );
        }
    }

    fclose ( f );
    return 0;
}

// ---------------------------------------------------------------------------------------------
// mode "dpsingle": SAM records of single-read DP results, written by the reference's own outputDPSingleResult2 (OutputDPResult.cpp:937)
// -> SingleDPOutputSAMAPI (BGS-IO.cpp:5857).  The DP results come from a text file (one per line, grouped by read, in the order the
// output stage receives them): readID strand algnmt score editdist num_sameScore cigar   (algnmt 4294967295 + cigar * = unaligned)
static int modeDpSingle ( int argc, char ** argv )
{
    // argv: ref_harness dpsingle <ini> <results.txt> single <index> <reads> [options]
    Ctx C;

    if ( ParseIniFile ( argv[2], C.ini ) != 0 ) { fprintf ( stderr, "bad ini\n" ); return 1; }

    const char * resultFile = argv[3];
    memset ( &C.opt, 0, sizeof ( C.opt ) );
    argv[3] = argv[0];

    if ( !parseInputArgs ( argc - 3, argv + 3, C.opt ) ) { return 1; }

    C.index = INDEXLoad ( &C.ini, C.opt.indexName, 0 );
    uint maxNumQueries = 1 << 20;
    size_t roundUp = ( ( size_t ) maxNumQueries + 31 ) / 32 * 32;
    initHspAux ( C.index, C.ini, C.opt, roundUp );
    HSP * hsp = C.index->sraIndex->hsp;
    HSPAux * hspaux = C.index->sraIndex->hspaux;
    Reads R = loadReads ( C.opt.queryFileName, C.index->charMap, C.opt.maxReadLength, C.ini.Ini_maxReadNameLen, maxNumQueries );
    hspaux->isFastq = R.isFastq;
    // the DP stage of soap3_dp_single_align (alignment.cu:2558-2585): parameters from the detected read length
    uint detected = GetReadLength ( R.readLengths, R.numQueries, 10 );
    DPParameters dp;
    memset ( &dp, 0, sizeof ( dp ) );
    getParameterForAllDP ( dp, C.opt, C.ini );
    getParameterForSingleDP ( dp, C.opt, C.ini, detected );
    hspaux->singleDPcutoffThreshold = dp.paramRead[0].cutoffThreshold;
    C.outName = ( char * ) malloc ( strlen ( C.opt.outputPrefix ) + 16 );
    sprintf ( C.outName, "%s.dpout.1", C.opt.outputPrefix );
    bam_header_t samHeader;
    SAMOutputHeaderConstruct ( &samHeader, hsp, hspaux, C.opt.maxReadLength );
    C.samOut = samopen ( C.outName, "wh", &samHeader );
    std::vector<SingleAlgnmtResult> res;
    FILE * f = fopen ( resultFile, "r" );

    if ( !f ) { fprintf ( stderr, "cannot open %s\n", resultFile ); return 1; }

    unsigned int readID, algnmt;
    int strand, score, editdist, same;
    char cigar[4096];

    while ( fscanf ( f, "%u %d %u %d %d %d %4095s", &readID, &strand, &algnmt, &score, &editdist, &same, cigar ) == 7 )
    {
        SingleAlgnmtResult r;
        r.readID = readID; r.strand = ( char ) strand; r.algnmt = algnmt; r.score = score; r.editdist = editdist; r.num_sameScore = same;
        r.cigarString = cigar[0] == '*' ? NULL : strdup ( cigar );
        res.push_back ( r );
    }

    fclose ( f );
    fprintf ( stderr, "[harness] %u reads, %zu DP results, detected read length %u, cutoff %d\n", R.numQueries, res.size (), detected, hspaux->singleDPcutoffThreshold );

    // one call per read, as DPSOutputThread does after OutputBuffer::ready (DV-DPfunctions.cu:1776-1830)
    for ( size_t i = 0; i < res.size (); )
    {
        size_t j = i;

        while ( j < res.size () && res[j].readID == res[i].readID ) { j++; }

        outputDPSingleResult2 ( &res[i], ( unsigned int ) ( j - i ), R.queries, R.readLengths, R.readIDs, R.upkdQueryNames, R.upkdQualities,
                                C.opt.maxReadLength, 0, SRA_OUTPUT_FORMAT_SAM_API, NULL, C.samOut, C.index );
        i = j;
    }

    samclose ( C.samOut );
    return 0;
}

// ---------------------------------------------------------------------------------------------
// mode "dppair": SAM records of the DP stages for pairs, written by the reference's own outputRead2 (OutputDPResult.cpp:217) ->
// pairDPOutputSAMAPI (BGS-IO.cpp:4504) and outputDeepDPResult2 (OutputDPResult.cpp:564) -> pairDeepDPOutputSAMAPI (BGS-IO.cpp:3824).
// Result vectors come from a text file, grouped by pair, in the order the output stage receives them:
//   D readID which pos1 strand1 score1 pos2 strand2 score2 editdist insert same cigar              (AlgnmtDPResult)
//   P readID pos1 strand1 score1 edit1 same1 cigar1 pos2 strand2 score2 edit2 same2 cigar2 insert  (DeepDPAlignResult)
// and what the BWT stage left in hspaux (x0_array / x1_array / mismatch_array) from a second file:  readID x0 x1 mismatch
static int modeDpPair ( int argc, char ** argv )
{
    // argv: ref_harness dppair <ini> <results.txt> <stats.txt> pair <index> <reads1> <reads2> -u <ins> [options]
    Ctx C;

    if ( ParseIniFile ( argv[2], C.ini ) != 0 ) { fprintf ( stderr, "bad ini\n" ); return 1; }

    const char * resultFile = argv[3];
    const char * statsFile = argv[4];
    memset ( &C.opt, 0, sizeof ( C.opt ) );
    argv[4] = argv[0];

    if ( !parseInputArgs ( argc - 4, argv + 4, C.opt ) ) { return 1; }

    C.index = INDEXLoad ( &C.ini, C.opt.indexName, 0 );
    uint maxNumQueries = 1 << 20;
    size_t roundUp = ( ( size_t ) maxNumQueries + 31 ) / 32 * 32;
    initHspAux ( C.index, C.ini, C.opt, roundUp );
    HSP * hsp = C.index->sraIndex->hsp;
    HSPAux * hspaux = C.index->sraIndex->hspaux;
    memset ( hspaux->x0_array, 0, roundUp * sizeof ( int ) );
    memset ( hspaux->x1_array, 0, roundUp * sizeof ( int ) );
    memset ( hspaux->mismatch_array, 0, roundUp * sizeof ( int ) );
    Reads R = loadPairs ( C.opt.queryFileName, C.opt.queryFileName2, C.index->charMap, C.opt.maxReadLength, C.ini.Ini_maxReadNameLen, maxNumQueries );
    hspaux->isFastq = R.isFastq;
    C.outName = ( char * ) malloc ( strlen ( C.opt.outputPrefix ) + 16 );
    sprintf ( C.outName, "%s.dpout.1", C.opt.outputPrefix );
    bam_header_t samHeader;
    SAMOutputHeaderConstruct ( &samHeader, hsp, hspaux, C.opt.maxReadLength );
    C.samOut = samopen ( C.outName, "wh", &samHeader );
    FILE * f = fopen ( statsFile, "r" );

    if ( !f ) { fprintf ( stderr, "cannot open %s\n", statsFile ); return 1; }

    unsigned int readID;
    int x0, x1, mm;

    while ( fscanf ( f, "%u %d %d %d", &readID, &x0, &x1, &mm ) == 4 )
    { hspaux->x0_array[readID] = x0; hspaux->x1_array[readID] = x1; hspaux->mismatch_array[readID] = mm; }

    fclose ( f );
    std::vector<AlgnmtDPResult> dres;
    std::vector<DeepDPAlignResult> pres;
    std::vector<SingleAlgnmtResult> ures;
    f = fopen ( resultFile, "r" );

    if ( !f ) { fprintf ( stderr, "cannot open %s\n", resultFile ); return 1; }

    char kind[8], cigar[4096], cigar2[4096];

    while ( fscanf ( f, "%7s", kind ) == 1 )
    {
        if ( kind[0] == 'U' )
        {
            SingleAlgnmtResult r;
            int s1, fromDP;
            memset ( &r, 0, sizeof ( r ) );

            if ( fscanf ( f, "%u %d %u %d %d %d %4095s", &r.readID, &fromDP, &r.algnmt, &s1, &r.score, &r.editdist, cigar ) != 7 ) { fprintf ( stderr, "bad U line\n" ); return 1; }

            r.strand = ( char ) s1; r.num_sameScore = fromDP;      // carried to the patch below
            r.cigarString = cigar[0] == '*' ? NULL : strdup ( cigar );
            ures.push_back ( r );
        }
        else if ( kind[0] == 'D' )
        {
            AlgnmtDPResult r;
            int which, s1, s2;
            memset ( &r, 0, sizeof ( r ) );

            if ( fscanf ( f, "%u %d %u %d %d %u %d %d %d %d %d %4095s", &r.readID, &which, &r.algnmt_1, &s1, &r.score_1, &r.algnmt_2, &s2, &r.score_2, &r.editdist,
                          &r.insertSize, &r.num_sameScore, cigar ) != 12 ) { fprintf ( stderr, "bad D line\n" ); return 1; }

            r.whichFromDP = ( char ) which; r.strand_1 = ( char ) s1; r.strand_2 = ( char ) s2;
            r.cigarString = cigar[0] == '*' ? NULL : strdup ( cigar );
            dres.push_back ( r );
        }
        else
        {
            DeepDPAlignResult r;
            int s1, s2;
            memset ( &r, 0, sizeof ( r ) );

            if ( fscanf ( f, "%u %u %d %d %d %d %4095s %u %d %d %d %d %4095s %d", &r.readID, &r.algnmt_1, &s1, &r.score_1, &r.editdist_1, &r.num_sameScore_1, cigar,
                          &r.algnmt_2, &s2, &r.score_2, &r.editdist_2, &r.num_sameScore_2, cigar2, &r.insertSize ) != 14 ) { fprintf ( stderr, "bad P line\n" ); return 1; }

            r.strand_1 = ( char ) s1; r.strand_2 = ( char ) s2;
            r.cigarString_1 = cigar[0] == '*' ? NULL : strdup ( cigar );
            r.cigarString_2 = cigar2[0] == '*' ? NULL : strdup ( cigar2 );
            pres.push_back ( r );
        }
    }

    fclose ( f );
    fprintf ( stderr, "[harness] %u reads, %zu default-DP and %zu deep-DP results\n", R.numQueries, dres.size (), pres.size () );

    // one call per pair, as DPOutputThread / DP2OutputThread do after OutputBuffer::ready (DV-DPfunctions.cu:2490-2505, 3840-3860)
    for ( size_t i = 0; i < dres.size (); )
    {
        size_t j = i;

        while ( j < dres.size () && dres[j].readID == dres[i].readID ) { j++; }

        outputRead2 ( &dres[i], ( unsigned int ) ( j - i ), R.queries, R.readLengths, R.readIDs, R.upkdQueryNames, R.upkdQualities,
                      C.opt.maxReadLength, 0, SRA_OUTPUT_FORMAT_SAM_API, NULL, C.samOut, C.index, C.ini.Ini_PEStrandLeftLeg, C.ini.Ini_PEStrandRightLeg );
        i = j;
    }

    for ( size_t i = 0; i < pres.size (); )
    {
        size_t j = i;

        while ( j < pres.size () && pres[j].readID == pres[i].readID ) { j++; }

        outputDeepDPResult2 ( &pres[i], ( unsigned int ) ( j - i ), R.queries, R.readLengths, R.readIDs, R.upkdQueryNames, R.upkdQualities,
                              C.opt.maxReadLength, 0, SRA_OUTPUT_FORMAT_SAM_API, NULL, C.samOut, C.index, C.ini.Ini_PEStrandLeftLeg, C.ini.Ini_PEStrandRightLeg );
        i = j;
    }

    // U lines: the hits each end of a leftover pair has on its own (fromDP 1: single-end DP, 0: carried over from the BWT stage), through the
    // reference's AllHits store and outputSingleResultForPairEnds (OutputDPResult.cpp:1062) -> unproperlypairDPOutputSAMAPI (BGS-IO.cpp:2932)
    if ( !ures.empty () )
    {
        uint detected = GetReadLength ( R.readLengths, R.numQueries, 10 );
        DPParameters dp;
        memset ( &dp, 0, sizeof ( dp ) );
        getParameterForAllDP ( dp, C.opt, C.ini );
        getParameterForSingleDP ( dp, C.opt, C.ini, detected );
        hspaux->singleDPcutoffThreshold = dp.paramRead[0].cutoffThreshold;
        AllHits * allHits = ( AllHits * ) hspaux->allHits;
        resetAllHits ( allHits );

        for ( size_t i = 0; i < ures.size (); )
        {
            size_t j = i;

            while ( j < ures.size () && ures[j].readID == ures[i].readID ) { j++; }

            unsigned int before = allHits->hitNum;
            inputAlgnmtsToArray ( allHits, &ures[i], ( int ) ( j - i ) );
            unsigned int at = before;

            for ( size_t q = i; q < j; q++ )
            {
                if ( ures[q].algnmt == 0xFFFFFFFF ) { continue; }

                allHits->hitArray[at].isFromDP = ures[q].num_sameScore;
                allHits->hitArray[at].num_sameScore = ures[q].num_sameScore ? 1 : 0;
                at++;
            }

            i = j;
        }

        sortReadPtrs ( allHits );
        outputSingleResultForPairEnds ( allHits, R.queries, R.readLengths, R.readIDs, R.upkdQueryNames, R.upkdQualities, C.opt.maxReadLength, 0, SRA_OUTPUT_FORMAT_SAM_API,
                                        C.samOut, C.index );
    }

    samclose ( C.samOut );
    return 0;
}

int main ( int argc, char ** argv )
{
    if ( argc < 2 ) { fprintf ( stderr, "usage: ref_harness models|search|sam ...\n" ); return 1; }

    if ( strcmp ( argv[1], "models" ) == 0 ) { dumpModels ( argv[2] ); return 0; }

    if ( strcmp ( argv[1], "dpparams" ) == 0 ) { return modeDpParams ( argc, argv ); }

    if ( strcmp ( argv[1], "dpsingle" ) == 0 ) { return modeDpSingle ( argc, argv ); }

    if ( strcmp ( argv[1], "dppair" ) == 0 ) { return modeDpPair ( argc, argv ); }

    if ( strcmp ( argv[1], "search" ) == 0 ) { return modeSearch ( argc, argv ); }

    if ( strcmp ( argv[1], "sam" ) == 0 )
    {
        const int rc = modeSam ( argc, argv );

        if ( getenv ( "REF_HARNESS_TIMING" ) )
        {
            fprintf ( stderr, "[ref_harness] hostKernel: %.6f seconds in %llu passes over %llu reads in all (one CPU thread; the harness's own pre-pass that tells "
                              "hostKernel which cases have hits is not in it)\n", g_hostKernelSeconds, g_hostKernelPasses, g_hostKernelReads );
        }

        return rc;
    }

    fprintf ( stderr, "unknown mode %s\n", argv[1] );
    return 1;
}
