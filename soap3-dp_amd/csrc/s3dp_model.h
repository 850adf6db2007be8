// s3dp_model.h -- search-model tables shared by host (builder) and device (interpreter).
//
// A "model" is the reference's declarative description of the pigeonhole search for one read
// length and one mismatch budget: cases x steps (2bwt-flex/SRA2BWTMdl.h:127-157, built by
// SRAMismatchCasesPopulate, SRA2BWTMdl.c:1171, and post-processed at :4520-4602).  Here it is a
// flat POD table that the HIP kernel interprets; it is built on the host so that the float
// cell-boundary arithmetic is done exactly once, by the host compiler, never by device fma.
#pragma once
#include <stdint.h>

#define S3_MAX_STEPS 5
#define S3_MAX_CASES 10

enum : int8_t { S3_T_BACK_LOOKUP = 2, S3_T_FWD_LOOKUP = 4, S3_T_BACK_ONLY = 5, S3_T_BIDIR = 6 };

struct S3Step {
    int16_t start, end;        // read positions, inclusive; start > end means a backward step
    int16_t ceStart, ceEnd;    // check-and-extend window on the read position
    int16_t lm0;               // leftMostAligned at the first position of the step (forward: constant; backward: lm0 - i)
    int16_t len;
    int8_t  dir;               // +1 / -1
    int8_t  type;
    int8_t  minErr, maxErr;
    int8_t  ceThr;
    int8_t  pad[3];
};
static_assert(sizeof(S3Step) == 20, "S3Step layout");

struct S3Case {
    int32_t nSteps;
    int32_t id;
    S3Step  st[S3_MAX_STEPS];
};

struct S3Model {               // one read length, one mismatch budget
    int32_t nCases;
    int32_t readLength;
    S3Case  cs[S3_MAX_CASES];
};

// number of cases for a mismatch budget (definitions.h:116-120)
static inline int s3_num_cases(int maxError) { return maxError == 0 ? 1 : maxError == 1 ? 2 : maxError == 2 ? 4 : maxError == 3 ? 6 : 10; }

void s3_model_build(int readLength, int maxError, S3Model *out);
int  s3_model_text(int readLength, int maxError, char *buf, unsigned long bufLen);
