// s3dp_dpstage.cpp -- host-side parameters of the DP stages, part of the C-ABI (include/s3dp.h):
//   s3_dp_stage_parameters  <->  getParameterFor{Default,NewDefault,Deep,Single}DP + getMaxHitNumForDefaultDP (CPUfunctions.cpp:36-260)
//   s3_seed_positions       <->  getSeedPositions (definitions.h:323-442)
// Plain integer rules; checked line by line against the reference's own functions (tests/golden/dpparams.txt.gz).
#include <math.h>
#include <string.h>
#include "../../include/s3dp.h"

extern "C" int s3_dp_stage_parameters(int stage, uint32_t L, const s3_dp_ini *ini, s3_dp_stage_params *o)
{
    if (!ini || !o || stage < S3_STAGE_SINGLE_DP || stage > S3_STAGE_DEEP_DP_ROUND2) return S3_EINVAL;
    memset(o, 0, sizeof *o);
    // DPScoreThreshold=DEFAULT: 0.3 x read length, rounded up (DP_SCORE_THRESHOLD_RATIO, definitions.h:152)
    o->cutoffThreshold = ini->dpScoreThreshold < 0 ? (int)ceil(0.3 * (double)L) : ini->dpScoreThreshold;
    o->softClipLeft = ini->softClipLeft;
    o->softClipRight = ini->softClipRight;
    switch (stage) {
    case S3_STAGE_DEFAULT_DP:
        o->maxHitNum = L > 50 ? 50 : 70;
        break;
    case S3_STAGE_NEW_DEFAULT_DP:
        o->maxHitNum = L > 50 ? 150 : 200;
        o->seedLength = L > 75 ? 26 : L > 50 ? 24 : 22;
        break;
    case S3_STAGE_DEEP_DP_ROUND1:
    case S3_STAGE_DEEP_DP_ROUND2:
        // round 2 (the pairs that lost a seed to round 1's hit limit) runs on round 1's parameter object with ONE field replaced: the hit limit of both
        // ends becomes MAX_SEED_HITS_DEEP_DP_FOR_NORMAL_READ_2 = 1000 whatever the read length (DV-DPForBothUnalign.cu:137-139, definitions.h:179); its
        // seeds come from getSeedPositions(STAGE_DEEP_DP_ROUND2) (s3_seed_positions).  Until round 5 this function refused the stage and the command line
        // ran round 2 on a zeroed structure -- hit limit 0: every seed with a hit dropped -- which no fixture noticed because none held a seed with more than
        // a hundred hits; the scale test with a 600-copy family does (tests/test_gpu_cli_vs_oracle_scale.py::...[pair_every_route]: 326 pairs).
        o->maxHitNum = stage == S3_STAGE_DEEP_DP_ROUND2 ? 1000 : (L > 50 ? 100 : 150);
        o->seedLength = L > 150 ? 45 : L > 80 ? 26 : L > 60 ? 24 : L > 40 ? 22 : 20;
        o->sampleDist = (int)(o->seedLength * 0.5);
        o->tailTrimLen = 0;
        break;
    case S3_STAGE_SINGLE_DP:
        o->maxHitNum = L > 300 ? 4 : L > 80 ? 10 : L > 60 ? 20 : L > 40 ? 30 : 40;
        o->seedLength = L > 300 ? 70 : L > 80 ? 38 : L > 60 ? 32 : L > 40 ? 26 : 22;
        o->singleDPSeedNum = L > 100 ? 3 + (int)(L / 100) : 3;
        break;
    }
    return S3_OK;
}

extern "C" int s3_seed_positions(int stage, int L, int *seedLength, int *pos, int *seedNum)
{
    if (!seedLength || !pos || !seedNum || L < 1) return S3_EINVAL;
    if (stage == S3_STAGE_SINGLE_DP || stage == S3_STAGE_NEW_DEFAULT_DP) {
        *seedLength = L > 300 ? 70 : L > 80 ? 38 : L > 60 ? 32 : L > 40 ? 26 : 22;
        *seedNum = L > 120 ? 3 + L / 100 : 3;
        const int H = L > 300 ? (int)(L * 0.15) : 0;
        const int X = L > 300 ? (int)(L * 0.15) : L > 80 ? 10 : L > 60 ? 4 : L > 40 ? 4 : 0;
        const int apart = (L - X - H) / *seedNum;
        for (int i = 0; i < *seedNum; i++) pos[i] = H + i * apart;
        if (pos[*seedNum - 1] > L - *seedLength - X) pos[*seedNum - 1] = L - *seedLength - X;
        return S3_OK;
    }
    if (stage == S3_STAGE_DEEP_DP_ROUND1 || stage == S3_STAGE_DEEP_DP_ROUND2) {
        const bool r2 = stage == S3_STAGE_DEEP_DP_ROUND2;
        *seedLength = L > 150 ? (r2 ? 52 : 45) : L > 80 ? (r2 ? 30 : 26) : L > 60 ? (r2 ? 28 : 24) : L > 40 ? (r2 ? 26 : 22) : (r2 ? 24 : 20);
        int H = 0, T = 0;
        if (L > 150) { H = (int)(L * 0.1); T = (int)(L * 0.2); }
        int n = 0;
        for (int i = L - *seedLength - T; i >= H; i -= *seedLength / 2) pos[n++] = i;
        // (the reference reads seedPositions[-1] when no seed fits; lengths that short never reach the DP stages: MIN_READ_LEN_FOR_DP = 30)
        if (n == 0 || pos[n - 1] > H) pos[n++] = H;
        *seedNum = n;
        return S3_OK;
    }
    return S3_EINVAL;
}
