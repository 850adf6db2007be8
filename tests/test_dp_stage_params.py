"""not gpu: the DP stage parameters and seed tables of the product (host-side rules in libs3dp.so, no device needed) against the reference's
own getParameterFor*DP / getSeedPositions for every read length 25..1024 (tests/golden/dpparams.txt.gz, dumped by oracle/ref_harness.cpp)."""
import ctypes as C
import gzip
import os

import s3dp
from conftest import GOLDEN


class Ini(C.Structure):
    _fields_ = [("dpScoreThreshold", C.c_int), ("softClipLeft", C.c_int), ("softClipRight", C.c_int)]


class Params(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("cutoffThreshold", "maxHitNum", "seedLength", "sampleDist", "tailTrimLen", "softClipLeft", "softClipRight",
                                       "singleDPSeedNum")]


def test_stage_parameters_and_seed_positions_match_reference():
    lib = s3dp.lib()
    ini = Ini(-1, 49, 49)                                   # DPScoreThreshold=DEFAULT, MaxFrontLenClipped / MaxEndLenClipped of host/soap3-dp.ini
    lines = gzip.open(os.path.join(GOLDEN, "dpparams.txt.gz"), "rt").read().split("\n")
    checked = 0
    for l in lines:
        f = l.split()
        if not f:
            continue
        L = int(f[1])
        kind = f[2]
        p = Params()
        if kind == "default":
            assert lib.s3_dp_stage_parameters(2, L, C.byref(ini), C.byref(p)) == 0
            assert [p.cutoffThreshold, p.softClipLeft, p.softClipRight, p.maxHitNum] == [int(f[4]), int(f[6]), int(f[7]), int(f[9])], l
        elif kind == "newdefault":
            assert lib.s3_dp_stage_parameters(3, L, C.byref(ini), C.byref(p)) == 0
            assert [p.cutoffThreshold, p.maxHitNum, p.seedLength] == [int(f[4]), int(f[6]), int(f[8])], l
        elif kind == "deep":
            assert lib.s3_dp_stage_parameters(4, L, C.byref(ini), C.byref(p)) == 0
            assert [p.cutoffThreshold, p.maxHitNum, p.seedLength, p.sampleDist, p.tailTrimLen] == [int(f[4]), int(f[6]), int(f[8]), int(f[10]), int(f[12])], l
            # round 2 of the deep DP runs on round 1's parameter object with the hit limit of both ends replaced by
            # MAX_SEED_HITS_DEEP_DP_FOR_NORMAL_READ_2 = 1000 (DV-DPForBothUnalign.cu:137-139, definitions.h:179) -- whatever the read length
            p2 = Params()
            assert lib.s3_dp_stage_parameters(5, L, C.byref(ini), C.byref(p2)) == 0
            assert [p2.cutoffThreshold, p2.maxHitNum, p2.seedLength, p2.sampleDist, p2.tailTrimLen, p2.softClipLeft, p2.softClipRight] == \
                   [p.cutoffThreshold, 1000, p.seedLength, p.sampleDist, p.tailTrimLen, p.softClipLeft, p.softClipRight], l
        elif kind == "single":
            assert lib.s3_dp_stage_parameters(1, L, C.byref(ini), C.byref(p)) == 0
            assert [p.cutoffThreshold, p.maxHitNum, p.seedLength, p.singleDPSeedNum] == [int(f[4]), int(f[6]), int(f[8]), int(f[10])], l
        elif kind == "defaultmaxhit":
            assert lib.s3_dp_stage_parameters(2, L, C.byref(ini), C.byref(p)) == 0 and p.maxHitNum == int(f[3]), l
        elif kind == "seeds":
            stage = int(f[3])
            want_len, want_n = int(f[5]), int(f[7])
            want = [int(x) for x in f[9:]]
            sl, n = C.c_int(), C.c_int()
            pos = (C.c_int * 128)()
            assert lib.s3_seed_positions(stage, L, C.byref(sl), pos, C.byref(n)) == 0
            assert (sl.value, n.value, list(pos[:n.value])) == (want_len, want_n, want), l
        else:
            continue
        checked += 1
    assert checked > 8000
