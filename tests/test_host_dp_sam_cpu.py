"""not gpu: SAM records of DP results (single reads: dp1; pairs: dp2).  tests/golden/dp1 holds reads, DP result vectors (synthetic, tools/simulate_dp_results.py) and
the records the reference's own outputDPSingleResult2 / SingleDPOutputSAMAPI wrote for them (oracle/ref_harness.cpp, mode dpsingle); the
product's writer (soap3-dp_amd/host/s3_host_dp.cpp) must reproduce every record: CIGAR conversion, boundary trimming, MD / NM, X0 / X1 /
XM / XO / XG, XA:Z, MAPQ."""
import gzip
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT
from test_gpu_cli_sam import sam_records

TOOL = os.path.join(ROOT, "tests", "tools", "dp_sam")


def tool():
    src = [os.path.join(ROOT, "tests", "tools", "dp_sam.cpp"), os.path.join(ROOT, "soap3-dp_amd", "host", "s3_host.cpp"), os.path.join(ROOT, "soap3-dp_amd", "host", "s3_bam.cpp"),
           os.path.join(ROOT, "soap3-dp_amd", "host", "s3_host_dp.cpp")]
    if not os.path.exists(TOOL) or os.path.getmtime(TOOL) < max(os.path.getmtime(s) for s in src):
        lib = os.path.join(ROOT, "soap3-dp_amd")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", TOOL] + src + ["-L" + lib, "-ls3dp", "-lz", "-lm", "-Wl,-rpath," + lib])
    return TOOL


@pytest.mark.parametrize("variant,opts", [("h2", []), ("h1", ["-h", "1"]), ("h2_p", ["-p"]), ("h1_p", ["-h", "1", "-p"])])
def test_single_read_dp_records(tmp_path, variant, opts):
    d = os.path.join(GOLDEN, "dp1")
    shutil.copy(os.path.join(d, "reads.fq"), tmp_path / "dp1.fq")
    out = subprocess.run([tool(), os.path.join(d, "index"), "dp1.fq", os.path.join(d, "results.txt"), "out", "-L", "100"] + opts, cwd=tmp_path,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    mine = sam_records(open(tmp_path / "out.dpout.1").read())
    ref = sam_records(gzip.open(os.path.join(d, "dpout_%s.sam.gz" % variant), "rt").read())
    assert mine[0] == ref[0]
    assert set(mine[1]) == set(ref[1])
    bad = [(k, ref[1][k], mine[1][k]) for k in ref[1] if ref[1][k] != mine[1][k]]
    assert not bad, "%d differing records, first:\n%s\n%s" % (len(bad), bad[0][1], bad[0][2])


PAIR_TOOL = os.path.join(ROOT, "tests", "tools", "dp_pair_sam")


def pair_tool():
    host = os.path.join(ROOT, "soap3-dp_amd", "host")
    src = [os.path.join(ROOT, "tests", "tools", "dp_pair_sam.cpp")] + [os.path.join(host, f) for f in ("s3_host.cpp", "s3_bam.cpp", "s3_host_dp.cpp", "s3_host_dp_pair.cpp")]
    if not os.path.exists(PAIR_TOOL) or os.path.getmtime(PAIR_TOOL) < max(os.path.getmtime(s) for s in src + [os.path.join(host, "s3_host.h")]):
        lib = os.path.join(ROOT, "soap3-dp_amd")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", PAIR_TOOL] + src + ["-L" + lib, "-ls3dp", "-lz", "-lm", "-Wl,-rpath," + lib])
    return PAIR_TOOL


@pytest.mark.parametrize("variant,opts", [("h2", []), ("h1", ["-h", "1"]), ("h2p", ["-p"]), ("h1p", ["-h", "1", "-p"]), ("h3", ["-h", "3"]),
                                          ("h4p", ["-h", "4", "-p"]), ("nb2", ["-p", "-B", "0"]), ("nb1", ["-h", "1", "-B", "0"])])
def test_paired_dp_records(tmp_path, variant, opts):
    """tests/golden/dp2: result vectors of the default DP (one end from the BWT stage) and of the deep DP (both ends from the DP) for 480
    pairs -- equal / weaker extra results, both ends through the DP, boundary crossings, read-through pairs, BWT-stage statistics -- and
    the records the reference's outputRead2 / pairDPOutputSAMAPI and outputDeepDPResult2 / pairDeepDPOutputSAMAPI wrote for them
    (oracle/ref_harness.cpp, mode dppair); 240 more pairs no stage could pair, each end with hits of the single-end DP or of the BWT stage or
    none, through the reference's AllHits store and outputSingleResultForPairEnds / unproperlypairDPOutputSAMAPI.  nb*: the ini with BWALikeScore=0, i.e. the table-driven MAPQ rules."""
    d = os.path.join(GOLDEN, "dp2")
    shutil.copy(os.path.join(d, "reads_1.fq"), tmp_path / "dp2_1.fq")
    shutil.copy(os.path.join(d, "reads_2.fq"), tmp_path / "dp2_2.fq")
    out = subprocess.run([pair_tool(), os.path.join(GOLDEN, "dp1", "index"), "dp2_1.fq", "dp2_2.fq", os.path.join(d, "results.txt"), os.path.join(d, "stats.txt"), "out",
                          "-L", "100"] + opts, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    mine = [l for l in open(tmp_path / "out.dpout.1").read().split("\n") if l and not l.startswith("@")]
    ref = [l for l in gzip.open(os.path.join(d, "dpout_%s.sam.gz" % variant), "rt").read().split("\n") if l and not l.startswith("@")]
    assert len(mine) == len(ref) == 1440
    bad = [(a, b) for a, b in zip(ref, mine) if a != b]
    assert not bad, "%d differing records, first:\n%s\n%s" % (len(bad), bad[0][0], bad[0][1])


def test_the_survivor_among_deep_dp_results_with_equal_first_positions_is_pinned():
    """a pair's deep-DP results are sorted with std::sort by the first end's position alone and the first of every position is kept, as the
    reference's OutputBuffer::arrayCopyNRemoveDuplicate does -- with more than sixteen results the survivor among equal positions is what libstdc++'s
    introsort leaves in front.  The golden list was written by this toolchain's libstdc++ (the one the reference harness is built with): a library whose
    sort permutes differently must show up here, not as a silent change of X0 / X1 / XA in SAM files."""
    d = os.path.join(GOLDEN, "dp2")
    inp = open(os.path.join(d, "deep_dedup_input.txt")).read()
    want = open(os.path.join(d, "deep_dedup_survivors.txt")).read()
    out = subprocess.run([pair_tool(), "--dedup"], input=inp, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0
    assert out.stdout == want
    rows = [tuple(int(x) for x in l.split()) for l in inp.split("\n") if l]
    got = [tuple(int(x) for x in l.split()) for l in out.stdout.split("\n") if l]
    assert [p for p, _ in got] == sorted({p for p, _ in rows})                  # one result per position, ascending
    first = {}
    for p, t in rows:
        first.setdefault(p, t)
    assert any(first[p] != t for p, t in got)                                   # the case does leave the insertion-sort branch: a stable sort would fail it
