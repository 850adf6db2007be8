"""-m gpu: the drop-in `soap3-dp` command line (host C++ above the C-ABI) must write, read for read, the SAM records the
reference's CPU path wrote (tests/golden/*/sam_s*.sam.gz, produced by oracle/_ref/ref_harness with the reference's own
hostKernel + OCCOutputSAMAPI): position, strand, CIGAR, MAPQ, X0/X1/XM and the XA:Z list."""
import gzip
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "soap3-dp_amd", "soap3-dp")


def sam_records(text):
    hdr, rec = [], {}
    for l in text.split("\n"):
        if not l:
            continue
        if l.startswith("@"):
            if not l.startswith("@PG"):
                hdr.append(l)
        else:
            rec[l.split("\t", 1)[0]] = l
    return hdr, rec


# (fixture, extra options): -h 2 all best (default), -h 1 all valid, -h 3 unique best, -h 4 random best (+ -p: NM/MD tags)
VARIANTS = [("s2", ["-s", "2"]), ("s4", ["-s", "4"]), ("h1_s2", ["-s", "2", "-h", "1"]), ("h1_s0", ["-s", "0", "-h", "1"]),
            ("h3_s3", ["-s", "3", "-h", "3"]), ("h4_s3", ["-s", "3", "-h", "4", "-p"])]


@pytest.mark.parametrize("name,maxlen", [("g1", 100), ("g2", 120)])
@pytest.mark.parametrize("variant,opts", VARIANTS)
def test_cli_sam_matches_reference_cpu_path(tmp_path, name, maxlen, variant, opts):
    assert os.path.exists(BIN), "soap3-dp binary not built"
    fq = tmp_path / (name + ".fq")                      # the read-group id is the query file name as typed
    shutil.copy(os.path.join(GOLDEN, name, "reads.fq"), fq)
    out = subprocess.run([BIN, "single", os.path.join(GOLDEN, name, "index"), name + ".fq", "-L", str(maxlen)] + opts + ["-b", "2",
                          "-o", "out"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    mine_h, mine = sam_records(open(tmp_path / "out.gout.1").read())
    ref_h, ref = sam_records(gzip.open(os.path.join(GOLDEN, name, "sam_%s.sam.gz" % variant), "rt").read())
    assert mine_h == ref_h
    assert set(mine) == set(ref)
    bad = [(k, ref[k], mine[k]) for k in ref if ref[k] != mine[k]]
    assert not bad, "first differing record:\n%s\n%s" % (bad[0][1], bad[0][2])
    assert os.path.exists(tmp_path / "out.done")


# long-read mode: -L 200, reads of 121..200 bases
LONG_VARIANTS = [("s3", ["-s", "3"]), ("s1", ["-s", "1", "-p"]), ("h1_s2", ["-s", "2", "-h", "1"]), ("h3_s2", ["-s", "2", "-h", "3"]),
                 ("h4_s3", ["-s", "3", "-h", "4"])]


@pytest.mark.parametrize("variant,opts", LONG_VARIANTS)
def test_cli_sam_long_read_mode(tmp_path, variant, opts):
    name = "g3"
    fq = tmp_path / (name + ".fq")
    shutil.copy(os.path.join(GOLDEN, name, "reads.fq"), fq)
    out = subprocess.run([BIN, "single", os.path.join(GOLDEN, name, "index"), name + ".fq", "-L", "200"] + opts + ["-b", "2", "-o", "out"],
                         cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    mine_h, mine = sam_records(open(tmp_path / "out.gout.1").read())
    ref_h, ref = sam_records(gzip.open(os.path.join(GOLDEN, name, "sam_%s.sam.gz" % variant), "rt").read())
    assert mine_h == ref_h
    assert set(mine) == set(ref)
    bad = [(k, ref[k], mine[k]) for k in ref if ref[k] != mine[k]]
    assert not bad, "%d differing records, first:\n%s\n%s" % (len(bad), bad[0][1], bad[0][2])


# paired-end, DP stages off: pair <index> <r1> <r2> -u 500 -v 200 -s k
PAIR_VARIANTS = [("s3", ["-s", "3"]), ("s1", ["-s", "1", "-p"]), ("s4", ["-s", "4"]), ("h1_s2", ["-s", "2", "-h", "1"]),
                 ("h3_s3", ["-s", "3", "-h", "3"]), ("h4_s2", ["-s", "2", "-h", "4"])]


def sam_pair_records(text):
    hdr, rec = [], {}
    for l in text.split("\n"):
        if not l:
            continue
        if l.startswith("@"):
            if not l.startswith("@PG"):
                hdr.append(l)
        else:
            f = l.split("\t", 2)
            rec[(f[0], int(f[1]) & 0xC0)] = l             # (name, first / second in pair)
    return hdr, rec


@pytest.mark.parametrize("variant,opts", PAIR_VARIANTS)
def test_cli_sam_pairs(tmp_path, variant, opts):
    name = "g4"
    for k in ("1", "2"):
        shutil.copy(os.path.join(GOLDEN, name, "reads_%s.fq" % k), tmp_path / ("%s_%s.fq" % (name, k)))
    out = subprocess.run([BIN, "pair", os.path.join(GOLDEN, name, "index"), name + "_1.fq", name + "_2.fq", "-L", "100", "-u", "500", "-v", "200"] + opts +
                         ["-b", "2", "-o", "out"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    mine_h, mine = sam_pair_records(open(tmp_path / "out.gout.1").read())
    ref_h, ref = sam_pair_records(gzip.open(os.path.join(GOLDEN, name, "pe_%s.sam.gz" % variant), "rt").read())
    assert mine_h == ref_h
    assert set(mine) == set(ref)
    bad = [(k, ref[k], mine[k]) for k in ref if ref[k] != mine[k]]
    assert not bad, "%d differing records, first:\n%s\n%s" % (len(bad), bad[0][1], bad[0][2])
