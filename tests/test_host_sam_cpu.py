"""not gpu: the product's host code (index + read loaders, SAM writer with the reference's MAPQ / XA / boundary rules) checked against the
reference's golden SAM WITHOUT a device: tests/tools/oracle_sam.cpp drives it with the CPU oracle's alignments.  This pins, on the CPU,
both the oracle's ladders (single-end -h 1..4, long-read mode, paired-end) and the host formatting, record for record."""
import gzip
import os
import shutil
import subprocess

import pytest

import s3o
from conftest import GOLDEN, ROOT
from test_gpu_cli_sam import C3_VARIANTS, LONG_VARIANTS, PAIR_VARIANTS, VARIANTS, sam_pair_records, sam_records

TOOL = os.path.join(ROOT, "tests", "tools", "oracle_sam")
_built = {}


def tool():
    if "ok" not in _built:
        s3o.build()
        src = [os.path.join(ROOT, "tests", "tools", "oracle_sam.cpp"), os.path.join(ROOT, "soap3-dp_amd", "host", "s3_host.cpp"), os.path.join(ROOT, "soap3-dp_amd", "host", "s3_bam.cpp")]
        osrc = [os.path.join(ROOT, "oracle", f) for f in ("s3_oracle.c", "s3_oracle_pe.c")]
        objs = []
        for c in osrc:
            o = os.path.join("/tmp", "s3o_" + os.path.basename(c) + ".o")
            subprocess.check_call(["gcc", "-O2", "-c", c, "-o", o])
            objs.append(o)
        lib = os.path.join(ROOT, "soap3-dp_amd")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", TOOL] + src + objs + ["-L" + lib, "-ls3dp", "-lz", "-lm", "-Wl,-rpath," + lib])
        _built["ok"] = True
    return TOOL


def run(tmp_path, args, env=None):
    out = subprocess.run([tool()] + args + ["-o", "out"], cwd=tmp_path, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    return open(tmp_path / "out.gout.1").read()


def compare(mine, ref):
    assert mine[0] == ref[0]
    assert set(mine[1]) == set(ref[1])
    bad = [(k, ref[1][k], mine[1][k]) for k in ref[1] if ref[1][k] != mine[1][k]]
    assert not bad, "%d differing records, first:\n%s\n%s" % (len(bad), bad[0][1], bad[0][2])


@pytest.mark.parametrize("name,maxlen", [("g1", 100), ("g2", 120)])
@pytest.mark.parametrize("variant,opts", VARIANTS)
def test_single_end_sam(tmp_path, name, maxlen, variant, opts):
    shutil.copy(os.path.join(GOLDEN, name, "reads.fq"), tmp_path / (name + ".fq"))
    text = run(tmp_path, ["single", os.path.join(GOLDEN, name, "index"), name + ".fq", "-L", str(maxlen)] + opts)
    compare(sam_records(text), sam_records(gzip.open(os.path.join(GOLDEN, name, "sam_%s.sam.gz" % variant), "rt").read()))


def test_index_files_read_in_pieces_by_all_host_threads(tmp_path):
    """the loader reads an index file's payload at its final place with one pread stream per host thread (64 MiB and more each); here in pieces of 4 KiB on
    eight threads, so that the fixture's small files are split too: same records as the golden SAM"""
    name, maxlen, (variant, opts) = "g2", 120, VARIANTS[0]
    shutil.copy(os.path.join(GOLDEN, name, "reads.fq"), tmp_path / (name + ".fq"))
    env = dict(os.environ, S3DP_LOAD_PIECE="4096", S3DP_HOST_THREADS="8")
    text = run(tmp_path, ["single", os.path.join(GOLDEN, name, "index"), name + ".fq", "-L", str(maxlen)] + opts, env=env)
    compare(sam_records(text), sam_records(gzip.open(os.path.join(GOLDEN, name, "sam_%s.sam.gz" % variant), "rt").read()))


@pytest.mark.parametrize("variant,opts", LONG_VARIANTS)
def test_long_read_sam(tmp_path, variant, opts):
    name = "g3"
    shutil.copy(os.path.join(GOLDEN, name, "reads.fq"), tmp_path / (name + ".fq"))
    text = run(tmp_path, ["single", os.path.join(GOLDEN, name, "index"), name + ".fq", "-L", "200"] + opts)
    compare(sam_records(text), sam_records(gzip.open(os.path.join(GOLDEN, name, "sam_%s.sam.gz" % variant), "rt").read()))


@pytest.mark.parametrize("variant,opts", PAIR_VARIANTS)
def test_pair_sam(tmp_path, variant, opts):
    name = "g4"
    for k in ("1", "2"):
        shutil.copy(os.path.join(GOLDEN, name, "reads_%s.fq" % k), tmp_path / ("%s_%s.fq" % (name, k)))
    text = run(tmp_path, ["pair", os.path.join(GOLDEN, name, "index"), name + "_1.fq", name + "_2.fq", "-L", "100", "-u", "500", "-v", "200"] + opts)
    compare(sam_pair_records(text), sam_pair_records(gzip.open(os.path.join(GOLDEN, name, "pe_%s.sam.gz" % variant), "rt").read()))


@pytest.mark.parametrize("variant,opts", C3_VARIANTS)
def test_pair_sam_c3_shape(tmp_path, variant, opts):
    """fixture g7: BASELINE config C3's shape -- 2 x 150 bp, -u 500 -v 1 (wordPerQuery 16), DP stages off"""
    name = "g7"
    for k in ("1", "2"):
        shutil.copy(os.path.join(GOLDEN, name, "reads_%s.fq" % k), tmp_path / ("%s_%s.fq" % (name, k)))
    text = run(tmp_path, ["pair", os.path.join(GOLDEN, name, "index"), name + "_1.fq", name + "_2.fq", "-L", "150", "-u", "500", "-v", "1"] + opts)
    compare(sam_pair_records(text), sam_pair_records(gzip.open(os.path.join(GOLDEN, name, "pe_%s.sam.gz" % variant), "rt").read()))
