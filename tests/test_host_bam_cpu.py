"""`soap3-dp -b 3`: the records the command line writes as SAM, written as BAM by soap3-dp_amd/host/s3_bam.cpp, must come back from the REFERENCE's
samtools (oracle/_ref/ref_bam2sam, built from /root/reference/samtools-0.1.18 by oracle/ref.mk) as the same SAM text -- header and every record.
The inputs are the golden SAM files the reference itself wrote (single-end, long reads, pairs, DP stages on)."""
import gzip
import os
import subprocess
import pytest
from conftest import GOLDEN, ROOT

TOOL = os.path.join(ROOT, "tests", "tools", "sam2bam")
REF = os.path.join(ROOT, "oracle", "_ref", "ref_bam2sam")
HOST = os.path.join(ROOT, "soap3-dp_amd", "host")


def tool():
    src = [os.path.join(ROOT, "tests", "tools", "sam2bam.cpp"), os.path.join(HOST, "s3_bam.cpp")]
    if not os.path.exists(TOOL) or any(os.path.getmtime(s) > os.path.getmtime(TOOL) for s in src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + HOST, "-o", TOOL] + src + ["-lz"])
    return TOOL


def golden_sams():
    out = []
    for d in sorted(os.listdir(GOLDEN)):
        p = os.path.join(GOLDEN, d)
        if os.path.isdir(p):
            for f in sorted(os.listdir(p)):
                if f.endswith(".sam.gz"):
                    out.append(os.path.join(d, f))
    return out


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference's samtools is not built (oracle/ref.mk needs /root/reference)")
@pytest.mark.parametrize("rel", golden_sams())
@pytest.mark.parametrize("piece", [300, 100000])
def test_bam_round_trip_through_reference_samtools(tmp_path, rel, piece):
    text = gzip.open(os.path.join(GOLDEN, rel), "rt").read()
    if not text.startswith("@"):
        pytest.skip("no header in this fixture")
    sam = str(tmp_path / "in.sam")
    with open(sam, "w") as f:
        f.write(text)
    bam = str(tmp_path / "out.bam")
    subprocess.check_call([tool(), sam, bam, str(piece)])
    back = subprocess.run([REF, bam], capture_output=True, text=True, timeout=300)
    assert back.returncode == 0, back.stderr[-1000:]
    a, b = text.splitlines(), back.stdout.splitlines()
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x == y
    assert len(text) < 10000 or piece < 100000 or os.path.getsize(bam) < len(text)          # (it is compressed)
