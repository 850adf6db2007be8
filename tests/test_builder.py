"""Our own index builder (soap3-dp_amd/py/s3dp/builder.py) must write, for the same text, exactly what the
reference tools (soap3-dp-builder + BGS-Build) wrote into the golden index files."""
import os

import numpy as np
import pytest

import s3dp
from s3dp import builder
from conftest import GOLDEN, host_index


@pytest.mark.parametrize("name", ["g1", "g2"])
def test_builder_matches_reference_index_files(name):
    ref = host_index(name)
    mine = builder.build_host_index(ref.codes, device="cpu")
    assert mine.n == ref.n
    for half in ("fwd", "rev"):
        a, b = getattr(mine, half), getattr(ref, half)
        assert a["inverseSa0"] == b["inverseSa0"]
        assert a["cum"] == b["cum"]
        assert (a["bwt"] == b["bwt"]).all()
        assert (a["occ"] == b["occ"]).all()
    assert (mine.sa == ref.sa).all()
    assert (mine.lkt == ref.lkt).all() and (mine.rev_lkt == ref.rev_lkt).all()
    assert (mine.pac == ref.pac).all()


def test_suffix_array_with_long_repeats_and_small_buckets():
    import torch
    rng = np.random.default_rng(5)
    unit = rng.integers(0, 4, size=97, dtype=np.uint8)
    codes = np.concatenate([unit] * 9 + [rng.integers(0, 4, size=500, dtype=np.uint8), unit, unit, np.zeros(70, np.uint8)])
    sa = builder.suffix_array(torch.from_numpy(codes), max_bucket=64).numpy()
    s = codes.tobytes()
    want = sorted(range(len(codes)), key=lambda p: s[p:] + b"")   # bytes compare: shorter prefix first == '$' smallest
    assert list(sa) == want


def test_builder_chunked_paths(monkeypatch):
    """same result when every scan / nonzero / gather is forced through the multi-chunk code path (used above 2^30 elements)"""
    monkeypatch.setattr(builder, "CH", 1 << 12)
    ref = host_index("g2")
    mine = builder.build_host_index(ref.codes, device="cpu")
    for half in ("fwd", "rev"):
        a, b = getattr(mine, half), getattr(ref, half)
        assert a["inverseSa0"] == b["inverseSa0"] and a["cum"] == b["cum"]
        assert (a["bwt"] == b["bwt"]).all() and (a["occ"] == b["occ"]).all()
    assert (mine.sa == ref.sa).all()
    assert (mine.lkt == ref.lkt).all() and (mine.rev_lkt == ref.rev_lkt).all()
    assert (mine.pac == ref.pac).all()


REF_BUILDER = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "soap3-dp-builder")


@pytest.mark.skipif(not os.path.exists(REF_BUILDER), reason="oracle/_ref/soap3-dp-builder is not built (make -f oracle/ref.mk where /root/reference exists)")
@pytest.mark.parametrize("name", ["g1", "g2", "g7"])
def test_fmv_files_equal_the_reference_builders(tmp_path, name):
    """<prefix>.fmv / .rev.fmv -- the CPU 2BWT's occurrence arrays, which the product never reads but the reference's own host code needs to load an index
    (bench.py's DP parity leg runs oracle/_ref/ref_harness on the index files of the benchmark genome): byte-identical to what soap3-dp-builder writes."""
    import shutil
    import subprocess
    refdir = os.path.dirname(REF_BUILDER)
    shutil.copy(os.path.join(GOLDEN, name, "genome.fa"), tmp_path / "g.fa")
    shutil.copy(os.path.join(refdir, "soap3-dp-builder.ini"), tmp_path / "soap3-dp-builder.ini")
    subprocess.check_call([REF_BUILDER, "g.fa"], cwd=tmp_path, stdout=subprocess.DEVNULL)
    hi = host_index(name)
    hi.write_fmv_files(str(tmp_path / "mine"))
    for mid in ("", ".rev"):
        want = open(tmp_path / ("g.fa.index" + mid + ".fmv"), "rb").read()
        got = open(tmp_path / ("mine" + mid + ".fmv"), "rb").read()
        assert len(got) == len(want) and got == want
