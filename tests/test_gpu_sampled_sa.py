"""-m gpu: indexes built with a sampled suffix array (the reference's SaValueFreq 2 / 4, README.md:55-65; reader 2bwt-lib/BWT.c:254-255, decode
BWTSaValue :1694-1724).  The library expands the samples to the full array on the device at upload; everything behind it must not notice."""
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

import s3dp
from conftest import GOLDEN, ROOT, golden_reads, host_index
from test_gpu_cli_sam import sam_records

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "soap3-dp_amd", "soap3-dp")


@pytest.mark.parametrize("interval", [2, 4, 7])
def test_sampled_suffix_array_gives_the_same_alignments(interval):
    import copy
    hi = host_index("g1")
    reads = golden_reads("g1")[:1024]
    q, lens, wpq = s3dp.pack_reads(reads, 100)

    def run(h):
        ix = s3dp.Index(h)
        b = s3dp.Batch(ix, len(reads), wpq)
        b.upload(q, lens)
        b.align_single_allbest(4, 1000)
        res = b.fetch_alignments()
        b.close()
        ix.close()
        return res

    full = run(hi)
    hs = copy.copy(hi)
    hs.sa = np.ascontiguousarray(hi.sa[::interval]).copy()
    hs.sa_interval = interval
    assert len(hs.sa) == (hi.n + interval) // interval
    for x, y in zip(full, run(hs)):
        assert (x == y).all()


def test_cli_with_the_reference_builders_sampled_index(tmp_path):
    """tests/golden/g1/index_sa4.sa: the reference builder's own .sa for the g1 genome with SaValueFreq=4"""
    for f in os.listdir(os.path.join(GOLDEN, "g1")):
        if f.startswith("index.") and f != "index.sa":
            os.symlink(os.path.join(GOLDEN, "g1", f), tmp_path / f)
    shutil.copy(os.path.join(GOLDEN, "g1", "index_sa4.sa"), tmp_path / "index.sa")
    shutil.copy(os.path.join(GOLDEN, "g1", "reads.fq"), tmp_path / "g1.fq")
    out = subprocess.run([BIN, "single", str(tmp_path / "index"), "g1.fq", "-L", "100", "-s", "4", "-b", "2", "-o", "out"], cwd=tmp_path, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    mine_h, mine = sam_records(open(tmp_path / "out.gout.1").read())
    ref_h, ref = sam_records(gzip.open(os.path.join(GOLDEN, "g1", "sam_s4.sam.gz"), "rt").read())
    assert mine_h == ref_h and mine == ref
