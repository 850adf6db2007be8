"""gpu: the flank tables of check-and-extend (csrc/s3dp_flank.hip) against a suffix sort -- flankBack[k] = {SA[k], the 48 characters before SA[k], nearest first},
flankFwd[k'] = {E, the 48 characters from E on} in the order of the REVERSE index, which the upload finds by walking LF steps of the reverse index along the text
from anchors.  The expected tables come from numpy: the forward suffix array is the index file's, the reverse index's order from sorting the reversed text."""
import os

import numpy as np
import pytest

import s3dp
from conftest import host_index

pytestmark = pytest.mark.gpu


def suffix_array(t):
    """suffix array of t (uint8 codes 0..3) with the empty suffix first, by prefix doubling"""
    n = len(t)
    rank = np.concatenate([t.astype(np.int64) + 1, [0]])
    sa = np.argsort(rank, kind="stable")
    h = 1
    while True:
        r2 = np.concatenate([rank[h:], np.zeros(h, dtype=np.int64)]) if h <= n else np.zeros(n + 1, dtype=np.int64)
        key = rank * (n + 2) + r2
        sa = np.argsort(key, kind="stable")
        ks = key[sa]
        newrank = np.zeros(n + 1, dtype=np.int64)
        newrank[sa] = np.concatenate([[0], np.cumsum(ks[1:] != ks[:-1])])
        rank = newrank
        if rank.max() == n:
            return sa
        h *= 2


def windows(t, starts, step):
    """[len(starts), 3] uint32: 48 characters from starts in direction step (+1 / -1), 16 per word MSB first; outside the text: 0"""
    n = len(t)
    out = np.zeros((len(starts), 3), dtype=np.uint32)
    for j in range(48):
        p = starts + step * j
        ok = (p >= 0) & (p < n)
        c = np.where(ok, t[np.clip(p, 0, n - 1)], 0).astype(np.uint32)
        out[:, j // 16] |= c << np.uint32(30 - 2 * (j % 16))
    return out


@pytest.mark.parametrize("name", ["g1", "g6"])
def test_flank_tables_equal_a_suffix_sort(name):
    hi = host_index(name)
    ix = s3dp.Index(hi)
    assert ix.flank_tables() == 2
    t = np.asarray(hi.codes, dtype=np.uint8)
    n = len(t)
    assert hi.sa_interval == 1 and n == hi.n
    sa = np.asarray(hi.sa, dtype=np.int64)[: n + 1]
    back = ix.fetch_flank(0, 0, n + 1)
    assert (back[:, 0].astype(np.int64) == sa).all()
    assert (back[1:, 1:] == windows(t, sa[1:] - 1, -1)).all() and (back[0, 1:] == 0).all()
    # the reverse index orders the occurrences by what lies to their LEFT, read leftwards: entry k' <-> the suffix of the reversed text at rsa[k'],
    # which ends (in the text) at n - 1 - rsa[k']; E = the position behind it
    rsa = suffix_array(t[::-1])
    E = n - rsa
    fwd = ix.fetch_flank(1, 0, n + 1)
    assert (fwd[1:, 0].astype(np.int64) == E[1:]).all()
    assert (fwd[1:, 1:] == windows(t, E[1:], +1)).all()
    assert (fwd[0] == 0).all()                                                   # the empty suffix: never a candidate
    ix.close()


def test_the_switch_leaves_tables_out(monkeypatch):
    hi = host_index("g1")
    for want in (0, 1):
        monkeypatch.setenv("S3DP_FLANK", str(want))
        ix = s3dp.Index(hi)
        assert ix.flank_tables() == want
        ix.close()
    monkeypatch.setenv("S3DP_FLANK", "2")
    monkeypatch.setenv("S3DP_FLANK_RESERVE_GB", "100000")                         # more than any device has: no room for a table
    ix = s3dp.Index(hi)
    assert ix.flank_tables() == 0
    ix.close()


def test_fmv_words_counted_on_the_gpu_equal_numpy():
    """s3dp.fmv_words (the CPU 2BWT's occurrence arrays, written for the reference's own host code: bench.py's DP parity leg) counts on the GPU for genome-scale
    texts; the same words as the numpy path (which tests/test_builder.py holds byte-identical to the reference builder's files)"""
    rng = np.random.default_rng(4)
    n = 200_000_123                                              # (several pieces of 2^18 intervals)
    words = rng.integers(0, 1 << 32, size=(n + 15) // 16, dtype=np.uint64).astype(np.uint32)
    words[-1] &= np.uint32(0xFFFFFFFF) << np.uint32(2 * (16 - n % 16))      # the trailing bits of the last word are clear, as in a .bwt file
    cum = [0, 10, 20, 30, n]
    a = s3dp.fmv_words(words, n, 12345, cum, use_gpu=False)
    b = s3dp.fmv_words(words, n, 12345, cum, use_gpu=True)
    assert a.dtype == b.dtype and (a == b).all() and len(a) > 5 + n // 128


def test_the_tables_give_way_when_the_device_is_full():
    """An allocation of a batch that finds the device full releases the flank tables one at a time and tries again (a 10 M-read batch of a satellite-heavy genome
    outgrew the 69 GB beside the 219 GB index: round 6).  S3DP_TEST_OOM_EVERY makes every n-th allocation pretend; the search goes on with fewer tables and the
    same records.  (Own process: the switch is read once.)"""
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "tests")); sys.path.insert(0, os.path.join(%r, "soap3-dp_amd", "py"))
import os

import numpy as np
import s3dp, s3o
from conftest import host_index, golden_reads
hi = host_index("g1")
ix = s3dp.Index(hi)
assert ix.flank_tables() == 2
reads = golden_reads("g1")[:800]
q, lens, wpq = s3dp.pack_reads(reads, 100)
oix = s3o.Index(host_index=hi)
want = oix.align_single_allbest(q, lens, wpq, 3, 1000)
seen = set()
for rnd in range(4):
    b = s3dp.Batch(ix, len(reads), wpq)
    b.upload(q, lens)
    b.align_single_allbest(3, 1000)
    got = b.fetch_alignments()
    assert (got[0] == want[0]).all() and (got[2] == want[2]).all() and all((got[k] == want[k]).all() for k in (3, 4, 5))
    b.close()
    seen.add(ix.flank_tables())
assert ix.flank_tables() == 0 and seen >= {0}, seen
print("OK", sorted(seen))
''' % (ROOT, ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, S3DP_TEST_OOM_EVERY="7"))
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    assert "is released" in out.stderr
