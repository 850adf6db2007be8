"""-m gpu: edge cases of the search / ladder path against the oracle: empty and one-read batches, the shortest and the longest reads the
model tables cover, reads on the first and last bases of the text, homopolymer reads in a genome with long homopolymer runs (hit-limit
truncation), reads with N."""
import numpy as np
import pytest

import s3dp
import s3o
from s3dp import builder

pytestmark = pytest.mark.gpu

COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


@pytest.fixture(scope="module")
def world():
    rng = np.random.default_rng(2718)
    n = 70001
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    codes[5000:9000] = 0                                   # a 4 kbp poly-A run: far more than MaxOutputPerRead hits for an all-A read
    codes[20000:20600] = np.tile(np.array([0, 1], dtype=np.uint8), 300)   # a dinucleotide repeat
    hi = builder.build_host_index(codes, device="cpu")
    return rng, codes, hi, s3dp.Index(hi), s3o.Index(host_index=hi)


def seq(codes):
    return bytes(b"ACGT"[c] for c in codes)


def compare(ix, oix, reads, maxlen, mm, max_out=1000, report_type=2):
    q, lens, wpq = s3dp.pack_reads(reads, maxlen)
    b = s3dp.Batch(ix, max(1, len(reads)), wpq)
    b.upload(q, lens)
    b.align_single(mm, max_out, report_type)
    got = b.fetch_alignments()
    want = oix.align_single_allbest(q, lens, wpq, mm, max_out, report_type=report_type)
    assert (got[0] == want[0]).all() and (got[2] == want[2]).all()
    for k in (3, 4, 5):
        assert len(got[k]) == len(want[k]) and (got[k] == want[k]).all()
    b.search(mm, max_out)
    hits, offs, tot, werr = b.fetch_hits()
    oh, oo, ot, ow, _ = oix.search(q, lens, wpq, mm, max_out=max_out)
    assert (offs == oo).all() and (tot == ot).all() and (werr == ow).all() and (hits == oh).all()
    b.close()
    return got


def test_empty_and_single_read_batches(world):
    rng, codes, hi, ix, oix = world
    b = s3dp.Batch(ix, 32, 8)
    q, lens, wpq = s3dp.pack_reads([], 100)
    b.upload(q, lens)
    b.align_single(2, 1000, 2)
    phase, werr, offs, pos, strand, mis = b.fetch_alignments()
    assert len(phase) == 0 and list(offs) == [0] and len(pos) == 0
    b.search(2, 1000)
    hits, offs, tot, _ = b.fetch_hits()
    assert len(hits) == 0 and list(offs) == [0]
    b.align_pair(2, 8000, 2, 1, 500, 1000)
    assert len(b.fetch_pairs()[0]) == 0
    b.close()
    got = compare(ix, oix, [seq(codes[1234:1334])], 100, 3)
    assert got[0][0] > 0 and 1234 in list(got[3])


@pytest.mark.parametrize("L", [16, 17, 20, 25, 26, 31, 32, 33, 63, 64, 65])
def test_short_reads(world, L):
    rng, codes, hi, ix, oix = world
    reads = []
    for k in range(96):
        st = int(rng.integers(0, len(codes) - L))
        s = codes[st:st + L].copy()
        if k % 3 == 1:
            s[int(rng.integers(0, L))] ^= 1
        if k % 2:
            s = COMP[s[::-1]]
        reads.append(seq(s))
    for mm in (0, 1, 2):
        compare(ix, oix, reads, 64 if L <= 64 else 128, mm)


@pytest.mark.parametrize("L,mm", [(1024, 2), (1000, 4), (513, 1), (300, 3)])
def test_longest_reads_without_seeds(world, L, mm):
    """whole-read search of very long reads (no long-read mode): 64 words per read, the widest LDS layout of the kernel"""
    rng, codes, hi, ix, oix = world
    reads = []
    for k in range(40):
        st = int(rng.integers(10000, len(codes) - L))
        s = codes[st:st + L].copy()
        for p in rng.choice(L, size=k % (mm + 2), replace=False):
            s[p] = (s[p] + 1) % 4
        if k % 2:
            s = COMP[s[::-1]]
        reads.append(seq(s))
    compare(ix, oix, reads, 1024, mm)


def test_text_boundaries_and_homopolymers(world):
    rng, codes, hi, ix, oix = world
    n = len(codes)
    reads = [seq(codes[0:100]), seq(codes[n - 100:n]), seq(COMP[codes[0:100][::-1]]), seq(COMP[codes[n - 100:n][::-1]]),
             seq(codes[1:101]), seq(codes[n - 101:n - 1]),
             b"A" * 100, b"T" * 100, b"AC" * 50, b"GT" * 50, b"A" * 99 + b"C", b"N" * 100, seq(codes[300:350]) + b"N" * 3 + seq(codes[353:400])]
    for mm in (0, 2, 4):
        for cap in (1000, 7):
            got = compare(ix, oix, reads, 100, mm, max_out=cap)
    counts = got[2][1:] - got[2][:-1]
    assert counts[6] == 7 and counts[7] == 7                 # all-A / all-T: the hit limit cuts the list
    for rt in (1, 3, 4):
        compare(ix, oix, reads, 100, 3, report_type=rt)


def test_a_batch_that_lost_a_buffer_while_growing_refuses_further_calls():
    """bgrow frees the old buffer before the larger one exists when both do not fit; if the second allocation fails too the batch has a null
    buffer under an old capacity.  It must answer every later call with S3_ENOMEM instead of launching kernels on it (S3DP_TEST_FAIL_GROW=1: the first
    growth of the process finds the device full; a tiny record floor forces that growth)."""
    import os
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, "soap3-dp_amd/py"); sys.path.insert(0, "tests")
import s3dp
from conftest import golden_reads, host_index
reads = golden_reads("g2")
q, lens, wpq = s3dp.pack_reads(reads, 120)
ix = s3dp.Index(host_index("g2"))
b = s3dp.Batch(ix, len(reads), wpq)
b.upload(q, lens)
errs = []
for k in range(3):
    try:
        b.search(2, 1000)
        errs.append("ok")
    except s3dp.S3Error as e:
        errs.append(str(e))
assert "unusable" in errs[0] or "hipMalloc" in errs[0], errs
assert all("lost a buffer" in e for e in errs[1:]), errs
try:
    b.upload(q, lens)
    raise SystemExit("upload on a failed batch succeeded")
except s3dp.S3Error as e:
    assert "lost a buffer" in str(e)
b.close()
b2 = s3dp.Batch(ix, len(reads), wpq)          # the growth hook is spent: a fresh batch works, and grows
b2.upload(q, lens)
b2.search(2, 1000)
hits, offs, tot, werr = b2.fetch_hits()
assert len(hits) > 4096
print("refused-ok", len(hits))
'''
    env = dict(os.environ, S3DP_TEST_FAIL_GROW="1", S3DP_REC_FLOOR="16", S3DP_REC_PER_READ="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "refused-ok" in r.stdout, r.stdout + r.stderr
