"""-m gpu: the streamed form of the single-end ladder (include/s3dp.h: s3_stream_*) against the resident batch and the oracle -- batches in flight on
their own HIP streams, results in submission order; the pinned staging from the process's cache (s3_host_staging_reserve), buffers that grow in the
middle of a run (the retired list instead of hipFree, s3dp_internal.h: bgrow), fewer ladders on the device than slots (S3DP_STREAM_DEVICE_JOBS)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import s3dp
import s3o
from s3dp import builder

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def world():
    rng = np.random.default_rng(99)
    n = 400_003
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    unit = rng.integers(0, 4, size=300, dtype=np.uint8)
    for k in range(150):                                       # a repeat family: reads inside it come back with a hundred occurrences and more
        p = int(rng.integers(0, n - 400))
        codes[p:p + 300] = unit
        codes[p + int(rng.integers(0, 300))] ^= 1
    hi = builder.build_host_index(codes, device="cpu")
    return rng, codes, hi, s3dp.Index(hi), s3o.Index(host_index=hi)


def make_reads(rng, codes, n, L=100):
    out = []
    for _ in range(n):
        p = int(rng.integers(0, len(codes) - L))
        r = codes[p:p + L].copy()
        for j in rng.choice(L, size=int(rng.integers(0, 4)), replace=False):
            r[j] = (r[j] + 1 + int(rng.integers(0, 3))) & 3
        if rng.integers(0, 2):
            r = (3 - r)[::-1]
        out.append(bytes(b"ACGT"[c] for c in r))
    return out


def run_stream(ix, q, lens, wpq, batch_reads, depth, mm=3, max_out=1000):
    n = len(lens)
    assert batch_reads % 32 == 0
    st = s3dp.Stream(ix, batch_reads, wpq, depth)
    got, starts, nxt, inflight = [], list(range(0, n, batch_reads)), 0, 0
    try:
        while nxt < len(starts) or inflight:
            while nxt < len(starts) and inflight < depth:
                a = starts[nxt]
                m = min(batch_reads, n - a)
                ru = (m + 31) // 32 * 32                                  # (the packed layout interleaves the words of 32 reads: batches start at multiples of 32)
                bq, bl, _ = st.acquire()
                bq[:ru * wpq] = q[a * wpq:(a + ru) * wpq]
                bl[:m] = lens[a:a + m]
                st.submit(m, mm, max_out, 2)
                nxt += 1
                inflight += 1
            got.append(st.next(copy=True))
            st.release()
            inflight -= 1
    finally:
        st.close()
    return got


def same_as_resident(ix, q, lens, wpq, got, batch_reads, mm=3, max_out=1000):
    n = len(lens)
    b = s3dp.Batch(ix, n, wpq)
    b.upload(q, lens)
    b.align_single(mm, max_out, 2)
    phase, werr, offs, pos, strand, mis = b.fetch_alignments()
    b.close()
    a = 0
    for (ph, we, of, po, sr, mi, _ms) in got:
        m = len(ph)
        assert (ph == phase[a:a + m]).all() and (we == werr[a:a + m]).all()
        assert (of == offs[a:a + m + 1] - offs[a]).all()
        lo, hi_ = int(offs[a]), int(offs[a + m])
        assert (po == pos[lo:hi_]).all() and (sr == strand[lo:hi_]).all() and (mi == mis[lo:hi_]).all()
        a += m
    assert a == n
    return phase, werr, offs, pos, strand, mis


@pytest.mark.parametrize("depth", [1, 3, 6])
def test_streamed_batches_equal_the_resident_batch_and_the_oracle(world, depth):
    rng, codes, hi, ix, oix = world
    reads = make_reads(np.random.default_rng(5), codes, 5000)
    q, lens, wpq = s3dp.pack_reads(reads, 100)
    got = run_stream(ix, q, lens, wpq, 704, depth)                       # (eight batches, the last one short)
    res = same_as_resident(ix, q, lens, wpq, got, 704)
    want = oix.align_single_allbest(q, lens, wpq, 3, 1000, report_type=2)
    assert (res[0] == want[0]).all() and (res[2] == want[2]).all()
    for k in (3, 4, 5):
        assert len(res[k]) == len(want[k]) and (res[k] == want[k]).all()
    assert int(res[2][-1]) > 20 * len(reads) // 10                      # the family's reads bring their lists


def test_reserved_staging_is_used_and_given_back(world):
    rng, codes, hi, ix, oix = world
    reads = make_reads(np.random.default_rng(6), codes, 3000)
    q, lens, wpq = s3dp.pack_reads(reads, 100)
    s3dp.host_staging_reserve(512, wpq, 3, 64)
    got = run_stream(ix, q, lens, wpq, 512, 3)
    same_as_resident(ix, q, lens, wpq, got, 512)
    got = run_stream(ix, q, lens, wpq, 512, 3)                           # (a second stream of the process takes the first one's blocks)
    same_as_resident(ix, q, lens, wpq, got, 512)
    s3dp.host_staging_release()
    with pytest.raises(Exception):
        s3dp.host_staging_reserve(0, wpq, 3, 64)


def test_fewer_ladders_on_the_device_than_slots(world):
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_stream as t
import s3dp, s3o
from s3dp import builder
rng = np.random.default_rng(99)
codes = rng.integers(0, 4, size=200003, dtype=np.uint8)
hi = builder.build_host_index(codes, device="cpu")
ix = s3dp.Index(hi)
reads = t.make_reads(np.random.default_rng(7), codes, 4000)
q, lens, wpq = s3dp.pack_reads(reads, 100)
got = t.run_stream(ix, q, lens, wpq, 512, 5)
t.same_as_resident(ix, q, lens, wpq, got, 512)
print("ok")
""" % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "soap3-dp_amd", "py"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, S3DP_STREAM_DEVICE_JOBS="2"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
