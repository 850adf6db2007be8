import gzip
import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "soap3-dp_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def read_gz_lines(path):
    with gzip.open(path, "rt") as f:
        return [l for l in f.read().split("\n") if l]


_HOST_INDEX = {}


def host_index(name):
    """HostIndex of a golden genome; lookup tables are rebuilt by our own builder and cached in /tmp"""
    import s3dp
    if name in _HOST_INDEX:
        return _HOST_INDEX[name]
    prefix = os.path.join(GOLDEN, name, "index")
    cache = "/tmp/s3dp_lkt_%s_%s.npz" % (name, hashlib.sha1(open(prefix + ".pac", "rb").read()).hexdigest()[:12])
    if os.path.exists(cache):
        z = np.load(cache)
        hi = s3dp.HostIndex.from_files(prefix, lkt=z["lkt"], rev_lkt=z["rev_lkt"])
    else:
        hi = s3dp.HostIndex.from_files(prefix)
        try:
            np.savez(cache, lkt=hi.lkt, rev_lkt=hi.rev_lkt)
        except OSError:
            pass
    _HOST_INDEX[name] = hi
    return hi


def golden_reads(name):
    import s3o
    return s3o.read_fastq(os.path.join(GOLDEN, name, "reads.fq"))


def hits_to_lines(hits, offs, tot, werr, n):
    """the text form of `ref_harness search` (per case: SA ranges, then check-and-extend hits)"""
    lines = []
    for r in range(n):
        hs = hits[int(offs[r]):int(offs[r + 1])]
        if len(hs):
            for c in sorted(set(int(x) for x in hs["cas"])):
                hc = hs[hs["cas"] == c]
                for h in hc[hc["kind"] == 0]:
                    lines.append("%d %d S %d %d %d %d" % (r, c, h["a"], h["b"], h["strand"], h["nmis"]))
                for h in hc[hc["kind"] == 1]:
                    lines.append("%d %d O %d %d %d %d" % (r, c, h["a"], h["b"], h["strand"], h["nmis"]))
        lines.append("%d T %d %s" % (r, tot[r], " ".join(str(int(x)) for x in werr[r][:5])))
    return lines
