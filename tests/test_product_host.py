"""Host-side pieces of the product that need no GPU: model tables, lookup-table builder, read packer,
and that libs3dp.so loads and exports the whole C-ABI of include/s3dp.h."""
import hashlib
import os
import re

import numpy as np
import pytest

import s3dp
import s3o
from conftest import GOLDEN, ROOT, golden_reads, host_index, read_gz_lines


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "s3dp.h")).read()
    declared = set(re.findall(r"\b(s3_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = s3dp.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "libs3dp.so does not export " + name
    assert set(s3dp.EXPORTS) <= declared


def test_product_model_tables_match_reference():
    ref = read_gz_lines(os.path.join(GOLDEN, "models.txt.gz"))
    mine = []
    for L in range(25, 1025):
        for e in range(5):
            mine += [l for l in s3dp.model_text(L, e).split("\n") if l]
    assert mine == ref


@pytest.mark.parametrize("name", ["g1", "g2"])
def test_lookup_table_builder_matches_reference_files(name):
    hi = host_index(name)
    want = open(os.path.join(GOLDEN, name, "lkt.sha256")).read().split()
    for tab, sha in zip((hi.lkt, hi.rev_lkt), want):
        blob = np.array([13], dtype=np.int32).tobytes() + tab.tobytes()
        assert hashlib.sha256(blob).hexdigest() == sha


def test_packer_matches_reference_layout():
    reads = golden_reads("g2")
    q1, l1, w1 = s3o.pack_reads(reads, 120)
    q2, l2, w2 = s3dp.pack_reads(reads, 120)
    assert w1 == w2 == 8
    assert (q1 == q2).all() and (l1 == l2).all()


def test_no_device_means_loud_failure():
    if s3dp.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(s3dp.S3Error) as e:
        s3dp.Index(host_index("g1"))
    assert "no HIP device" in str(e.value)


def test_shared_index_mapping_equals_the_private_copy():
    """HostIndex.from_files(shared=True) maps the files copy-on-write (ranks of a node share the pages) and must describe the same index"""
    import numpy as np
    import s3dp
    from conftest import GOLDEN, host_index
    for g in ("g1", "g6"):
        a = host_index(g)
        b = s3dp.HostIndex.from_files(os.path.join(GOLDEN, g, "index"), lkt=a.lkt, rev_lkt=a.rev_lkt, shared=True)
        assert a.n == b.n and a.sa_interval == b.sa_interval and b.codes is None
        assert np.array_equal(a.pac, b.pac) and np.array_equal(a.sa, b.sa)
        for h in ("fwd", "rev"):
            ha, hb = getattr(a, h), getattr(b, h)
            assert ha["inverseSa0"] == hb["inverseSa0"] and ha["cum"] == hb["cum"]
            assert np.array_equal(ha["bwt"], hb["bwt"]) and np.array_equal(ha["occ"], hb["occ"])
        da, db = a.desc(), b.desc()
        assert da.fwd.bwtWords == db.fwd.bwtWords and da.packedWords == db.packedWords and da.textLength == db.textLength
