"""The CPU oracle (oracle/s3_oracle.c) pinned against the REAL reference CPU path.

Golden files under tests/golden/ were produced by oracle/_ref/ref_harness (the reference's own
BGS-HostAlgnmtAlgo2.cpp / SRA2BWTMdl.c / CPUfunctions.cpp objects) -- see tests/golden/make_fixtures.sh.
"""
import os

import pytest

import s3o
from conftest import GOLDEN, golden_reads, hits_to_lines, host_index, read_gz_lines


def test_model_tables_all_lengths():
    ref = read_gz_lines(os.path.join(GOLDEN, "models.txt.gz"))
    mine = []
    for L in range(25, 1025):
        for e in range(5):
            mine += s3o.model_lines(L, e)
    assert len(ref) == len(mine) == 23000
    assert ref == mine


@pytest.mark.parametrize("name,maxlen", [("g1", 100), ("g2", 120), ("g6", 100)])
@pytest.mark.parametrize("mm", [0, 1, 2, 3, 4])
def test_search_matches_reference(name, maxlen, mm):
    ix = s3o.Index(host_index=host_index(name))
    reads = golden_reads(name)
    q, lens, wpq = s3o.pack_reads(reads, maxlen)
    hits, offs, tot, werr, st = ix.search(q, lens, wpq, mm, max_out=1000)
    mine = hits_to_lines(hits, offs, tot, werr, len(reads))
    ref = read_gz_lines(os.path.join(GOLDEN, name, "search_mm%d.txt.gz" % mm))
    assert mine == ref
    ix.close()
