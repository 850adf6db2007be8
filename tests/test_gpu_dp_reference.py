"""-m gpu: the semi-global DP against the REFERENCE'S OWN device code.

oracle/_ref/libdpref.so is built in the authoring container by oracle/ref.mk: lines 35-512 of the reference's DV-DPfunctions.cu
(DPScoreNHitPos, GenerateDPTable, SemiGlobalAligntment, GPUBacktrack) compiled with hipcc for gfx950 from where they lie, behind a host
driver of ours (oracle/ref_dp_driver.hip).  CUDA and HIP share the kernel language; the two stream edits the hardware forces (no texture
units on CDNA) are listed in that driver.  The reference kernels, the C restatement (oracle/s3_oracle_dp.c) and the product's wavefront
kernels (s3_dp_align) must agree on scores, alignment start columns, start offsets (scheme 2), actual right clips, tie counts and the
traceback op strings -- on random problems with soft-clip allowances and anchors, both alignment schemes.  This pins rows a13-a16."""
import ctypes as C
import os

import numpy as np
import pytest

import s3dp
import s3o
from conftest import ROOT
from dp_cases import make_problems

pytestmark = pytest.mark.gpu
SC = (1, -2, -3, -1)
LIB = os.path.join(ROOT, "oracle", "_ref", "libdpref.so")


def dpref_align(P, max_dna, max_table, max_read, scheme, clip=True, anchors=True):
    lib = C.CDLL(LIB)
    f = lib.dpref_align
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5
    n = len(P["read_len"])
    u32 = lambda a: np.ascontiguousarray(a, dtype=np.uint32)
    dna, dl, rd, rl = u32(P["dna"]), u32(P["dna_len"]), u32(P["reads"]), u32(P["read_len"])
    cl = u32(P["clip_lt"]) if clip else None
    cr = np.array(P["clip_rt"], dtype=np.uint32) if clip else None
    al = u32(P["a_lt"]) if anchors else None
    ar = u32(P["a_rt"]) if anchors else None
    cut = np.ascontiguousarray(P["cutoffs"], dtype=np.int32)
    sc = np.zeros(n, dtype=np.int32)
    hit, start, cnt = (np.zeros(n, dtype=np.uint32) for _ in range(3))
    pat = np.zeros((n, max_read + max_table), dtype=np.uint8)
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    assert f(p(dna), p(dl), max_dna, max_table, p(rd), p(rl), max_read, n, p(cl), p(cr), p(al), p(ar), SC[0], SC[1], SC[2], SC[3], p(cut), scheme, p(sc), p(hit),
             p(start), p(cnt), p(pat)) == 0
    return sc, hit, cnt, pat, cr, start


def op_string(row):
    b = bytes(row)
    return b[:b.index(0) + 1]


@pytest.mark.parametrize("scheme", [1, 2])
# the last two: windows of 1548 columns (single -L 1024: four waves need 80 KB of dynamic LDS, above the default limit) and of 5000 columns (two waves per block)
@pytest.mark.parametrize("max_read,max_dna,seed,rl", [(152, 650, 31, (60, 150)), (104, 200, 32, (30, 100)), (304, 700, 33, (150, 300)), (68, 130, 34, (16, 64)), (128, 300, 35, (100, 127)),
                                                      (1028, 1548, 36, (900, 1024)), (204, 5000, 37, (150, 200))])
def test_reference_kernels_restatement_and_product_agree(scheme, max_read, max_dna, seed, rl):
    if max_dna > 2000 and scheme == 2:
        pytest.skip("the 5000-column case pins the kernel's two-waves-per-block launch; scheme 2 (small-memory mode of the reference, DV-DPfunctions.cu:600-612) "
                    "is compared on the shapes its engines use")
    # reads are always shorter than maxReadLength in the reference's engines (maxReadLength = (longest / 4 + 1) * 4, DV-DPfunctions.cu:1580,2223):
    # its table rows hold maxReadLength cells for the read positions 0..readLength, so readLength == maxReadLength is outside its range
    assert rl[1] < max_read
    assert os.path.exists(LIB), "oracle/_ref/libdpref.so is missing: run `make -f oracle/ref.mk` where /root/reference exists"
    n = 256
    P = make_problems(n, max_read, max_dna, seed, read_len_range=rl)
    al = s3dp.DpAligner(n, max_read, max_dna, scheme=scheme, scores=SC)
    mine = al.align(P["dna"], P["dna_len"], P["reads"], P["read_len"], P["cutoffs"], clip_lt=P["clip_lt"], clip_rt=P["clip_rt"], anchor_lt=P["a_lt"],
                    anchor_rt=P["a_rt"])
    orc = s3o.dp_align(P["dna"], P["dna_len"], max_dna, al.max_table, P["reads"], P["read_len"], max_read, P["cutoffs"], scheme=al.scheme, scores=SC,
                       clip_lt=P["clip_lt"], clip_rt=P["clip_rt"], anchor_lt=P["a_lt"], anchor_rt=P["a_rt"])
    ref = dpref_align(P, max_dna, al.max_table, max_read, al.scheme)
    for who, got in (("restatement", orc), ("product", mine)):
        assert (got[0] == ref[0]).all(), who + ": scores"
        assert (got[2] == ref[2]).all(), who + ": tie counts"
        assert (got[4] == ref[4]).all(), who + ": actual right clips"
        assert (got[1] == ref[1]).all(), who + ": start columns"
    assert (orc[5] == ref[5]).all(), "restatement: start offsets"
    ok = ref[0] >= P["cutoffs"]
    assert ok.sum() > n // 3
    for t in np.nonzero(ok)[0]:
        assert op_string(orc[3][t]) == op_string(ref[3][t]), ("restatement", t)
        assert op_string(mine[3][t]) == op_string(ref[3][t]), ("product", t)
    al.close()


@pytest.mark.parametrize("scheme", [1, 2])
def test_chunk_boundaries_and_long_clips(scheme):
    """read lengths around the 64-row chunks of the wavefront kernel (63, 64, 65, 127, 128, 129, 191, 192, 193), windows of 66 to 419
    columns, soft-clip allowances that reach into the second and third chunk: the reference's kernels, the restatement and the
    product agree on everything"""
    assert os.path.exists(LIB)
    n, max_read, max_dna = 288, 200, 420
    P = make_problems(n, max_read, max_dna, 51, lengths=[63, 64, 65, 127, 128, 129, 191, 192, 193], clip_max=100)
    assert (P["clip_lt"] > 64).sum() > 5 and (P["clip_rt"] > 64).sum() > 5
    al = s3dp.DpAligner(n, max_read, max_dna, scheme=scheme, scores=SC)
    mine = al.align(P["dna"], P["dna_len"], P["reads"], P["read_len"], P["cutoffs"], clip_lt=P["clip_lt"], clip_rt=P["clip_rt"], anchor_lt=P["a_lt"],
                    anchor_rt=P["a_rt"])
    orc = s3o.dp_align(P["dna"], P["dna_len"], max_dna, al.max_table, P["reads"], P["read_len"], max_read, P["cutoffs"], scheme=al.scheme, scores=SC,
                       clip_lt=P["clip_lt"], clip_rt=P["clip_rt"], anchor_lt=P["a_lt"], anchor_rt=P["a_rt"])
    ref = dpref_align(P, max_dna, al.max_table, max_read, al.scheme)
    for who, got in (("restatement", orc), ("product", mine)):
        assert (got[0] == ref[0]).all(), who + ": scores"
        assert (got[2] == ref[2]).all(), who + ": tie counts"
        assert (got[4] == ref[4]).all(), who + ": actual right clips"
        assert (got[1] == ref[1]).all(), who + ": start columns"
    ok = ref[0] >= P["cutoffs"]
    assert ok.sum() > n // 4
    for t in np.nonzero(ok)[0]:
        assert op_string(orc[3][t]) == op_string(ref[3][t]), ("restatement", t)
        assert op_string(mine[3][t]) == op_string(ref[3][t]), ("product", t)
    al.close()


def test_reference_kernels_without_clips_and_anchors():
    assert os.path.exists(LIB)
    P = make_problems(128, 104, 160, seed=41, clip=False, anchors=False, read_len_range=(40, 100))
    al = s3dp.DpAligner(128, 104, 160, scheme=1, scores=SC)
    mine = al.align(P["dna"], P["dna_len"], P["reads"], P["read_len"], P["cutoffs"])
    ref = dpref_align(P, 160, al.max_table, 104, 1, clip=False, anchors=False)
    assert (mine[0] == ref[0]).all() and (mine[1] == ref[1]).all() and (mine[2] == ref[2]).all()
    for t in np.nonzero(ref[0] >= P["cutoffs"])[0]:
        assert op_string(mine[3][t]) == op_string(ref[3][t])
    al.close()
