"""-m gpu: the wavefront DP kernels (s3dp_dp.hip) through the C-ABI against the CPU restatement of the reference's
DP (oracle/s3_oracle_dp.c): scores, alignment start columns, actual right clips, tie counts and the op strings must be
identical, for both alignment schemes, with soft-clip allowances and anchors."""
import numpy as np
import pytest

import s3dp
import s3o
from dp_cases import make_problems

pytestmark = pytest.mark.gpu
SC = (1, -2, -3, -1)


@pytest.mark.parametrize("scheme", [1, 2])
@pytest.mark.parametrize("max_read,max_dna,seed,rl", [(152, 650, 11, (60, 150)), (100, 200, 12, (30, 100)), (304, 700, 13, (150, 300)), (64, 130, 14, (16, 64))])
def test_dp_matches_oracle(scheme, max_read, max_dna, seed, rl):
    n = 200
    P = make_problems(n, max_read, max_dna, seed, read_len_range=rl)
    al = s3dp.DpAligner(n, max_read, max_dna, scheme=scheme, scores=SC)
    sc, hit, cnt, pat, crt = al.align(P["dna"], P["dna_len"], P["reads"], P["read_len"], P["cutoffs"], clip_lt=P["clip_lt"], clip_rt=P["clip_rt"],
                                      anchor_lt=P["a_lt"], anchor_rt=P["a_rt"])
    osc, ohit, ocnt, opat, ocrt, ostart = s3o.dp_align(P["dna"], P["dna_len"], max_dna, al.max_table, P["reads"], P["read_len"], max_read, P["cutoffs"],
                                                       scheme=al.scheme, scores=SC, clip_lt=P["clip_lt"], clip_rt=P["clip_rt"], anchor_lt=P["a_lt"],
                                                       anchor_rt=P["a_rt"])
    assert (sc == osc).all()
    assert (cnt == ocnt).all()
    assert (crt == ocrt).all()
    assert (hit == ohit).all()
    ok = sc >= P["cutoffs"]
    assert ok.sum() > n // 3
    for t in np.nonzero(ok)[0]:
        a, b = bytes(pat[t]), bytes(opat[t])
        assert a[:a.index(0) + 1] == b[:b.index(0) + 1], t
    al.close()


def test_dp_without_optional_arrays():
    P = make_problems(64, 100, 160, seed=21, clip=False, anchors=False)
    al = s3dp.DpAligner(64, 100, 160, scheme=1, scores=SC)
    sc, hit, cnt, pat, crt = al.align(P["dna"], P["dna_len"], P["reads"], P["read_len"], P["cutoffs"])
    osc, ohit, ocnt, opat, _, _ = s3o.dp_align(P["dna"], P["dna_len"], 160, al.max_table, P["reads"], P["read_len"], 100, P["cutoffs"], scheme=1, scores=SC)
    assert (sc == osc).all() and (hit == ohit).all() and (cnt == ocnt).all()
    al.close()


# (score set, what it exercises): the reference's defaults; larger penalties; a penalty beyond what the paired kernel accepts (the launch falls back to one
# problem per wavefront)
SCORE_SETS = [((1, -2, -3, -1), "defaults"), ((2, -3, -5, -2), "larger"), ((5, -4, -12, -3), "larger still"), ((1, -260, -3, -1), "outside the paired range")]


@pytest.mark.parametrize("scores,what", SCORE_SETS, ids=[w for _, w in SCORE_SETS])
@pytest.mark.parametrize("n", [1, 2, 201])
def test_two_problems_per_wavefront_equal_one(monkeypatch, scores, what, n):
    """dpFillPairKernel sweeps two problems in the 16-bit halves of a wave's registers (scheme 1, scores within range): same scores, start columns, tie
    counts, right clips and traceback strings as the one-problem kernel (S3DP_DP_PAIRED=0) and as the CPU restatement -- neighbours of different read
    length and window, an odd problem at the end, chunk-boundary lengths, clips into the second chunk."""
    max_read, max_dna = 200, 420
    P = make_problems(n, max_read, max_dna, 77 + n, lengths=[63, 150, 64, 65, 30, 127, 128, 129, 191, 192, 193, 100], clip_max=90)
    out = {}
    for mode in ("1", "0", "fused"):
        monkeypatch.setenv("S3DP_DP_PAIRED", "0" if mode == "0" else "1")
        monkeypatch.setenv("S3DP_DP_FUSED", "1" if mode == "fused" else "0")   # the paired kernel walking its own tables (measured slower, off by default)
        al = s3dp.DpAligner(n, max_read, max_dna, scheme=1, scores=scores)
        out[mode] = al.align(P["dna"], P["dna_len"], P["reads"], P["read_len"], P["cutoffs"], clip_lt=P["clip_lt"], clip_rt=P["clip_rt"], anchor_lt=P["a_lt"],
                             anchor_rt=P["a_rt"])
        al.close()
    orc = s3o.dp_align(P["dna"], P["dna_len"], max_dna, max_dna, P["reads"], P["read_len"], max_read, P["cutoffs"], scheme=1, scores=scores,
                       clip_lt=P["clip_lt"], clip_rt=P["clip_rt"], anchor_lt=P["a_lt"], anchor_rt=P["a_rt"])
    for mode in ("1", "0", "fused"):
        sc, hit, cnt, pat, crt = out[mode]
        assert (sc == orc[0]).all() and (hit == orc[1]).all() and (cnt == orc[2]).all() and (crt == orc[4]).all(), "S3DP_DP_PAIRED=" + mode
        for t in np.nonzero(sc >= P["cutoffs"])[0]:
            a, b = bytes(pat[t]), bytes(orc[3][t])
            assert a[:a.index(0) + 1] == b[:b.index(0) + 1], (mode, t)


def test_tie_counts_beyond_16_bits_in_the_paired_kernel(monkeypatch):
    """The paired fill kernel keeps a row's tie count in a 16-bit half (one increment per step at most: at most window + 63 per lane and chunk, which is why
    dpPairable refuses windows above 12,000 columns) and adds the rows' counts up in 32 bits.  Forced here: match score 0 on homopolymer windows of 3,000
    columns with right clips up to the whole read -- every cell of every row that may end the alignment ties at the maximum, W x (clipRt + 1) = hundreds of
    thousands of ties, far beyond 65,535 -- and the count must equal the one-problem kernel's and the CPU restatement's."""
    max_read, max_dna, n = 200, 3000, 6
    scores = (0, -2, -3, -1)
    rng = np.random.default_rng(5)
    dna, reads, clip_rt = [], [], []
    for t in range(n):
        L = (100, 150, 64, 190, 129, 65)[t]
        base = t % 4
        win = np.full(max_dna - 7 * t, base, dtype=np.uint8)
        rd = np.full(L, base, dtype=np.uint8)
        if t == 4:
            win[1500] = (base + 1) % 4                     # one foreign character: the ties split around it
        dna.append(win); reads.append(rd); clip_rt.append(L - 1 if t != 2 else 40)
    P = dict(dna=s3dp.pack_dp_sequences(dna, max_dna), dna_len=np.array([len(x) for x in dna], dtype=np.uint32),
             reads=s3dp.pack_dp_sequences(reads, max_read), read_len=np.array([len(x) for x in reads], dtype=np.uint32),
             clip_lt=np.zeros(n, dtype=np.uint32), clip_rt=np.array(clip_rt, dtype=np.uint32), cutoffs=np.full(n, -5, dtype=np.int32))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("S3DP_DP_PAIRED", mode)
        monkeypatch.setenv("S3DP_DP_FUSED", "0")
        al = s3dp.DpAligner(n, max_read, max_dna, scheme=1, scores=scores)
        out[mode] = al.align(P["dna"], P["dna_len"], P["reads"], P["read_len"], P["cutoffs"], clip_lt=P["clip_lt"], clip_rt=P["clip_rt"])
        al.close()
    orc = s3o.dp_align(P["dna"], P["dna_len"], max_dna, max_dna, P["reads"], P["read_len"], max_read, P["cutoffs"], scheme=1, scores=scores,
                       clip_lt=P["clip_lt"], clip_rt=P["clip_rt"])
    assert (orc[2] > 65535).sum() >= 4, orc[2]
    for mode in ("1", "0"):
        sc, hit, cnt, pat, crt = out[mode]
        assert (sc == orc[0]).all() and (hit == orc[1]).all() and (cnt == orc[2]).all() and (crt == orc[4]).all(), ("S3DP_DP_PAIRED=" + mode, cnt, orc[2])
