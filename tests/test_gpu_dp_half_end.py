"""-m gpu: default DP for half-aligned pairs (s3_dp_half_end: window geometry + device packing + DP + CIGAR decode).
Property test kept from round 1 (the record-for-record comparison with the reference's own engine is tests/test_gpu_dp_pipeline.py); it checks the
semantics on simulated pairs whose second end carries a known insertion / deletion / substitutions: the end is found at the right
place, and the CIGAR, applied to the reference, reproduces the read with exactly the reported score and edit distance."""
import re

import numpy as np
import pytest

import s3dp
from s3dp import builder

pytestmark = pytest.mark.gpu

COMP = np.array([3, 2, 1, 0], dtype=np.uint8)
MATCH, MISMATCH, GAPO, GAPE = 1, -2, -3, -1


def apply_cigar(cigar, read, ref, pos):
    """walk the CIGAR; returns (score, edit distance, reference span)"""
    score = edits = 0
    rp, qp = pos, 0
    for n, op in re.findall(r"(\d+)([MmIDS])", cigar):
        n = int(n)
        if op in "Mm":                                         # the engine's CIGAR keeps mismatch runs apart as 'm' (DV-DPfunctions.cu:385)
            seg_q, seg_r = read[qp:qp + n], ref[rp:rp + n]
            mm = int((seg_q != seg_r).sum())
            assert mm == (n if op == "m" else 0), (cigar, op, n)
            score += (n - mm) * MATCH + mm * MISMATCH
            edits += mm
            qp += n
            rp += n
        elif op == "I":
            score += GAPO + (n - 1) * GAPE
            edits += n
            qp += n
        elif op == "D":
            score += GAPO + (n - 1) * GAPE
            edits += n
            rp += n
        elif op == "S":
            qp += n
    assert qp == len(read)
    return score, edits, rp - pos


@pytest.fixture(scope="module")
def setup():
    rng = np.random.default_rng(99)
    n = 60000
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    hi = builder.build_host_index(codes, device="cpu")
    ix = s3dp.Index(hi)
    return rng, codes, ix


def make_cases(rng, codes, L=100, count=120):
    """pairs: end A exact on '+', end B (the DP target) from the same fragment with an edit; returns reads, occ records, truths"""
    reads, occ, truth = [], [], []
    for p in range(count):
        ins = int(rng.integers(250, 480))
        st = int(rng.integers(1000, len(codes) - 2000))
        frag = codes[st:st + ins + 20].copy()
        a = frag[:L].copy()
        kind = p % 4
        lo = ins - L                                           # B covers frag[lo : lo + span]
        if kind == 0:                                          # 3-base deletion from the read (reference has 3 extra bases)
            cut = int(rng.integers(30, 70))
            b = np.concatenate([frag[lo - 3:lo - 3 + cut], frag[lo - 3 + cut + 3:lo + L]])
            ref_start, edits, want_score = lo - 3, 3, (L) * MATCH + GAPO + 2 * GAPE
        elif kind == 1:                                        # 2-base insertion in the read
            cut = int(rng.integers(30, 70))
            extra = (frag[lo + cut:lo + cut + 2] + 1 + rng.integers(0, 2, size=2, dtype=np.uint8)) % 4
            b = np.concatenate([frag[lo + 2:lo + cut], extra.astype(np.uint8), frag[lo + cut:lo + L]])
            ref_start, edits, want_score = lo + 2, 2, (L - 2) * MATCH + GAPO + GAPE
        elif kind == 2:                                        # 5 substitutions: too many for the BWT stage
            b = frag[lo:lo + L].copy()
            for q in rng.choice(np.arange(10, 90), size=5, replace=False):
                b[q] = (b[q] + 1 + rng.integers(0, 3)) % 4
            ref_start, edits, want_score = lo, 5, (L - 5) * MATCH + 5 * MISMATCH
        else:                                                  # exact
            b = frag[lo:lo + L].copy()
            ref_start, edits, want_score = lo, 0, L * MATCH
        assert len(b) == L
        swap = p % 8 >= 4                                      # which end is the BWT-aligned one
        read_a, read_b = a, COMP[b[::-1]]                      # B is sequenced from the '-' strand
        if not swap:
            reads += [read_a, read_b]
            occ.append((2 * p, st, 1, 0))                      # A aligned on '+', B goes to the DP on '-'
            truth.append(dict(dp_end=2, pos=st + ref_start, fwd=b, edits=edits, score=want_score, strand=2))
        else:
            # B is the aligned one (on '-', exact kinds only make sense as BWT hits; use its reference span start), A goes to the DP on '+'
            reads += [read_b, read_a]
            bpos = st + ref_start
            occ.append((2 * p, bpos, 2, 0))
            truth.append(dict(dp_end=2, pos=st, fwd=a, edits=0, score=L * MATCH, strand=1))
    return reads, occ, truth


def test_half_end_dp_recovers_simulated_mates(setup):
    rng, codes, ix = setup
    L = 100
    reads, occ, truth = make_cases(rng, codes, L)
    strs = [bytes(b"ACGT"[c] for c in r) for r in reads]
    q, lens, wpq = s3dp.pack_reads(strs, 120)
    b = s3dp.Batch(ix, len(reads), wpq)
    b.upload(q, lens)
    dp = s3dp.DpAligner(4096, 104, 500 - 1 + 104 + 1, scheme=1)
    occ_arr = np.zeros(len(occ), dtype=s3dp.OCC_DTYPE)
    for k, (rid, pos, strand, mis) in enumerate(occ):
        occ_arr[k] = (rid, pos, strand, mis, (0, 0))
    res, cigars = dp.half_end(b, lens, occ_arr, 1, 500, (30, 30), 49, 49)
    assert len(res) == len(occ)
    kinds = set()
    for k, (r, cg, t) in enumerate(zip(res, cigars, truth)):
        assert r["readID"] == 2 * k
        assert r["whichFromDP"] == 1, (k, r)                   # the second read of the pair came from the DP
        assert cg is not None
        assert r["strand_2"] == t["strand"] and r["strand_1"] == occ[k][2] and r["algnmt_1"] == occ[k][1]
        pos = int(r["algnmt_2"])
        score, edits, span = apply_cigar(cg, t["fwd"], codes, pos)
        assert score == r["score_2"] == t["score"], (k, cg, score, r["score_2"], t["score"])
        assert edits == r["editdist"] == t["edits"], (k, cg)
        assert abs(pos - t["pos"]) <= 3, (k, cg, pos, t["pos"])
        if not set(cg) & set("SID"):
            assert pos == t["pos"]
        lo, hi = min(pos, occ[k][1]), max(pos + span, occ[k][1] + L)
        assert r["insertSize"] == hi - lo, (k, cg, r["insertSize"], hi - lo)
        kinds.add(re.sub(r"\d+", "", cg).replace("m", "M").replace("MM", "M").replace("MM", "M"))
    assert {"M", "MDM", "MIM"} <= kinds
    dp.close()
    b.close()
