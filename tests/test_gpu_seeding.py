"""-m gpu: seeding for the DP stages (s3_seed_candidates: seed packing, exact + 1-mismatch seed search, hit limit, estimated positions,
merge) against a restatement built here on the pinned search oracle; plus the semantic check that reads too damaged for the BWT stage
get their true position among the candidates.  (Kept from round 1; the reference's own seeding engines are compared in tests/test_gpu_dp_pipeline.py.)"""
import ctypes as C

import numpy as np
import pytest

import s3dp
import s3o
from s3dp import builder

pytestmark = pytest.mark.gpu

COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def seed_table(stage, L):
    sl, n = C.c_int(), C.c_int()
    pos = (C.c_int * 128)()
    assert s3dp.lib().s3_seed_positions(stage, L, C.byref(sl), pos, C.byref(n)) == 0
    return sl.value, list(pos[:n.value])


@pytest.mark.parametrize("stage,max_hit", [(1, 10), (4, 100), (5, 1000)])
def test_seed_candidates_match_restatement(stage, max_hit):
    rng = np.random.default_rng(5 + stage)
    n = 80000
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    # a planted repeat family so that some seeds exceed the hit limit
    unit = codes[1000:1300].copy()
    for k in range(15):
        p = int(rng.integers(2000, n - 400))
        codes[p:p + 300] = unit
    hi = builder.build_host_index(codes, device="cpu")
    ix = s3dp.Index(hi)
    reads, truth = [], []
    for r in range(160):
        L = 100 if r % 2 == 0 else 150
        st = int(rng.integers(0, n - L)) if r % 10 else 1000 + int(rng.integers(0, 100))
        seg = codes[st:st + L].copy()
        kind = r % 4
        if kind == 1:                                     # 7 substitutions
            for q in rng.choice(L, size=7, replace=False):
                seg[q] = (seg[q] + 1 + rng.integers(0, 3)) % 4
        elif kind == 2:                                   # 4-base deletion
            c = int(rng.integers(40, 60))
            seg = np.concatenate([codes[st:st + c], codes[st + c + 4:st + L + 4]])
        elif kind == 3:                                   # 3-base insertion
            c = int(rng.integers(40, 60))
            seg = np.concatenate([codes[st:st + c], rng.integers(0, 4, size=3, dtype=np.uint8), codes[st + c:st + L - 3]])
        strand = int(rng.integers(1, 3))
        if strand == 2:
            seg = COMP[seg[::-1]]
        reads.append(bytes(b"ACGT"[c] for c in seg))
        truth.append((st, strand))
    q, lens, wpq = s3dp.pack_reads(reads, 150)
    b = s3dp.Batch(ix, len(reads), wpq)
    b.upload(q, lens)
    sb = s3dp.Batch(ix, len(reads) * 16, 8)
    cand, status = b.seed_candidates(sb, lens, stage, max_hit)
    # ---- restatement on the search oracle -----------------------------------------------------------------
    oix = s3o.Index(host_index=hi)
    seeds = []
    for r, rd in enumerate(reads):
        sl, pos = seed_table(stage, len(rd))
        for off in pos:
            if off >= 0 and sl >= 16 and off + sl <= len(rd):
                seeds.append((r, off, sl))
    sq, slens, swpq = s3dp.pack_reads([reads[r][off:off + sl] for r, off, sl in seeds], 128)
    keys, want_status = [], []
    h0, o0, t0, _, _ = oix.search(sq, slens, swpq, 0, max_out=0xFFFFFFFF)
    h1, o1, t1, _, _ = oix.search(sq, slens, swpq, 1, max_out=0xFFFFFFFF)
    for j, (r, off, sl) in enumerate(seeds):
        hits, offs, tot = (h0, o0, t0) if t0[j] > 0 else (h1, o1, t1)
        if tot[j] == 0:
            want_status.append(0)
            continue
        if tot[j] > max_hit:
            want_status.append(4)
            continue
        want_status.append(1)
        p, s, _ = oix.decode(hits[int(offs[j]):int(offs[j + 1])])
        for pos, strand in zip(p, s):
            est = (int(pos) - off) if strand == 1 else (int(pos) + sl + off - len(reads[r]))
            keys.append((r, est & 0xFFFFFFFF, int(strand)))
    # (read, position), the strands together: the reference's third sort pass -- by strand -- leaves its result in a buffer that is freed
    # (MC_RadixSort_8_8, DV-DPfunctions.h:90; s3dp_seed.hip singleMergeKernel tells the story), so one run per READ is merged, a kept position keeping its strand
    keys.sort()
    want = []
    prev_run, prev = None, 0
    for r, est, strand in keys:
        if r != prev_run:
            want.append((r, est, strand)); prev_run = r; prev = est
        elif prev + 50 < est:
            want.append((r, est, strand)); prev = est
    got = [(int(c["readID"]), int(c["pos"]), int(c["strand"])) for c in cand]
    assert list(status) == want_status
    assert got == want
    assert 4 in want_status or max_hit >= 100
    # ---- semantics: the true place of every read is among its candidates (+- the indel size) -------------
    found = 0
    for r, (st, strand) in enumerate(truth):
        if any(rr == r and ss == strand and abs(pp - st) <= 4 for rr, pp, ss in got):
            found += 1
    assert found >= 0.8 * len(truth), found            # reads inside the planted repeat lose their seeds to the hit limit
    oix.close()
    b.close()
    sb.close()


def test_single_end_dp_pipeline_recovers_damaged_reads():
    """seeds -> candidates -> DP: reads with 7 substitutions or an indel (beyond the 4-mismatch BWT stage) come back at their true place
    with a CIGAR that reproduces the read and the reported score"""
    import re
    from test_gpu_dp_half_end import apply_cigar, MATCH
    rng = np.random.default_rng(17)
    n = 60000
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    hi = builder.build_host_index(codes, device="cpu")
    ix = s3dp.Index(hi)
    reads, truth = [], []
    for r in range(96):
        L = 100 if r % 2 == 0 else 150
        st = int(rng.integers(200, n - L - 200))
        kind = r % 3
        if kind == 0:
            seg = codes[st:st + L].copy()
            for q in rng.choice(np.arange(5, L - 5), size=7, replace=False):
                seg[q] = (seg[q] + 1 + rng.integers(0, 3)) % 4
            best = (L - 7) * MATCH + 7 * -2
        elif kind == 1:
            c = int(rng.integers(40, 60))
            seg = np.concatenate([codes[st:st + c], codes[st + c + 4:st + L + 4]])
            best = L * MATCH + (-3 - 3)
        else:
            c = int(rng.integers(40, 60))
            seg = np.concatenate([codes[st:st + c], (codes[st + c:st + c + 3] + 1) % 4, codes[st + c:st + L - 3]])
            best = (L - 3) * MATCH + (-3 - 2)
        strand = int(rng.integers(1, 3))
        fwd = seg.copy()
        if strand == 2:
            seg = COMP[seg[::-1]]
        reads.append(bytes(b"ACGT"[c] for c in seg))
        truth.append((st, strand, fwd, best))
    q, lens, wpq = s3dp.pack_reads(reads, 150)
    b = s3dp.Batch(ix, len(reads), wpq)
    b.upload(q, lens)
    sb = s3dp.Batch(ix, len(reads) * 16, 8)
    cand, status = b.seed_candidates(sb, lens, 1, 10)
    dp = s3dp.DpAligner(4096, 152, 152 + 2 * 38 + 8, scheme=1)
    res, cigars = dp.single_end(b, lens, cand, 30, 49, 49)
    best_by_read = {}
    for r, cg in zip(res, cigars):
        k = int(r["readID"])
        if k not in best_by_read or r["score"] > best_by_read[k][0]["score"]:
            best_by_read[k] = (r, cg)
    ok = 0
    for k, (st, strand, fwd, best) in enumerate(truth):
        if k not in best_by_read:
            continue
        r, cg = best_by_read[k]
        score, edits, span = apply_cigar(cg, fwd, codes, int(r["algnmt"]))
        assert score == r["score"] and edits == r["editdist"], (k, cg)
        if r["strand"] == strand and abs(int(r["algnmt"]) - st) <= 4 and r["score"] >= best:
            ok += 1
    assert ok >= 0.8 * len(truth), ok                  # 7 substitutions can spoil all three 38-base seeds (one mismatch each at most)
    dp.close()
    b.close()
    sb.close()


@pytest.mark.parametrize("stage,max_hit", [(4, 100), (5, 1000)])
def test_deep_dp_recovers_pairs_whose_ends_both_fail_the_bwt_stage(stage, max_hit):
    """paired seeding -> pair candidates inside the insert window -> left DP -> right DP bounded by the left hit: both ends damaged
    (6 substitutions or an indel each), both orientations of the pair"""
    from test_gpu_dp_half_end import apply_cigar
    rng = np.random.default_rng(23 + stage)
    n = 80000
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    hi = builder.build_host_index(codes, device="cpu")
    ix = s3dp.Index(hi)
    L = 100

    def damage(seg_start, kind):
        if kind == 0:
            s = codes[seg_start:seg_start + L].copy()
            for q in rng.choice(np.arange(3, L - 3), size=6, replace=False):
                s[q] = (s[q] + 1 + rng.integers(0, 3)) % 4
            return s, seg_start, (L - 6) - 12, 6
        if kind == 1:
            c = int(rng.integers(35, 65))
            return np.concatenate([codes[seg_start:seg_start + c], codes[seg_start + c + 3:seg_start + L + 3]]), seg_start, L - 5, 3
        c = int(rng.integers(35, 65))
        s = np.concatenate([codes[seg_start:seg_start + c], (codes[seg_start + c:seg_start + c + 2] + 2) % 4, codes[seg_start + c:seg_start + L - 2]])
        return s, seg_start, (L - 2) - 4, 2

    reads, truth = [], []
    for p in range(90):
        ins = int(rng.integers(260, 470))
        st = int(rng.integers(500, n - 1500))
        a, a_pos, a_score, a_ed = damage(st, p % 3)
        b, b_pos, b_score, b_ed = damage(st + ins - L, (p + 1) % 3)
        first, second = (a, COMP[b[::-1]]) if p % 2 == 0 else (COMP[b[::-1]], a)
        reads += [bytes(b"ACGT"[c] for c in first), bytes(b"ACGT"[c] for c in second)]
        # (forward sequence, true position, best score, edits, strand) of read 1 and read 2 of the pair
        ta, tb = (a, a_pos, a_score, a_ed, 1), (b, b_pos, b_score, b_ed, 2)
        truth.append((ta, tb) if p % 2 == 0 else (tb, ta))
    q, lens, wpq = s3dp.pack_reads(reads, 120)
    b = s3dp.Batch(ix, len(reads), wpq)
    b.upload(q, lens)
    sb = s3dp.Batch(ix, len(reads) * 16, 8)
    dp = s3dp.DpAligner(4096, 104, 104 + 2 * 25 + 8, scheme=1)
    res, cigars, too_many = dp.deep(b, sb, lens, np.arange(0, len(reads), 2), stage, max_hit, 200, 500, (30, 30), 49, 49)
    assert (too_many & 1).sum() == 0 and (too_many & 2).sum() > 0
    best = {}
    for r, cg in zip(res, cigars):
        k = int(r["readID"]) // 2
        tot = int(r["score_1"]) + int(r["score_2"])
        if k not in best or tot > best[k][2]:
            best[k] = (r, cg, tot)
    ok = 0
    for k, (t1, t2) in enumerate(truth):
        if k not in best:
            continue
        r, cg, _ = best[k]
        good = True
        for end, t in ((1, t1), (2, t2)):
            fwd, pos, score, edits, strand = t
            got_pos, got_score, got_ed, got_strand = int(r["algnmt_%d" % end]), int(r["score_%d" % end]), int(r["editdist_%d" % end]), int(r["strand_%d" % end])
            s, e, span = apply_cigar(cg[end - 1], fwd, codes, got_pos)
            assert s == got_score and e == got_ed, (k, end, cg)
            good &= got_strand == strand and abs(got_pos - pos) <= 4 and got_score >= score
        lo = min(int(r["algnmt_1"]), int(r["algnmt_2"]))
        assert 200 <= int(r["insertSize"]) <= 520
        ok += good
    assert ok >= 0.8 * len(truth), ok
    dp.close()
    b.close()
    sb.close()


def test_new_default_dp_pairs_given_hits_with_seeded_mates():
    """s3_dp_new_default: the aligned end's hits are taken as positions (70 copies of a repeat), only the mate is seeded; the deep DP's pairing keeps the
    copy next to the mate and both ends come back aligned by the DP with the mate's deletion in its CIGAR"""
    import re
    rng = np.random.default_rng(9)
    n, L, copies = 120000, 100, 70
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    rep = rng.integers(0, 4, size=L, dtype=np.uint8)
    starts = [800 + 1600 * k for k in range(copies)]
    for st in starts:
        codes[st:st + L] = rep
    hi = builder.build_host_index(codes, device="cpu")
    ix = s3dp.Index(hi)
    reads, truth = [], []
    for k in range(0, copies, 5):
        st, m = starts[k], starts[k] + 310
        c = int(rng.integers(30, 70))
        mate = np.concatenate([codes[m:m + c], codes[m + c + 3:m + L + 3]])
        reads.append(bytes(b"ACGT"[x] for x in rep))
        reads.append(bytes(b"ACGT"[x] for x in COMP[mate[::-1]]))
        truth.append((st, m))
    q, lens, wpq = s3dp.pack_reads(reads, 100)
    b = s3dp.Batch(ix, len(reads), wpq)
    b.upload(q, lens)
    sb = s3dp.Batch(ix, len(reads) * 16, 8)
    occ = np.zeros(len(truth) * copies, dtype=s3dp.OCC_DTYPE)
    for p in range(len(truth)):
        for j, st in enumerate(starts):
            o = occ[p * copies + j]
            o["readID"], o["pos"], o["strand"], o["mismatchCount"] = 2 * p, st, 1, 0
    dp = s3dp.DpAligner(4096, 104, 104 + 2 * 25 + 8, scheme=1)
    res, cigars, pairs, flags = dp.new_default(b, sb, lens, occ, 100, 100, 500, (30, 30), 49, 49)
    assert list(pairs) == [2 * p for p in range(len(truth))]
    assert (flags & 2).all()                                  # every pair had candidates
    ok = 0
    for p, (st, m) in enumerate(truth):
        mine = [(r, cg) for r, cg in zip(res, cigars) if int(r["readID"]) == 2 * p]
        assert mine, p
        best = max(mine, key=lambda x: int(x[0]["score_1"]) + int(x[0]["score_2"]))
        r, (c1, c2) = best
        ok += bool(int(r["algnmt_1"]) == st and abs(int(r["algnmt_2"]) - m) <= 3 and r["strand_1"] == 1 and r["strand_2"] == 2 and
                   re.sub(r"\d+", "", c1) == "M" and re.sub(r"\d+", "", c2) == "MDM")
    assert ok == len(truth), (ok, len(truth))
    dp.close(); b.close(); sb.close(); ix.close()
