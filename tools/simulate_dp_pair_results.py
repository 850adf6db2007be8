#!/usr/bin/env python3
"""Read pairs plus SYNTHETIC paired DP results for the output-stage fixtures (tests/golden/dp2).

Reads are cut from an index's own packed text and damaged in a known way.  The result lines state the alignments that damage implies, in
the DP engines' CIGAR dialect ('m' = mismatch run):
  D readID which pos1 strand1 score1 pos2 strand2 score2 editdist insert same cigar              default DP: one end from the BWT stage
  P readID pos1 strand1 score1 edit1 same1 cigar1 pos2 strand2 score2 edit2 same2 cigar2 insert  deep DP: both ends from the DP
  U readID fromDP pos strand score editdist cigar      a hit of ONE read of a pair no stage could pair (fromDP 0: carried over from the BWT stage;
                                                       pos 4294967295: the read has no hit)
and a stats file (readID x0 x1 mismatch) what the BWT stage would have left for the reads.  Fabricated extra results per pair (equal,
weaker, other positions) walk the X0 / X1 / XA:Z / MAPQ branches; some pairs cross a chromosome end or read through each other.  Only the
SAM formatting is under test with these vectors."""
import argparse
import numpy as np

COMP = np.array([3, 2, 1, 0], dtype=np.uint8)
M, X, GO, GE = 1, -2, -3, -1


def mcigar(pos, L):
    cigar, prev, i = "", 0, 0
    while i < len(pos):
        j = i
        while j + 1 < len(pos) and pos[j + 1] == pos[j] + 1:
            j += 1
        if pos[i] > prev:
            cigar += "%dM" % (pos[i] - prev)
        cigar += "%dm" % (j - i + 1)
        prev = pos[j] + 1
        i = j + 1
    if prev < L:
        cigar += "%dM" % (L - prev)
    return cigar


def dp_end(rng, text, p, L, kind):
    """a read end with gaps / clips at text position p: (forward bases, position, cigar, score, edit distance)"""
    if kind == 0:
        k, c = int(rng.integers(1, 6)), int(rng.integers(20, L - 20))
        return np.concatenate([text[p:p + c], text[p + c + k:p + L + k]]), p, "%dM%dD%dM" % (c, k, L - c), L * M + GO + (k - 1) * GE, k
    if kind == 1:
        k, c = int(rng.integers(1, 5)), int(rng.integers(20, L - 20))
        return (np.concatenate([text[p:p + c], rng.integers(0, 4, size=k, dtype=np.uint8), text[p + c:p + L - k]]), p, "%dM%dI%dM" % (c, k, L - c - k),
                (L - k) * M + GO + (k - 1) * GE, k)
    if kind == 2:
        fwd = text[p:p + L].copy()
        pos = sorted(set(int(x) for x in rng.choice(np.arange(2, L - 2), size=4, replace=False)) | {40, 41})
        for q in pos:
            fwd[q] = (fwd[q] + 1 + rng.integers(0, 3)) % 4
        return fwd, p, mcigar(pos, L), (L - len(pos)) * M + len(pos) * X, len(pos)
    if kind == 3:
        k = int(rng.integers(3, 20))
        return np.concatenate([rng.integers(0, 4, size=k, dtype=np.uint8), text[p + k:p + L]]), p + k, "%dS%dM" % (k, L - k), (L - k) * M, 0
    if kind == 4:
        k = int(rng.integers(3, 20))
        fwd = np.concatenate([text[p:p + L - k], rng.integers(0, 4, size=k, dtype=np.uint8)])
        fwd[10] = (fwd[10] + 1) % 4
        return fwd, p, "10M1m%dM%dS" % (L - k - 11, k), (L - k - 1) * M + X, 1
    return text[p:p + L].copy(), p, "%dM" % L, L * M, 0


def bwt_end(rng, text, p, L, mm):
    fwd = text[p:p + L].copy()
    for q in rng.choice(np.arange(1, L - 1), size=mm, replace=False):
        fwd[q] = (fwd[q] + 1 + rng.integers(0, 3)) % 4
    return fwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--index", required=True)
    ap.add_argument("--pairs", type=int, default=480)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--unpair", type=int, default=240, help="more pairs, for the leftover records (U lines)")
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--fastq1", required=True)
    ap.add_argument("--fastq2", required=True)
    ap.add_argument("--results", required=True)
    ap.add_argument("--stats", required=True)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    raw = np.fromfile(a.index + ".pac", dtype=np.uint8)
    n = (len(raw) - 2) * 4 + int(raw[-1])
    text = np.zeros(len(raw) * 4, dtype=np.uint8)
    for k in range(4):
        text[k::4] = (raw >> (6 - 2 * k)) & 3
    text = text[:n]
    ann = open(a.index + ".ann").read().split("\n")
    nseq = int(ann[0].split()[1])
    bounds = []
    for i in range(nseq):
        st, ln, _ = ann[2 + 2 * i].split()
        bounds.append((int(st), int(st) + int(ln)))
    L = a.read_len
    ends, dlines, plines, stats = [[], []], [], [], []
    half = a.pairs // 2
    for pr in range(a.pairs):
        rid = 2 * pr
        scen = pr % 12
        lo, hi = bounds[int(rng.integers(0, nseq - 1 if scen in (7, 8) else nseq))]      # boundary cases need text behind the chromosome
        insert = int(rng.integers(250, 500))
        left = int(rng.integers(lo + 10, hi - insert - 2 * L - 10))
        right = left + insert - L
        firstIsLeft = bool(rng.integers(0, 2))          # the '+' (left) end is read 1, or read 2
        if scen == 7:                                     # the right end crosses the end of its chromosome
            right = hi - int(rng.integers(30, 80))
            left = right - insert + L
        if scen == 8:                                     # the left end starts before the chromosome's start is impossible: let the LEFT end cross instead
            left = hi - int(rng.integers(30, 80))
            right = left + int(rng.integers(5, 20))
        if scen == 9:                                     # read-through: the right end starts left of the left end
            right = left - int(rng.integers(1, 30))
        if pr < half:
            # ---------------- default DP: one end from the BWT stage ----------------
            dpIsFirst = bool(rng.integers(0, 2))
            dpIsLeft = dpIsFirst == firstIsLeft
            mm = int(rng.integers(0, 4))
            kind = int(rng.integers(0, 6))
            if dpIsLeft:
                dfwd, dpos, cig, score, ed = dp_end(rng, text, left, L, kind)
                bfwd, bpos = bwt_end(rng, text, right, L, mm), right
            else:
                dfwd, dpos, cig, score, ed = dp_end(rng, text, right, L, kind)
                bfwd, bpos = bwt_end(rng, text, left, L, mm), left
            dstrand, bstrand = (1, 2) if dpIsLeft else (2, 1)
            seq = {True: dfwd if dstrand == 1 else COMP[dfwd[::-1]], False: bfwd if bstrand == 1 else COMP[bfwd[::-1]]}
            ends[0].append(seq[dpIsFirst]); ends[1].append(seq[not dpIsFirst])
            which = 0 if dpIsFirst else 1

            def dline(dp, ds, dsc, de, dc, bp, bs, bm, same=1, w=which):
                ins = abs((dp if ds == 2 else bp) + L - (bp if ds == 2 else dp))
                if w == 0:
                    return "D %d %d %d %d %d %d %d %d %d %d %d %s" % (rid, w, dp, ds, dsc, bp, bs, bm, de, ins, same, dc)
                return "D %d %d %d %d %d %d %d %d %d %d %d %s" % (rid, w, bp, bs, bm, dp, ds, dsc, de, ins, same, dc)
            group = [dline(dpos, dstrand, score, ed, cig, bpos, bstrand, mm, same=1 + int(scen == 5))]
            if scen in (1, 2, 3, 4, 5, 6):               # more results of the same kind for this pair
                for t in range(int(rng.integers(1, 5))):
                    q = int(rng.integers(lo + 10, hi - 2 * L - 10))
                    sc = score if (scen == 1 and t == 0) else score - int(rng.integers(0, 9))
                    bm = mm if scen in (1, 2) else mm + int(rng.integers(0, 3))
                    bp2 = bpos if scen != 3 else int(rng.integers(lo + 10, hi - 2 * L - 10))
                    cg = cig if sc == score else "50M1m%dM" % (L - 51)
                    group.append(dline(q, dstrand, sc, ed + (score - sc) // 3, cg, bp2, bstrand, bm, same=1 + int(t == 2)))
                if scen == 6:                             # the other end also went through the DP (both ends had hits, no proper pairing)
                    group.append(dline(bpos, bstrand, L - 4, 2, "30M2m%dM" % (L - 32), dpos, dstrand, 1, w=1 - which))
                order = rng.permutation(len(group))
                group = [group[i] for i in order]
            dlines += group
            # BWT-stage statistics: the BWT end has them; sometimes the DP end as well (the pair had hits on both ends)
            bread, dread = (rid + 1, rid) if dpIsFirst else (rid, rid + 1)
            stats.append("%d %d %d %d" % (bread, 1 + int(rng.integers(0, 3)) * int(scen % 3 == 0), int(rng.integers(0, 4)), mm))
            if scen in (6, 10):
                stats.append("%d %d %d %d" % (dread, 1 + int(rng.integers(0, 2)), int(rng.integers(0, 3)), int(rng.integers(0, 4))))
        else:
            # ---------------- deep DP: both ends from the DP ----------------
            k1, k2 = int(rng.integers(0, 6)), int(rng.integers(0, 6))
            lf, lp, lc, ls, le = dp_end(rng, text, left, L, k1)
            rf, rp, rc, rs_, re = dp_end(rng, text, right, L, k2)
            ends[0 if firstIsLeft else 1].append(lf)
            ends[1 if firstIsLeft else 0].append(COMP[rf[::-1]])

            def pline(lp, ls, le, lc, rp, rs_, re, rc, s1=1, s2=1):
                ins = abs(rp + L - lp)
                if firstIsLeft:
                    return "P %d %d 1 %d %d %d %s %d 2 %d %d %d %s %d" % (rid, lp, ls, le, s1, lc, rp, rs_, re, s2, rc, ins)
                return "P %d %d 2 %d %d %d %s %d 1 %d %d %d %s %d" % (rid, rp, rs_, re, s2, rc, lp, ls, le, s1, lc, ins)
            group = [pline(lp, ls, le, lc, rp, rs_, re, rc, s1=1 + int(scen == 5))]
            if scen in (1, 2, 3, 4, 5, 6):
                for t in range(int(rng.integers(1, 5))):
                    q1 = lp if scen == 3 else int(rng.integers(lo + 10, hi - 2 * L - 10))
                    q2 = int(rng.integers(lo + 10, hi - 2 * L - 10))
                    d1 = 0 if (scen == 1 and t == 0) else int(rng.integers(0, 7))
                    d2 = 0 if (scen == 1 and t == 0) else int(rng.integers(0, 7))
                    group.append(pline(q1, ls - d1, le + d1 // 3, lc if d1 == 0 else "20M1m%dM" % (L - 21), q2, rs_ - d2, re + d2 // 3,
                                       rc if d2 == 0 else "%dM2I10M" % (L - 12), s1=1 + int(t == 1), s2=1 + int(t == 2)))
                order = rng.permutation(len(group))
                group = [group[i] for i in order]
            plines += group
            if scen in (2, 6, 10, 11):
                stats.append("%d %d %d %d" % (rid, 1 + int(rng.integers(0, 3)), int(rng.integers(0, 3)), int(rng.integers(0, 5))))
                stats.append("%d %d %d %d" % (rid + 1, 1 + int(rng.integers(0, 3)), int(rng.integers(0, 3)), int(rng.integers(0, 5))))
    # ---------------- leftover pairs: each end on its own ----------------
    ulines = []
    for pr in range(a.pairs, a.pairs + a.unpair):
        for e in range(2):
            rid = 2 * pr + e
            scen = (pr + 5 * e) % 8
            lo, hi = bounds[int(rng.integers(0, nseq - 1 if scen == 6 else nseq))]
            p = int(rng.integers(lo + 10, hi - 3 * L - 10))
            strand = 1 + int(rng.integers(0, 2))
            if scen == 7:                                     # no hit at all
                fwd = rng.integers(0, 4, size=L, dtype=np.uint8)
                ulines.append("U %d 1 4294967295 1 0 0 *" % rid)
            elif scen in (4, 5):                              # hits carried over from the BWT stage
                mm = int(rng.integers(0, 4))
                fwd = bwt_end(rng, text, p, L, mm)
                grp = ["U %d 0 %d %d %d %d %dM" % (rid, p, strand, L * M + mm * X, mm, L)]
                for t in range(int(rng.integers(0, 4)) if scen == 5 else 0):
                    m2 = mm + int(rng.integers(0, 3)) * int(t > 0)
                    grp.append("U %d 0 %d %d %d %d %dM" % (rid, int(rng.integers(lo + 10, hi - 2 * L)), 1 + int(rng.integers(0, 2)), L * M + m2 * X, m2, L))
                ulines += [grp[i] for i in rng.permutation(len(grp))]
            else:                                             # hits of the single-end DP
                if scen == 6:
                    p = hi - int(rng.integers(30, 80))        # crosses the end of its chromosome
                fwd, q, cig, score, ed = dp_end(rng, text, p, L, int(rng.integers(0, 6)))
                grp = ["U %d 1 %d %d %d %d %s" % (rid, q, strand, score, ed, cig)]
                for t in range(int(rng.integers(1, 5)) if scen in (1, 2, 3) else 0):
                    sc = score if (scen == 1 and t == 0) else score - int(rng.integers(1, 40))
                    grp.append("U %d 1 %d %d %d %d %s" % (rid, int(rng.integers(lo + 10, hi - 2 * L)), 1 + int(rng.integers(0, 2)), sc, ed + (score - sc) // 3,
                                                          cig if sc == score else "48M1m%dM" % (L - 49)))
                ulines += [grp[i] for i in rng.permutation(len(grp))]
            ends[e].append(fwd if strand == 1 else COMP[fwd[::-1]])
    for k, name in ((0, a.fastq1), (1, a.fastq2)):
        with open(name, "w") as f:
            for i, s in enumerate(ends[k]):
                qual = "".join(chr(33 + int(x)) for x in rng.integers(5, 41, size=len(s)))
                f.write("@q%d/%d\n%s\n+\n%s\n" % (i, k + 1, bytes(b"ACGT"[c] for c in s).decode(), qual))
    with open(a.results, "w") as f:
        f.write("\n".join(dlines + plines + ulines) + "\n")
    with open(a.stats, "w") as f:
        f.write("\n".join(stats) + "\n")


if __name__ == "__main__":
    main()
