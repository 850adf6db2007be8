#!/usr/bin/env python3
"""Reads plus SYNTHETIC single-read DP results for the output-stage fixtures (tests/golden/dp1).

The reads are cut from the index's own packed text (so positions are index coordinates) and damaged in a known way; the DP result
lines (readID strand algnmt score editdist num_sameScore cigar) state the alignment that damage implies, in the DP engine's CIGAR
dialect (mismatch runs as 'm').  Some reads get extra, fabricated lower-scoring or equal results (XA / X0 / X1 / MAPQ branches), some
cross a chromosome end (boundary trimming), some stay unaligned.  Only the SAM formatting is under test with these vectors."""
import argparse
import numpy as np

COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--index", required=True)
    ap.add_argument("--reads", type=int, default=600)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--fastq", required=True)
    ap.add_argument("--results", required=True)
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
    M, X, GO, GE = 1, -2, -3, -1
    reads, lines = [], []
    for r in range(a.reads):
        lo, hi = bounds[int(rng.integers(0, nseq))]
        kind = r % 10
        p = int(rng.integers(lo + 10, hi - 2 * L - 10))
        strand = 1 + int(rng.integers(0, 2))
        extra = []
        if kind == 0:                                         # deletion of k bases
            k, c = int(rng.integers(1, 6)), int(rng.integers(20, 80))
            fwd = np.concatenate([text[p:p + c], text[p + c + k:p + L + k]])
            cigar, score, ed = "%dM%dD%dM" % (c, k, L - c), L * M + GO + (k - 1) * GE, k
        elif kind == 1:                                       # insertion of k bases
            k, c = int(rng.integers(1, 5)), int(rng.integers(20, 80))
            fwd = np.concatenate([text[p:p + c], rng.integers(0, 4, size=k, dtype=np.uint8), text[p + c:p + L - k]])
            cigar, score, ed = "%dM%dI%dM" % (c, k, L - c - k), (L - k) * M + GO + (k - 1) * GE, k
        elif kind in (2, 3):                                  # substitutions, some adjacent
            fwd = text[p:p + L].copy()
            pos = sorted(set(int(x) for x in rng.choice(np.arange(2, L - 2), size=5, replace=False)) | ({40, 41} if kind == 3 else set()))
            for q in pos:
                fwd[q] = (fwd[q] + 1 + rng.integers(0, 3)) % 4
            cigar, prev, run = "", 0, 0
            i = 0
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
            score, ed = (L - len(pos)) * M + len(pos) * X, len(pos)
        elif kind == 4:                                       # soft clip at the start (reference order)
            k = int(rng.integers(3, 20))
            fwd = np.concatenate([rng.integers(0, 4, size=k, dtype=np.uint8), text[p + k:p + L]])
            cigar, score, ed = "%dS%dM" % (k, L - k), (L - k) * M, 0
            p = p + k
        elif kind == 5:                                       # soft clip at the end + one mismatch
            k = int(rng.integers(3, 20))
            fwd = np.concatenate([text[p:p + L - k], rng.integers(0, 4, size=k, dtype=np.uint8)])
            fwd[10] = (fwd[10] + 1) % 4
            cigar, score, ed = "10M1m%dM%dS" % (L - k - 11, k), (L - k - 1) * M + X, 1
        elif kind == 6:                                       # exact, plus fabricated weaker / equal hits elsewhere
            fwd = text[p:p + L].copy()
            cigar, score, ed = "%dM" % L, L * M, 0
            for t in range(int(rng.integers(1, 5))):
                q = int(rng.integers(lo + 10, hi - 2 * L))
                sc = L - 3 * int(rng.integers(0, 12)) if t else (L if r % 20 == 6 else L - 9)
                extra.append((1 + int(rng.integers(0, 2)), q, sc, (L - sc) // 3, "50M1m49M" if sc < L else "%dM" % L))
        elif kind == 7:                                       # crosses the end of its chromosome
            p = hi - int(rng.integers(30, 80))
            fwd = text[p:p + L].copy()
            cigar, score, ed = "%dM" % L, L * M, 0
        elif kind == 8:                                       # deletion + mismatch + fabricated second hit with an insertion
            k, c = 2, 30
            fwd = np.concatenate([text[p:p + c], text[p + c + k:p + L + k]])
            fwd[70] = (fwd[70] + 2) % 4
            cigar, score, ed = "%dM%dD%dM1m%dM" % (c, k, 70 - c, L - 71), (L - 1) * M + X + GO + GE, 3
            extra.append((strand, int(rng.integers(lo + 10, hi - 2 * L)), score - 6, 5, "40M3I57M"))
        else:                                                 # stays unaligned
            fwd = rng.integers(0, 4, size=L, dtype=np.uint8)
            cigar = None
        seq = fwd if strand == 1 else COMP[fwd[::-1]]
        reads.append(bytes(b"ACGT"[c] for c in seq))
        if cigar is None:
            lines.append("%d 1 4294967295 0 0 0 *" % r)
        else:
            lines.append("%d %d %d %d %d 1 %s" % (r, strand, p, score, ed, cigar))
            for (s2, q, sc, e2, cg) in extra:
                lines.append("%d %d %d %d %d 1 %s" % (r, s2, q, sc, e2, cg))
    with open(a.fastq, "w") as f:
        for i, s in enumerate(reads):
            qual = "".join(chr(33 + int(x)) for x in rng.integers(5, 41, size=len(s)))
            f.write("@d%d\n%s\n+\n%s\n" % (i, s.decode(), qual))
    with open(a.results, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
