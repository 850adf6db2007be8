#!/usr/bin/env python3
"""Genome + read pairs that walk every branch of the paired DP routing (hostKernel with needProceedDP, CPUfunctions.cpp:2153-2262, 2444-2543):
a repeat family with more copies than the default DP takes (50), one with fewer, unique sequence; pairs whose mate is damaged / missing / far away.
Fixture tests/golden/g5."""
import argparse
import numpy as np

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def rc(s):
    return "".join(COMP[c] for c in reversed(s))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fasta", required=True)
    ap.add_argument("--fastq1", required=True)
    ap.add_argument("--fastq2", required=True)
    ap.add_argument("--seed", type=int, default=3)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    L = 100
    txt = lambda n: "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=n))
    chrom = [list(txt(70000)), list(txt(50000))]
    big, small = txt(L), txt(L)                       # 70 copies (some with one substitution), 20 copies
    bigAt, smallAt = [], []
    for k in range(70):
        c, st = k % 2, 500 + 650 * (k // 2) + (7 * k) % 40
        cp = list(big)
        if k % 5 == 4:
            q = int(rng.integers(10, 90)); cp[q] = "ACGT"[("ACGT".index(cp[q]) + 1) % 4]
        chrom[c][st:st + L] = cp
        bigAt.append((c, st, k % 5 == 4))
    for k in range(20):
        c, st = k % 2, 30000 + 900 * (k // 2)
        chrom[c][st:st + L] = list(small)
        smallAt.append((c, st))
    for c, at in ((0, 26000), (1, 27000)):           # a run of N in each sequence, as real assemblies have (and as every other fixture)
        chrom[c][at:at + 40] = "N" * 40
    chrom = ["".join(c) for c in chrom]
    with open(a.fasta, "w") as f:
        for i, c in enumerate(chrom):
            f.write(">chr%d\n" % (i + 1))
            for p in range(0, len(c), 60):
                f.write(c[p:p + 60] + "\n")

    def damaged(seg):
        c = int(rng.integers(30, 70))
        return seg[:c] + seg[c + 3:L + 3]
    pairs = []
    for k in range(240):
        kind = k % 8
        if kind in (0, 1, 2):                         # an end inside the big family: 0 mate damaged, 1 mate random, 2 mate exact but 4 kb away
            c, st, _ = bigAt[int(rng.integers(0, len(bigAt)))]
            left = chrom[c][st:st + L]
            m = st + (300 if kind != 2 else 4000)
            mate = chrom[c][m:m + L + 3]
            mate = damaged(mate) if kind == 0 else txt(L) if kind == 1 else mate[:L]
        elif kind in (3, 4):                          # the small family: 3 mate damaged, 4 mate exact but far
            c, st = smallAt[int(rng.integers(0, len(smallAt)))]
            left = chrom[c][st:st + L]
            m = st + (280 if kind == 3 else 5000)
            mate = chrom[c][m:m + L + 3]
            mate = damaged(mate) if kind == 3 else mate[:L]
        elif kind == 5:                               # unique end, damaged mate
            c = int(rng.integers(0, 2)); st = int(rng.integers(45000, len(chrom[c]) - 1000))
            left = chrom[c][st:st + L]
            mate = damaged(chrom[c][st + 320:st + 320 + L + 3])
        elif kind == 6:                               # both ends damaged
            c = int(rng.integers(0, 2)); st = int(rng.integers(45000, len(chrom[c]) - 1000))
            left = damaged(chrom[c][st:st + L + 3])
            mate = damaged(chrom[c][st + 320:st + 320 + L + 3])
        else:                                         # proper pair, one substitution
            c = int(rng.integers(0, 2)); st = int(rng.integers(45000, len(chrom[c]) - 1000))
            left = chrom[c][st:st + L]
            mate = chrom[c][st + 300:st + 300 + L]
        a_, b_ = left, rc(mate)
        if rng.integers(0, 2):
            a_, b_ = b_, a_
        pairs.append((a_, b_))
    for fn, idx in ((a.fastq1, 0), (a.fastq2, 1)):
        with open(fn, "w") as f:
            for i, pr in enumerate(pairs):
                q = "".join(chr(33 + int(x)) for x in rng.integers(10, 41, size=len(pr[idx])))
                f.write("@r%d/%d\n%s\n+\n%s\n" % (i, idx + 1, pr[idx], q))


if __name__ == "__main__":
    main()
