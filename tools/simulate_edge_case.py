#!/usr/bin/env python3
"""Tiny genome + reads that hug or overhang the ends of the text (fixture tests/golden/g6): the 13-mers the lookup tables count for the last / first
text positions (padded with A), positions before the text start, the text's own runs of A at both ends, a repeat, one run of N in the middle (the reference
tools need an ambiguity segment).  Two sequences, so chromosome boundaries are in play too."""
import argparse
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fasta", required=True)
    ap.add_argument("--fastq", required=True)
    ap.add_argument("--seed", type=int, default=6)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    L = 100
    n1, n2 = 2600, 2200
    c1 = rng.integers(0, 4, size=n1, dtype=np.uint8)
    c2 = rng.integers(0, 4, size=n2, dtype=np.uint8)
    c1[:5] = 0                                               # the text starts and ends with A's
    c2[n2 - 9:] = 0
    c1[700:900] = c1[100:300]                                # a repeat
    s1 = "".join("ACGT"[x] for x in c1)
    s1 = s1[:1500] + "N" * 30 + s1[1530:]
    s2 = "".join("ACGT"[x] for x in c2)
    with open(a.fasta, "w") as f:
        for name, s in (("chr1", s1), ("chr2", s2)):
            f.write(">%s\n" % name)
            for p in range(0, len(s), 60):
                f.write(s[p:p + 60] + "\n")
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    reads = []
    for k in range(900):
        kind = k % 7
        if kind == 0:                                         # inside, near the very end of the text
            st = int(rng.integers(n2 - 140, n2 - L + 1)); seg = s2[st:st + L]
        elif kind == 1:                                       # overhangs the end: the rest is A
            st = int(rng.integers(n2 - 95, n2 - 5)); seg = s2[st:] + "A" * (L - (n2 - st))
        elif kind == 2:                                       # overhangs the start
            ov = int(rng.integers(1, 60)); seg = "A" * ov + s1[:L - ov]
        elif kind == 3:                                       # near the start
            st = int(rng.integers(0, 40)); seg = s1[st:st + L]
        elif kind == 4:                                       # from the repeat
            st = int(rng.integers(100, 200)); seg = s1[st:st + L]
        elif kind == 5:                                       # across the boundary of the two sequences
            st = int(rng.integers(n1 - 90, n1 - 10)); seg = (s1 + s2)[st:st + L]
        else:
            st = int(rng.integers(1600, n1 - L)); seg = s1[st:st + L]
        seg = list(seg.replace("N", "G"))
        for q in rng.choice(L, size=int(rng.integers(0, 4)), replace=False):
            seg[q] = "ACGT"[("ACGT".index(seg[q]) + 1 + int(rng.integers(0, 3))) % 4]
        seg = "".join(seg)
        if rng.integers(0, 2):
            seg = "".join(comp[c] for c in reversed(seg))
        reads.append(seg)
    with open(a.fastq, "w") as f:
        for i, s in enumerate(reads):
            f.write("@e%d\n%s\n+\n%s\n" % (i, s, "".join(chr(33 + int(x)) for x in rng.integers(5, 41, size=L))))


if __name__ == "__main__":
    main()
