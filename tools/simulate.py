#!/usr/bin/env python3
"""Synthetic genomes and reads for tests / fixtures / bench (SURVEY.md 8d recipe).

genome: uniform ACGT, several chromosomes, a few N-runs (>=10 N are removed from the
        indexed text by the index builder, shorter runs become G), optional planted repeats.
reads : uniform start, strand Bernoulli(0.5), substitutions per read drawn from a
        categorical mix, constant quality 'I', names r<idx>.
"""
import argparse
import numpy as np

ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    COMP[a] = b


def make_genome(total_len, n_chrom=3, n_runs=4, seed=1, repeats=0, repeat_len=300):
    rng = np.random.default_rng(seed)
    lens = np.full(n_chrom, total_len // n_chrom)
    lens[-1] += total_len - lens.sum()
    chroms = []
    for ci, L in enumerate(lens):
        s = ALPHA[rng.integers(0, 4, size=L)].copy()
        chroms.append(s)
    # planted repeats: copy a segment to other places (with a few substitutions)
    for _ in range(repeats):
        ci = rng.integers(0, n_chrom)
        src = rng.integers(0, len(chroms[ci]) - repeat_len)
        seg = chroms[ci][src:src + repeat_len].copy()
        for _c in range(rng.integers(2, 40)):
            cj = rng.integers(0, n_chrom)
            dst = rng.integers(0, len(chroms[cj]) - repeat_len)
            seg2 = seg.copy()
            nsub = rng.integers(0, 3)
            for p in rng.integers(0, repeat_len, size=nsub):
                seg2[p] = ALPHA[rng.integers(0, 4)]
            chroms[cj][dst:dst + repeat_len] = seg2
    # N runs: half long (>=10), half short (<10)
    for k in range(n_runs):
        ci = rng.integers(0, n_chrom)
        rl = int(rng.integers(10, 200)) if k % 2 == 0 else int(rng.integers(1, 10))
        st = rng.integers(100, len(chroms[ci]) - rl - 100)
        chroms[ci][st:st + rl] = ord("N")
    return chroms


def write_fasta(path, chroms, width=60):
    with open(path, "wb") as f:
        for i, s in enumerate(chroms):
            f.write(b">chr%d synthetic\n" % (i + 1))
            b = s.tobytes()
            for o in range(0, len(b), width):
                f.write(b[o:o + width])
                f.write(b"\n")


def make_reads(chroms, n, L, seed=2026, mix=(0.55, 0.25, 0.12, 0.05, 0.03), var_len=None):
    rng = np.random.default_rng(seed)
    clen = np.array([len(c) for c in chroms])
    out = []
    truth = []
    i = 0
    while i < n:
        ci = rng.choice(len(chroms), p=clen / clen.sum())
        rl = L if var_len is None else int(rng.integers(var_len[0], var_len[1] + 1))
        st = int(rng.integers(0, clen[ci] - rl))
        seg = chroms[ci][st:st + rl].copy()
        if (seg == ord("N")).any():
            continue
        nsub = int(rng.choice(len(mix), p=np.array(mix) / sum(mix)))
        pos = rng.choice(rl, size=nsub, replace=False)
        for p in pos:
            seg[p] = ALPHA[(np.searchsorted(ALPHA, seg[p]) + rng.integers(1, 4)) % 4]
        strand = int(rng.integers(0, 2))
        if strand:
            seg = COMP[seg[::-1]]
        out.append(seg.tobytes())
        truth.append((ci, st, strand, nsub))
        i += 1
    return out, truth


def write_fastq(path, reads, prefix="r"):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@%s%d\n" % (prefix.encode(), i))
            f.write(r)
            f.write(b"\n+\n")
            f.write(b"I" * len(r))
            f.write(b"\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-len", type=int, default=200000)
    ap.add_argument("--chroms", type=int, default=3)
    ap.add_argument("--n-runs", type=int, default=4)
    ap.add_argument("--repeats", type=int, default=0)
    ap.add_argument("--genome-seed", type=int, default=1)
    ap.add_argument("--fasta", required=True)
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--var-len", type=int, nargs=2, default=None)
    ap.add_argument("--read-seed", type=int, default=2026)
    ap.add_argument("--fastq", default=None)
    ap.add_argument("--truth", default=None)
    ap.add_argument("--mix", type=float, nargs="+", default=None, help="weights of 0,1,2,... substitutions per read")
    a = ap.parse_args()
    chroms = make_genome(a.genome_len, a.chroms, a.n_runs, a.genome_seed, a.repeats)
    write_fasta(a.fasta, chroms)
    if a.reads:
        reads, truth = make_reads(chroms, a.reads, a.read_len, a.read_seed, var_len=a.var_len, **({"mix": tuple(a.mix)} if a.mix else {}))
        write_fastq(a.fastq, reads)
        if a.truth:
            with open(a.truth, "w") as f:
                for t in truth:
                    f.write("%d %d %d %d\n" % t)


if __name__ == "__main__":
    main()
