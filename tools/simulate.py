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


def make_genome(total_len, n_chrom=3, n_runs=4, seed=1, repeats=0, repeat_len=300, family_copies=0, family_len=600, family_div=0.02):
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
    # one large family (own generator: genomes made without it stay what they were): family_copies copies of a random unit, every base of a copy replaced
    # with probability family_div -- reads inside it come with hundreds of hits (the hit limits of the DP stages, the new default DP, long merge runs)
    if family_copies:
        frng = np.random.default_rng(seed + 7777)
        unit = ALPHA[frng.integers(0, 4, size=family_len)].copy()
        for _c in range(family_copies):
            cj = frng.integers(0, n_chrom)
            dst = frng.integers(0, len(chroms[cj]) - family_len)
            cp = unit.copy()
            hit = frng.random(family_len) < family_div
            cp[hit] = ALPHA[frng.integers(0, 4, size=int(hit.sum()))]
            chroms[cj][dst:dst + family_len] = cp
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


def make_pairs(chroms, n, L, seed=2026, mix=(0.55, 0.25, 0.12, 0.05, 0.03), ins_mean=350, ins_sd=40, ins_max=500,
               frac_half=0.08, frac_none=0.04, frac_far=0.08, var_len=None, frac_indel=0.0, frac_damaged=0.0):
    """read pairs in +/- orientation: a fragment of `insert` bases, mate A = its first L bases, mate B = reverse complement of its last
    L bases; which of the two is written to file 1 is random.  A fraction has one or both mates replaced by random sequence
    (half-aligned / unaligned pairs) or the second mate taken from a distant place (both align, no proper pair).
    frac_indel: one mate of an otherwise proper pair carries a 1-8 base insertion or deletion (-> default DP); frac_damaged: both mates
    carry an indel or 5-8 substitutions (-> deep DP).  Both are 0 by default and then draw nothing from the generator."""
    rng = np.random.default_rng(seed)
    clen = np.array([len(c) for c in chroms])
    r1, r2, truth = [], [], []

    def mutate(seg):
        nsub = int(rng.choice(len(mix), p=np.array(mix) / sum(mix)))
        for p in rng.choice(len(seg), size=nsub, replace=False):
            seg[p] = ALPHA[(np.searchsorted(ALPHA, seg[p]) + rng.integers(1, 4)) % 4]
        return seg, nsub

    while len(r1) < n:
        ci = int(rng.choice(len(chroms), p=clen / clen.sum()))
        la = L if var_len is None else int(rng.integers(var_len[0], var_len[1] + 1))
        lb = L if var_len is None else int(rng.integers(var_len[0], var_len[1] + 1))
        ins = int(np.clip(round(rng.normal(ins_mean, ins_sd)), max(la, lb) + 1, ins_max))
        st = int(rng.integers(0, clen[ci] - ins - 4000))
        frag = chroms[ci][st:st + ins]
        kind = rng.random()
        a = frag[:la].copy()
        b = frag[ins - lb:].copy()
        if kind < frac_far:                                   # second mate from 1.5-3.5 kbp further on: both align, insert too large
            off = int(rng.integers(1500, 3500))
            b = chroms[ci][st + off:st + off + lb].copy()
        if (a == ord("N")).any() or (b == ord("N")).any():
            continue
        a, na = mutate(a)
        b, nb = mutate(b)
        if frac_indel > 0 or frac_damaged > 0:
            u = rng.random()

            def damage(which, heavy):
                # which 0: mate A, anchored at its left end; 1: mate B, anchored at its right end (both still in fragment orientation)
                L0 = la if which == 0 else lb
                if heavy and rng.random() < 0.5:
                    seg = (a if which == 0 else b).copy()
                    for p in rng.choice(L0, size=int(rng.integers(5, 9)), replace=False):
                        seg[p] = ALPHA[(np.searchsorted(ALPHA, seg[p]) + rng.integers(1, 4)) % 4]
                    return seg
                k = int(rng.integers(1, 9))
                p = int(rng.integers(12, L0 - 12))
                src = chroms[ci][st:st + L0 + 16].copy() if which == 0 else chroms[ci][st + ins - L0 - 16:st + ins].copy()
                if (src == ord("N")).any() or len(src) < L0 + 16:
                    return (a if which == 0 else b)
                if which == 1:
                    p += 16
                if rng.random() < 0.5:
                    seq = np.concatenate([src[:p], src[p + k:]])
                else:
                    seq = np.concatenate([src[:p], ALPHA[rng.integers(0, 4, size=k)], src[p:]])
                return seq[:L0].copy() if which == 0 else seq[len(seq) - L0:].copy()
            if kind >= frac_far + frac_half + frac_none:
                if u < frac_indel:
                    if rng.random() < 0.5:
                        a = damage(0, False)
                    else:
                        b = damage(1, False)
                elif u < frac_indel + frac_damaged:
                    a = damage(0, True)
                    b = damage(1, True)
        b = COMP[b[::-1]]
        if frac_far <= kind < frac_far + frac_half:           # one mate unalignable
            if rng.random() < 0.5:
                a = ALPHA[rng.integers(0, 4, size=la)]
            else:
                b = ALPHA[rng.integers(0, 4, size=lb)]
        elif frac_far + frac_half <= kind < frac_far + frac_half + frac_none:
            a = ALPHA[rng.integers(0, 4, size=la)]
            b = ALPHA[rng.integers(0, 4, size=lb)]
        swap = int(rng.integers(0, 2))
        if swap:
            a, b = b, a
        r1.append(a.tobytes())
        r2.append(b.tobytes())
        truth.append((ci, st, ins, swap))
    return r1, r2, truth


def write_fastq(path, reads, prefix="r", suffix=""):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@%s%d%s\n" % (prefix.encode(), i, suffix.encode()))
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
    ap.add_argument("--family-copies", type=int, default=0, help="plant one large repeat family with this many copies")
    ap.add_argument("--family-len", type=int, default=600)
    ap.add_argument("--family-div", type=float, default=0.02, help="per-base divergence of a copy from the family's unit")
    ap.add_argument("--fasta", required=True)
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--var-len", type=int, nargs=2, default=None)
    ap.add_argument("--read-seed", type=int, default=2026)
    ap.add_argument("--fastq", default=None)
    ap.add_argument("--truth", default=None)
    ap.add_argument("--mix", type=float, nargs="+", default=None, help="weights of 0,1,2,... substitutions per read")
    ap.add_argument("--pairs", type=int, default=0, help="number of read pairs (written to --fastq and --fastq2)")
    ap.add_argument("--fastq2", default=None)
    ap.add_argument("--insert-mean", type=float, default=350)
    ap.add_argument("--insert-sd", type=float, default=40)
    ap.add_argument("--indel-frac", type=float, default=0.0, help="pairs: share with a 1-8 base indel in one mate")
    ap.add_argument("--damaged-frac", type=float, default=0.0, help="pairs: share with an indel or 5-8 substitutions in both mates")
    a = ap.parse_args()
    chroms = make_genome(a.genome_len, a.chroms, a.n_runs, a.genome_seed, a.repeats, family_copies=a.family_copies, family_len=a.family_len, family_div=a.family_div)
    write_fasta(a.fasta, chroms)
    if a.reads:
        reads, truth = make_reads(chroms, a.reads, a.read_len, a.read_seed, var_len=a.var_len, **({"mix": tuple(a.mix)} if a.mix else {}))
        write_fastq(a.fastq, reads)
        if a.truth:
            with open(a.truth, "w") as f:
                for t in truth:
                    f.write("%d %d %d %d\n" % t)
    if a.pairs:
        kw = {"mix": tuple(a.mix)} if a.mix else {}
        r1, r2, truth = make_pairs(chroms, a.pairs, a.read_len, a.read_seed, ins_mean=a.insert_mean, ins_sd=a.insert_sd, var_len=a.var_len, frac_indel=a.indel_frac,
                                  frac_damaged=a.damaged_frac, **kw)
        write_fastq(a.fastq, r1, prefix="p", suffix="/1")
        write_fastq(a.fastq2, r2, prefix="p", suffix="/2")
        if a.truth:
            with open(a.truth, "w") as f:
                for t in truth:
                    f.write("%d %d %d %d\n" % t)


if __name__ == "__main__":
    main()
