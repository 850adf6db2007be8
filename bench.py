#!/usr/bin/env python3
"""bench.py -- reads/s of the alignment hot path (BASELINE.json configs[1]): synthetic genome of GRCh38
scale with a full-SA 2BWT + GPU-2BWT index resident in HBM, synthetic 100 bp single-end reads,
<= 4 mismatches, all-best (soap3-dp single ... -s 4 -h 2 -b 2), DP off.

A "step" = one pass of the whole single-end ladder (phases 1..4 mismatches: BWT search with lookup-table
start and check-and-extend, ordering, per-read merge, SA decode) over one resident batch of reads.
One process per GPU; reads are sharded (weak scaling), the index is replicated per GPU; the only
collective is the final reduce of the counters.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

# the HIP runtime maps its streams onto four hardware queues unless told otherwise: batches in flight on their own streams then wait for each other
# where no dependency exists (libs3dp.so asks for sixteen itself; here torch may bring the runtime up first)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "soap3-dp_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
COLL_DEV = []               # [None] when the ranks share ONE GPU over gloo (S3DP_BENCH_SHARE_GPU=1, a test of the N > 1 path on a one-GPU box): collectives on CPU tensors


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# Repeat structure of a human-like genome (round-2 headline workload).  Shares are fractions of the genome at repeat_frac = 0.5 and
# scale with it.  GRCh38 is about half interspersed / tandem repeats: Alu ~ 1.1 M copies of a 300 bp unit (10 %), L1 ~ 0.5 M mostly
# 5'-truncated copies of a 6 kbp unit (17 %), many mid-size families, a few per cent of recent segmental duplications and satellites.
# Old copies sit 5-15 % from their consensus, young sub-families 0.5-2 % (those make reads that hit MaxOutputPerRead and multi-map).
REPEAT_FAMILIES = (
    # name, share, unit length, copy length as a fraction of the unit (lo, hi), young share, young divergence, old divergence
    ("sine_alu_like", 0.10, 300, (0.85, 1.0), 0.15, (0.005, 0.02), (0.05, 0.15)),
    ("line_l1_like", 0.17, 6000, (0.03, 1.0), 0.05, (0.005, 0.02), (0.03, 0.15)),
    ("mid_families", 0.15, 0, (0.3, 1.0), 0.10, (0.01, 0.03), (0.02, 0.12)),          # 40 families, units of 500-5000 bp
    ("segmental_duplications", 0.05, -1, (1.0, 1.0), 1.0, (0.005, 0.02), (0.005, 0.02)),   # units of 20-200 kbp, 2-10 copies
    ("satellites", 0.03, 171, (1.0, 1.0), 1.0, (0.01, 0.03), (0.01, 0.03)),            # tandem arrays of 1000-5000 units
)


# Other repeat models (--repeat-model), to see that the thresholds of the heavy-read machinery (S3DP_HEAVY, S3DP_HEAVY_RANK, S3DP_CE_GIVE_UP ...) are not
# tuned to one genome: "young" -- three times the share of young, near-identical copies (reads that hit MaxOutputPerRead everywhere); "old" -- hardly any
# young copies and older ones further from their consensus (many dead ends, few hits); "satellite-heavy" -- 15 % tandem arrays and fewer dispersed copies.
REPEAT_MODELS = {
    "grch38-like": REPEAT_FAMILIES,
    "young": (
        ("sine_alu_like", 0.10, 300, (0.85, 1.0), 0.45, (0.002, 0.01), (0.05, 0.15)),
        ("line_l1_like", 0.17, 6000, (0.03, 1.0), 0.15, (0.002, 0.01), (0.03, 0.15)),
        ("mid_families", 0.15, 0, (0.3, 1.0), 0.30, (0.005, 0.02), (0.02, 0.12)),
        ("segmental_duplications", 0.05, -1, (1.0, 1.0), 1.0, (0.002, 0.01), (0.002, 0.01)),
        ("satellites", 0.03, 171, (1.0, 1.0), 1.0, (0.005, 0.02), (0.005, 0.02)),
    ),
    "old": (
        ("sine_alu_like", 0.10, 300, (0.85, 1.0), 0.02, (0.01, 0.03), (0.10, 0.20)),
        ("line_l1_like", 0.17, 6000, (0.03, 1.0), 0.01, (0.01, 0.03), (0.08, 0.20)),
        ("mid_families", 0.15, 0, (0.3, 1.0), 0.02, (0.02, 0.04), (0.08, 0.18)),
        ("segmental_duplications", 0.05, -1, (1.0, 1.0), 1.0, (0.02, 0.05), (0.02, 0.05)),
        ("satellites", 0.03, 171, (1.0, 1.0), 1.0, (0.03, 0.08), (0.03, 0.08)),
    ),
    "satellite-heavy": (
        ("sine_alu_like", 0.08, 300, (0.85, 1.0), 0.15, (0.005, 0.02), (0.05, 0.15)),
        ("line_l1_like", 0.12, 6000, (0.03, 1.0), 0.05, (0.005, 0.02), (0.03, 0.15)),
        ("mid_families", 0.10, 0, (0.3, 1.0), 0.10, (0.01, 0.03), (0.02, 0.12)),
        ("segmental_duplications", 0.05, -1, (1.0, 1.0), 1.0, (0.005, 0.02), (0.005, 0.02)),
        ("satellites", 0.15, 171, (1.0, 1.0), 1.0, (0.005, 0.03), (0.005, 0.03)),
    ),
}


def _plant(codes, unit, copies, frac_lo, frac_hi, young_share, young_div, old_div, g, tandem=0):
    """write `copies` mutated (and half of them reverse-complemented) copies of `unit` at random places; copies are 3'-anchored pieces of the
    unit (LINE-like truncation); tandem > 0: arrays of `tandem` units in a row instead of dispersed copies"""
    import torch
    dev = codes.device
    n = codes.numel()
    ulen = unit.numel()
    ar = torch.arange(ulen, device=dev)
    per = max(1, (1 << 25) // ulen)
    done = 0
    while done < copies:
        m = min(per, copies - done)
        lens = (torch.rand(m, generator=g, device=dev) * (frac_hi - frac_lo) + frac_lo) * ulen
        lens = torch.clamp(lens.to(torch.int64), 20, ulen)
        young = torch.rand(m, generator=g, device=dev) < young_share
        u = torch.rand(m, generator=g, device=dev)
        div = torch.where(young, young_div[0] + u * (young_div[1] - young_div[0]), old_div[0] + u * (old_div[1] - old_div[0]))
        src = unit[None, :].expand(m, ulen)
        mut = torch.rand((m, ulen), generator=g, device=dev) < div[:, None]
        delta = torch.randint(1, 4, (m, ulen), generator=g, device=dev, dtype=torch.uint8)
        seq = torch.where(mut, (src + delta) % 4, src)
        # 3'-anchored piece of length lens: element j of the copy is unit position (ulen - lens + j)
        pos = torch.clamp(ar[None, :] + (ulen - lens)[:, None], max=ulen - 1)
        rc = torch.rand(m, generator=g, device=dev) < 0.5
        pos_rc = torch.clamp((ulen - 1 - ar)[None, :].expand(m, ulen), min=0)
        pos = torch.where(rc[:, None], torch.clamp(pos_rc, min=0), pos)
        piece = torch.gather(seq, 1, pos)
        piece = torch.where(rc[:, None], 3 - piece, piece)
        valid = ar[None, :] < lens[:, None]
        if tandem:
            arrays = max(1, m // tandem)
            base = torch.randint(0, max(1, n - tandem * ulen - 1), (arrays,), generator=g, device=dev)
            k = torch.arange(m, device=dev)
            dst = base[torch.clamp(k // tandem, max=arrays - 1)] + (k % tandem) * ulen
        else:
            dst = torch.randint(0, n - ulen - 1, (m,), generator=g, device=dev)
        flat = dst[:, None] + ar[None, :]
        # copies of one call may overlap: the later copy wins, whatever order the device writes in (a plain indexed store with duplicate indices is
        # not deterministic on the GPU, and every rank -- and every run that reuses the index files -- must see the SAME genome)
        idx, vals = flat[valid], piece[valid]
        idx, order = torch.sort(idx, stable=True)
        last = torch.ones_like(idx, dtype=torch.bool)
        last[:-1] = idx[1:] != idx[:-1]
        codes[idx[last]] = vals[order[last]]
        done += m


def plant_repeats(codes, repeat_frac, seed, model="grch38-like"):
    """overwrite about repeat_frac of `codes` with copies of repeat families (REPEAT_FAMILIES); returns the per-family copy counts"""
    import torch
    dev = codes.device
    n = codes.numel()
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    scale = repeat_frac / 0.5
    report = {}
    for name, share, ulen, (flo, fhi), yshare, ydiv, odiv in REPEAT_MODELS[model]:
        budget = int(n * share * scale)
        if budget <= 0:
            continue
        mean_frac = (flo + fhi) / 2
        if ulen > 0:
            tandem = 0
            copies = max(2, int(budget / (ulen * mean_frac)))
            if name == "satellites":
                tandem = min(3000, max(2, copies // 4))
            unit = torch.randint(0, 4, (ulen,), generator=g, device=dev, dtype=torch.uint8)
            _plant(codes, unit, copies, flo, fhi, yshare, ydiv, odiv, g, tandem)
            report[name] = {"unit": ulen, "copies": copies}
        elif ulen == 0:
            fams = 40
            tot = 0
            for f in range(fams):
                ul = int(torch.randint(500, 5001, (1,), generator=g, device=dev))
                copies = max(2, int(budget / fams / (ul * mean_frac)))
                unit = torch.randint(0, 4, (ul,), generator=g, device=dev, dtype=torch.uint8)
                _plant(codes, unit, copies, flo, fhi, yshare, ydiv, odiv, g)
                tot += copies
            report[name] = {"families": fams, "copies": tot}
        else:
            placed = 0
            cnt = 0
            while placed < budget:
                ul = int(torch.randint(20_000, 200_001, (1,), generator=g, device=dev))
                ul = min(ul, max(1000, n // 50))
                copies = int(torch.randint(2, 11, (1,), generator=g, device=dev))
                src = int(torch.randint(0, n - ul - 1, (1,), generator=g, device=dev))
                unit = codes[src:src + ul].clone()
                _plant(codes, unit, copies, 1.0, 1.0, yshare, ydiv, odiv, g)
                placed += ul * copies
                cnt += copies
            report[name] = {"copies": cnt}
    return report


def make_genome(n, seed, device, repeat_frac=0.0, model="grch38-like"):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    codes = torch.empty(n, dtype=torch.uint8, device=device)
    step = 1 << 28
    for a in range(0, n, step):
        b = min(n, a + step)
        codes[a:b] = torch.randint(0, 4, (b - a,), generator=g, device=device, dtype=torch.uint8)
    families = plant_repeats(codes, repeat_frac, seed + 1, model) if repeat_frac > 0 else {}
    return codes, families


def genome_checksum(codes):
    """position-weighted sum of the text (kept next to the index files of --index-dir: an index is only reused for the genome it was built from)"""
    import torch
    tot = 0
    step = 1 << 28
    for a in range(0, codes.numel(), step):
        c = codes[a:a + step].to(torch.int64)
        w = (torch.arange(a, a + c.numel(), device=codes.device, dtype=torch.int64) % 65521) + 1
        tot = (tot + int((c * w).sum())) % (1 << 61)
    return tot


def make_reads(codes, n_reads, L, seed, device, mix=(0.55, 0.25, 0.12, 0.05, 0.03)):
    """uniform start, strand Bernoulli(0.5), substitutions per read ~ Categorical(mix)"""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = codes.numel()
    out = torch.empty((n_reads, L), dtype=torch.uint8, device=device)
    truth_pos = torch.empty(n_reads, dtype=torch.int64, device=device)
    truth_strand = torch.empty(n_reads, dtype=torch.uint8, device=device)
    truth_sub = torch.empty(n_reads, dtype=torch.uint8, device=device)
    ar = torch.arange(L, device=device)
    probs = torch.tensor(mix, device=device, dtype=torch.float32)
    chunk = 1 << 20
    for a in range(0, n_reads, chunk):
        b = min(n_reads, a + chunk)
        m = b - a
        start = torch.randint(0, n - L, (m,), generator=g, device=device)
        seg = codes[start[:, None] + ar[None, :]]
        nsub = torch.multinomial(probs, m, replacement=True, generator=g)
        score = torch.rand((m, L), generator=g, device=device)
        kth = torch.topk(score, len(mix) - 1, dim=1).indices                  # up to 4 distinct positions
        delta = torch.randint(1, 4, (m, len(mix) - 1), generator=g, device=device, dtype=torch.uint8)
        for j in range(len(mix) - 1):
            sel = nsub > j
            rows = torch.nonzero(sel).squeeze(1)
            cols = kth[rows, j]
            seg[rows, cols] = (seg[rows, cols] + delta[rows, j]) % 4
        strand = torch.randint(0, 2, (m,), generator=g, device=device, dtype=torch.uint8)
        rc = 3 - torch.flip(seg, [1])
        seg = torch.where(strand[:, None] == 1, rc, seg)
        out[a:b] = seg
        truth_pos[a:b] = start
        truth_strand[a:b] = strand
        truth_sub[a:b] = nsub.to(torch.uint8)
    return out, truth_pos, truth_strand, truth_sub


def pack_on_device(codes2d, wpq):
    """reference packed layout (32-interleaved words, chars LSB first) built on the device"""
    import torch
    n, L = codes2d.shape
    dev = codes2d.device
    round_up = (n + 31) // 32 * 32
    buf = torch.zeros((round_up, wpq * 16), dtype=torch.int64, device=dev)
    buf[:n, :L] = codes2d.to(torch.int64)
    sh = 2 * torch.arange(16, dtype=torch.int64, device=dev)
    words = (buf.view(round_up, wpq, 16) << sh).sum(dim=2)
    q = words.view(round_up // 32, 32, wpq).permute(0, 2, 1).reshape(-1)
    return (q & 0xFFFFFFFF).cpu().numpy().astype(np.uint32)


def make_pairs(codes, n_pairs, L, seed, device):
    """2 x L pairs, insert ~ N(350, 40) clipped to [L + 1, 500], '+/-' orientation, 0-4 substitutions per end"""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = codes.numel()
    ins = torch.clamp(torch.round(torch.randn(n_pairs, generator=g, device=device) * 40 + 350), L + 1, 500).to(torch.int64)
    start = torch.randint(0, n - 600, (n_pairs,), generator=g, device=device)
    ar = torch.arange(L, device=device)
    a = codes[start[:, None] + ar[None, :]]
    bb = codes[(start + ins - L)[:, None] + ar[None, :]]
    probs = torch.tensor((0.55, 0.25, 0.12, 0.05, 0.03), device=device)
    for seg in (a, bb):
        nsub = torch.multinomial(probs, n_pairs, replacement=True, generator=g)
        kth = torch.topk(torch.rand((n_pairs, L), generator=g, device=device), 4, dim=1).indices
        delta = torch.randint(1, 4, (n_pairs, 4), generator=g, device=device, dtype=torch.uint8)
        for j in range(4):
            rows = torch.nonzero(nsub > j).squeeze(1)
            seg[rows, kth[rows, j]] = (seg[rows, kth[rows, j]] + delta[rows, j]) % 4
    bb = 3 - torch.flip(bb, [1])
    swap = torch.rand(n_pairs, generator=g, device=device) < 0.5
    first = torch.where(swap[:, None], bb, a)
    second = torch.where(swap[:, None], a, bb)
    out = torch.empty((2 * n_pairs, L), dtype=torch.uint8, device=device)
    out[0::2] = first
    out[1::2] = second
    return out


def make_pairs_c3(codes, n_pairs, L, seed, device, indel_frac=0.10, both_frac=0.0):
    """BASELINE.json configs[2]: 2 x L pairs, insert ~ N(350, 40) clipped to [L + 1, 500], '+/-' orientation, 0-4 substitutions per end
    (55/25/12/5/3 %), and `indel_frac` of the MATES carry one 1-8 base indel (a deletion skips reference bases, an insertion adds random ones): such a
    mate has no <= 4-mismatch hit, its pair is half-aligned and goes to the default DP.  both_frac (BASELINE.json configs[4], SURVEY.md 8d "C5"): that
    share of the PAIRS is damaged in BOTH mates -- each mate gets an indel or, with equal probability, 5-8 substitutions -- so neither end has a BWT hit
    and the pair goes to the deep DP (DV-DPForBothUnalign.cu).  Returns the reads (2 n x L codes, mates adjacent) and the truth:
    leftmost reference position of either mate (0-based, of the mate as simulated: first file = row 2p), its strand, whether it carries an indel,
    whether its pair is one of the both-damaged ones."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = codes.numel()
    ins = torch.clamp(torch.round(torch.randn(n_pairs, generator=g, device=device) * 40 + 350), L + 1, 500).to(torch.int64)
    start = torch.randint(0, n - 700, (n_pairs,), generator=g, device=device)
    ar = torch.arange(L, device=device)
    probs = torch.tensor((0.55, 0.25, 0.12, 0.05, 0.03), device=device)
    both = torch.rand(n_pairs, generator=g, device=device) < both_frac if both_frac > 0 else torch.zeros(n_pairs, dtype=torch.bool, device=device)
    ends, left_pos, has_indel = [], [], []
    for base in (start, start + ins - L):                                   # the left mate, the right mate (both read off the forward strand here)
        k = torch.randint(1, 9, (n_pairs,), generator=g, device=device)
        is_del = torch.rand(n_pairs, generator=g, device=device) < 0.5
        hit = torch.rand(n_pairs, generator=g, device=device) < indel_frac
        many = torch.zeros(n_pairs, dtype=torch.bool, device=device)
        if both_frac > 0:                                                   # a both-damaged pair: this mate gets an indel, or 5-8 substitutions
            by_indel = torch.rand(n_pairs, generator=g, device=device) < 0.5
            hit = torch.where(both, by_indel, hit)
            many = both & ~by_indel
        c = torch.randint(30, L - 30, (n_pairs,), generator=g, device=device)
        j = ar[None, :].expand(n_pairs, L)
        # deletion of k at c: read[j] = ref[j] (j < c), ref[j + k] (j >= c);  insertion of k at c: ref[j] (j < c), random (c <= j < c + k), ref[j - k] (j >= c + k)
        off = torch.where(is_del[:, None], (j >= c[:, None]) * k[:, None], -((j >= (c + k)[:, None]) * k[:, None]))
        off = torch.where(hit[:, None], off, torch.zeros_like(off))
        seg = codes[base[:, None] + j + off]
        rnd = torch.randint(0, 4, (n_pairs, L), generator=g, device=device, dtype=torch.uint8)
        inserted = hit[:, None] & ~is_del[:, None] & (j >= c[:, None]) & (j < (c + k)[:, None])
        seg = torch.where(inserted, rnd, seg)
        nsub = torch.multinomial(probs, n_pairs, replacement=True, generator=g)
        top = 8 if both_frac > 0 else 4
        if both_frac > 0:
            nsub = torch.where(many, torch.randint(5, 9, (n_pairs,), generator=g, device=device), nsub)
        kth = torch.topk(torch.rand((n_pairs, L), generator=g, device=device), top, dim=1).indices
        delta = torch.randint(1, 4, (n_pairs, top), generator=g, device=device, dtype=torch.uint8)
        for t in range(top):
            rows = torch.nonzero(nsub > t).squeeze(1)
            seg[rows, kth[rows, t]] = (seg[rows, kth[rows, t]] + delta[rows, t]) % 4
        del kth, delta
        ends.append(seg)
        left_pos.append(base)
        has_indel.append(hit)
    a, bb = ends[0], 3 - torch.flip(ends[1], [1])
    swap = torch.rand(n_pairs, generator=g, device=device) < 0.5
    out = torch.empty((2 * n_pairs, L), dtype=torch.uint8, device=device)
    out[0::2] = torch.where(swap[:, None], bb, a)
    out[1::2] = torch.where(swap[:, None], a, bb)
    pos = torch.empty(2 * n_pairs, dtype=torch.int64, device=device)
    pos[0::2] = torch.where(swap, left_pos[1], left_pos[0])
    pos[1::2] = torch.where(swap, left_pos[0], left_pos[1])
    strand = torch.empty(2 * n_pairs, dtype=torch.uint8, device=device)       # 0: read off the forward strand, 1: reverse complement
    strand[0::2] = swap.to(torch.uint8)
    strand[1::2] = (~swap).to(torch.uint8)
    indel = torch.empty(2 * n_pairs, dtype=torch.bool, device=device)
    indel[0::2] = torch.where(swap, has_indel[1], has_indel[0])
    indel[1::2] = torch.where(swap, has_indel[0], has_indel[1])
    return out, pos.cpu().numpy(), strand.cpu().numpy(), indel.cpu().numpy(), both.cpu().numpy()


def write_fastq_pairs(reads2d_np, L, path1, path2):
    """the two FASTQ files of a pair set, records of fixed width (names are the pair numbers, qualities 'I'), built as byte matrices"""
    n_pairs = reads2d_np.shape[0] // 2
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    names = np.char.zfill(np.arange(n_pairs).astype("U9"), 9).astype("S9")
    name_bytes = np.frombuffer(names.tobytes(), dtype=np.uint8).reshape(n_pairs, 9)
    for path, rows in ((path1, reads2d_np[0::2]), (path2, reads2d_np[1::2])):
        rec = np.empty((n_pairs, 1 + 9 + 1 + L + 3 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord("@")
        rec[:, 1:10] = name_bytes
        rec[:, 10] = ord("\n")
        rec[:, 11:11 + L] = letters[rows]
        rec[:, 11 + L:14 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        rec[:, 14 + L:14 + 2 * L] = ord("I")
        rec[:, 14 + 2 * L] = ord("\n")
        rec.tofile(path)


def sam_records(p, fix_sign=False):
    """the records of a SAM file as a multiset (collections.Counter of lines), headers apart"""
    from collections import Counter
    if not os.path.exists(p):
        return Counter()
    c = Counter()
    for l in open(p).read().split("\n"):
        if not l or l.startswith("@"):
            continue
        f = l.split("\t")
        if len(f) > 8 and f[8].lstrip("-").isdigit() and abs(int(f[8])) >= 1 << 31:
            # (likewise TLEN of two ends more than 2^31 apart -- never a proper pair: both sides are compared as the 32-bit signed number the reference prints)
            t = int(f[8]) & 0xFFFFFFFF
            f[8] = str(t - (1 << 32) if t >= 1 << 31 else t)
            l = "\t".join(f)
        if fix_sign:
            # the reference's record holds positions as 32-bit SIGNED integers (samtools' bam1_core_t.pos): on this benchmark's ONE sequence of 3.1 G
            # bases everything beyond 2^31 prints negative (GRCh38's longest chromosome is 0.25 G: the reference never sees such a position).  The
            # command line prints the position; for the comparison the reference's negative POS / PNEXT get their 2^32 back.
            f = l.split("\t")
            for k in (3, 7):
                if len(f) > k and f[k].startswith("-"):
                    f[k] = str(int(f[k]) + (1 << 32))
            l = "\t".join(f)
        c[l] += 1
    return c


def dp_reference_check(work, prefix, hi, reads_np, L, in_file, tindel, tboth, n_sample, exe):
    """DP parity at the size the DP stages are benchmarked at: pairs drawn from THIS run -- stratified by where the run's records ended
    (BWT stage / DP stages / no proper pairing) and by what was simulated into them (an indel in one mate: default DP; both mates damaged: deep DP; neither:
    the new default DP and the too-many-hits routes) -- go through the REFERENCE's own host pipeline on the SAME 3.1 Gbp index files (oracle/_ref/ref_harness:
    BWT ladder by its CPU twin, routing, the DP engines' packers / seeding / merges / decode / record writers compiled from /root/reference; the two DP
    kernels behind them are the restatement that tests/test_gpu_dp_reference.py holds bit-identical to the reference's device code) and through the command
    line again; the three SAM files must hold the same records.  The index's .fmv / .rev.fmv files (the CPU 2BWT's occurrence arrays, which only the
    reference's host code reads) are written here by HostIndex.write_fmv_files: byte-identical to soap3-dp-builder's (tests/test_builder.py)."""
    import subprocess
    from collections import Counter
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.path.exists(harness):
        return {"skipped": "oracle/_ref/ref_harness is not built"}
    t_all = time.perf_counter()
    n_pairs = reads_np.shape[0] // 2
    rng = np.random.default_rng(8)
    g, d, u = (in_file.get(k, np.zeros(0, dtype=np.int64)) for k in ("out.gout.1", "out.dpout.1", "out.unpair"))
    one_indel = tindel[0::2] ^ tindel[1::2]
    strata = [("bwt_stage", np.setdiff1d(g, np.union1d(d, u))), ("dp_one_mate_with_indel", d[one_indel[d] & ~tboth[d]]), ("dp_both_mates_damaged", d[tboth[d]]),
              ("dp_neither", d[~tindel[0::2][d] & ~tindel[1::2][d] & ~tboth[d]]), ("no_proper_pairing", u)]
    per = max(1, n_sample // len(strata))
    chosen, tags = [], {}
    for tag, idx in strata:
        if len(idx) > per:
            idx = rng.choice(idx, size=per, replace=False)
        tags[tag] = int(len(idx))
        chosen.append(np.asarray(idx, dtype=np.int64))
    sel = np.unique(np.concatenate(chosen))
    out = {"pairs": int(len(sel)), "strata": tags}
    if len(sel) == 0:
        return dict(out, skipped="no pairs to draw from")
    rows = np.stack([2 * sel, 2 * sel + 1], axis=1).ravel()
    f1, f2 = os.path.join(work, "s1.fq"), os.path.join(work, "s2.fq")
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    for path, rr in ((f1, rows[0::2]), (f2, rows[1::2])):
        with open(path, "w") as f:
            for p_, r_ in zip(sel, rr):
                f.write("@%09d\n%s\n+\n%s\n" % (p_, letters[reads_np[r_]].tobytes().decode(), "I" * L))
    t0 = time.perf_counter()
    if not os.path.exists(prefix + ".fmv"):
        hi.write_fmv_files(prefix)
    out["fmv_files_s"] = time.perf_counter() - t0
    ini = os.path.join(ROOT, "soap3-dp_amd", "host", "soap3-dp.ini")
    opts = ["-L", str(L), "-u", "500", "-v", "1", "-b", "2"]
    t0 = time.perf_counter()
    dbg = bool(os.environ.get("S3DP_BENCH_DP_VERBOSE"))              # (a debugging aid: the DP problems and seeding calls of both sides, compared)
    ref = subprocess.run([harness, "sam", ini, "pair", prefix, f1, f2] + opts + ["-o", "ref"], cwd=work, capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, REFDP_DUMP=os.path.join(work, "ref.problems"), REFDP_SEED_DUMP=os.path.join(work, "ref.seeds")) if dbg else None)
    out["reference_s"] = time.perf_counter() - t0
    if ref.returncode != 0:
        return dict(out, error="ref_harness failed: " + (ref.stdout[-300:] + ref.stderr[-300:]))
    out["reference_stages"] = [l.replace("[harness] ", "") for l in (ref.stdout + ref.stderr).split("\n") if l.startswith("[harness]") and " DP" in l]
    t0 = time.perf_counter()
    mine = subprocess.run([exe, "pair", prefix, f1, f2] + opts + ["-o", os.path.join(work, "sub")], capture_output=True, text=True, timeout=900,
                          env=dict(os.environ, S3DP_VERBOSE="1", S3DP_DUMP_DP=os.path.join(work, "sub.problems"), S3DP_DP_THREADS="1") if dbg else None)
    if dbg:
        out["command_line_stages"] = [l[:300] for l in mine.stderr.split("\n") if "deep DP stage" in l or "default DP call" in l][:40]
        try:
            out["reference_seeding_calls"] = open(os.path.join(work, "ref.seeds")).read().split("\n")[:12]
            rp = Counter(l.split(" | ")[0] for l in open(os.path.join(work, "ref.problems")) if l.startswith("P"))
            mp = Counter(l.split(" | ")[0] for l in open(os.path.join(work, "sub.problems")) if l.startswith("P"))
            only_r, only_m = list((rp - mp).elements()), list((mp - rp).elements())
            by_read_r, by_read_m = Counter(x.split(" ")[8] for x in rp.elements()), Counter(x.split(" ")[8] for x in mp.elements())
            odd = sorted({x.split(" ")[8] for x in only_r + only_m})
            out["dp_problems_by_read"] = [{"read": r_[:24] + "...", "len": len(r_), "reference": by_read_r.get(r_, 0), "command_line": by_read_m.get(r_, 0),
                                           "shapes_only_in_reference": sorted({" ".join(x.split(" ")[1:8]) for x in only_r if x.split(" ")[8] == r_})[:4]} for r_ in odd[:12]]
            out["dp_problems_only_in_reference_that_the_command_line_also_submits_once"] = sum(1 for x in only_r if x in mp)
            out["dp_problems"] = {"reference": sum(rp.values()), "command_line": sum(mp.values()), "only_in_reference": len(only_r), "only_in_command_line": len(only_m),
                                  "examples_only_in_reference": [" ".join(x.split(" ")[:8]) + " " + x.split(" ")[8][:40] + "... " + x.split(" ")[9][:30] + "..." for x in only_r[:6]]}
        except Exception as e:
            out["dp_problems"] = {"error": repr(e)}
    out["command_line_s"] = time.perf_counter() - t0
    if mine.returncode != 0:
        return dict(out, error="soap3-dp pair failed on the sample: " + mine.stderr[-300:])
    total, differing, files, names_differing = 0, [], {}, set()
    for kind in ("gout.1", "dpout.1", "unpair"):
        a, b_ = sam_records(os.path.join(work, "ref." + kind), True), sam_records(os.path.join(work, "sub." + kind))
        files[kind] = {"reference_records": int(sum(a.values())), "records": int(sum(b_.values()))}
        total += sum(a.values())
        for tag_, l in [("reference only", x) for x in list((a - b_).elements())[:6]] + [("command line only", x) for x in list((b_ - a).elements())[:6]]:
            differing.append("%s, %s: %s" % (kind, tag_, "\t".join(l.split("\t")[:9] + l.split("\t")[11:])[:400]))
        if a != b_:
            files[kind]["equal"] = False
            names_differing.update(l.split("\t", 1)[0] for l in list((a - b_).elements()) + list((b_ - a).elements()))
    routes = {k: int(v) for k, v in re.findall(r"(Number of pairs aligned by new default DP|Number of pairs aligned by DP|Number of pairs aligned by deep DP|Pairs left without a proper pairing): (\d+)", mine.stdout)}
    out["position_note"] = ("the reference prints POS / PNEXT beyond 2^31 as negative numbers (32-bit signed positions; the benchmark genome is ONE sequence of 3.1 G bases): "
                            "they are compared after adding 2^32")
    out.update({"records": int(total), "by_file": files, "routes_of_the_sample": routes, "differing_pairs": len(names_differing), "differing_records": differing[:24], "bit_exact": not differing and total > 0,
                "seconds": time.perf_counter() - t_all,
                "what": "pairs of this run, stratified by route, through the reference's own host pipeline (oracle/_ref/ref_harness) on the same index files and through the "
                        "command line again: the records of .gout / .dpout / .unpair as multisets"})
    if differing:
        log("DP parity check against the reference pipeline FAILED: %r" % differing[:3])
    return out


def single_dp_reference_check(work, prefix, hi, reads_np, L, out_prefix, tindel, tboth, n_sample, exe):
    """the single-end twin of dp_reference_check: first mates of THIS run as single reads in the default mode (reads above 120 bases: 100-base seeds in the
    BWT stage, verified over their whole length; what stays unaligned goes through the single-end DP, DV-DPForSingleReads.cu) -- stratified by where the run's
    records ended (BWT stage / DP stage aligned / DP stage unaligned) and by what was simulated into the read -- through the reference's own host pipeline
    (oracle/_ref/ref_harness sam ... single) on the same 3.1 Gbp index files and through the command line again: .gout.1 and .dpout.1 must hold the same
    records."""
    import io
    import subprocess
    import pandas as pd
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if not os.path.exists(harness):
        return {"skipped": "oracle/_ref/ref_harness is not built"}
    t_all = time.perf_counter()
    rng = np.random.default_rng(9)
    where = {}
    for kind in ("gout.1", "dpout.1"):
        p_ = out_prefix + "." + kind
        if not os.path.exists(p_) or os.path.getsize(p_) == 0:
            continue
        cut = subprocess.run("grep -v '^@' %s | cut -f1,2" % p_, shell=True, capture_output=True)
        if not cut.stdout:
            continue
        df = pd.read_csv(io.BytesIO(cut.stdout), sep="\t", header=None, names=["q", "flag"], dtype={"q": np.int64, "flag": np.int64})
        prim = df[(df.flag & 0x900) == 0]
        where[kind] = (prim.q.values, (prim.flag.values & 4) != 0)
    none = (np.zeros(0, dtype=np.int64), np.zeros(0, dtype=bool))
    gq, _ = where.get("gout.1", none)
    dq, dun = where.get("dpout.1", none)
    first_indel = tindel[0::2]
    dal = dq[~dun]
    strata = [("bwt_stage", gq), ("dp_read_with_indel", dal[first_indel[dal]]), ("dp_read_damaged", dal[tboth[dal] & ~first_indel[dal]]),
              ("dp_neither", dal[~first_indel[dal] & ~tboth[dal]]), ("dp_unaligned", dq[dun])]
    per = max(1, n_sample // len(strata))
    chosen, tags = [], {}
    for tag, idx in strata:
        if len(idx) > per:
            idx = rng.choice(idx, size=per, replace=False)
        tags[tag] = int(len(idx))
        chosen.append(np.asarray(idx, dtype=np.int64))
    sel = np.unique(np.concatenate(chosen))
    out = {"reads": int(len(sel)), "strata": tags}
    if len(sel) == 0:
        return dict(out, skipped="no reads to draw from")
    fq = os.path.join(work, "ss.fq")
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(fq, "w") as f:
        for p_ in sel:
            f.write("@%09d\n%s\n+\n%s\n" % (p_, letters[reads_np[2 * p_]].tobytes().decode(), "I" * L))
    t0 = time.perf_counter()
    if not os.path.exists(prefix + ".fmv"):
        hi.write_fmv_files(prefix)
    out["fmv_files_s"] = time.perf_counter() - t0
    ini = os.path.join(ROOT, "soap3-dp_amd", "host", "soap3-dp.ini")
    opts = ["-L", str(L), "-b", "2"]
    t0 = time.perf_counter()
    ref = subprocess.run([harness, "sam", ini, "single", prefix, fq] + opts + ["-o", "sref"], cwd=work, capture_output=True, text=True, timeout=900)
    out["reference_s"] = time.perf_counter() - t0
    if ref.returncode != 0:
        return dict(out, error="ref_harness failed: " + (ref.stdout[-300:] + ref.stderr[-300:]))
    out["reference_stages"] = [l.replace("[harness] ", "") for l in (ref.stdout + ref.stderr).split("\n") if l.startswith("[harness]") and " DP" in l]
    t0 = time.perf_counter()
    mine = subprocess.run([exe, "single", prefix, fq] + opts + ["-o", os.path.join(work, "ssub")], capture_output=True, text=True, timeout=900)
    out["command_line_s"] = time.perf_counter() - t0
    if mine.returncode != 0:
        return dict(out, error="soap3-dp single failed on the sample: " + mine.stderr[-300:])
    total, differing, files, names_differing = 0, [], {}, set()
    for kind in ("gout.1", "dpout.1"):
        a, b_ = sam_records(os.path.join(work, "sref." + kind), True), sam_records(os.path.join(work, "ssub." + kind))
        files[kind] = {"reference_records": int(sum(a.values())), "records": int(sum(b_.values()))}
        total += sum(a.values())
        for tag_, l in [("reference only", x) for x in list((a - b_).elements())[:6]] + [("command line only", x) for x in list((b_ - a).elements())[:6]]:
            differing.append("%s, %s: %s" % (kind, tag_, "\t".join(l.split("\t")[:9] + l.split("\t")[11:])[:400]))
        if a != b_:
            files[kind]["equal"] = False
            names_differing.update(l.split("\t", 1)[0] for l in list((a - b_).elements()) + list((b_ - a).elements()))
    sdp = re.search(r"Number of reads aligned by DP: (\d+)", mine.stdout)
    out.update({"records": int(total), "by_file": files, "reads_aligned_by_dp_in_the_sample": int(sdp.group(1)) if sdp else None, "differing_reads": len(names_differing),
                "differing_records": differing[:24], "bit_exact": not differing and total > 0, "seconds": time.perf_counter() - t_all,
                "what": "first mates of this run as single reads in the default mode, stratified by route, through the reference's own host pipeline "
                        "(oracle/_ref/ref_harness) on the same index files and through the command line again: the records of .gout / .dpout as multisets"})
    if differing:
        log("single-end DP parity check against the reference pipeline FAILED: %r" % differing[:3])
    return out


def c3_leg(s3dp, ix, hi, codes, dev, n_pairs, workdir=None, echo=False, defer_cli=False, both_frac=0.0, resident=True, index_prefix=None, single_too=False,
           seed_offset=0, device_no=None, host_threads=None, truth=True, dp_check=0, single_dp_check=0):
    """BASELINE.json configs[2] as far as one GPU and the time budget of a default run go: n_pairs 2 x 150 bp pairs on the benchmark genome,
    `pair -u 500 -v 1`, DP stages ON, end to end through the drop-in command line (the DP stages are driven by its host code), plus the resident BWT
    stage of the same pairs through the library with the oracle check on a sample.  Reports pairs/s, the stages' wall and device times, the BWT stage's
    algorithmic bytes against its kernels' time, the DP kernels' GCUPS against the vector-ALU peak, and a truth check of the SAM files."""
    import re
    import shutil
    import subprocess
    import tempfile
    import torch
    import s3o
    L = 150
    res = {"pairs": n_pairs, "read_len": L, "options": "pair -L 150 -u 500 -v 1 (default mode: BWT ladder + default / new default / deep / single-end DP)",
           "mixture": "insert N(350, 40) in [151, 500]; 0-4 substitutions per mate (55/25/12/5/3 %); 10 % of the mates carry a 1-8 base indel" +
                      ("; %.0f %% of the PAIRS damaged in both mates (an indel or 5-8 substitutions each): deep DP (BASELINE.json configs[4])" % (100 * both_frac) if both_frac > 0 else "")}
    reads2d, tpos, tstrand, tindel, tboth = make_pairs_c3(codes, n_pairs, L, (31337 if both_frac == 0 else 31338) + seed_offset, dev, both_frac=both_frac)
    wpq = s3dp.word_per_query(L)
    n_res = min(n_pairs, 1 << 20) if resident else 0
    q = pack_on_device(reads2d[:2 * n_res], wpq) if n_res else None
    reads_np = reads2d.cpu().numpy()
    del reads2d
    torch.cuda.empty_cache()
    lens = np.full(2 * n_pairs, L, dtype=np.uint32)
    if not resident:
        ix = None
    # ---- the BWT stage, resident (what the command line's first stage does per chunk), with counters and the oracle on a sample ----
    # (on at most 1 Mi pairs: one search pass numbers its records with 32 bits, and twenty million 150-base reads of a genome that is half repeats pass
    #  2^32 of them -- the command line works in chunks of 512 Ki pairs)
    n_all, n_pairs = n_pairs, n_res
    res["bwt_stage_resident_pairs"] = n_pairs
    b = s3dp.Batch(ix, 2 * n_pairs, wpq) if resident else None
    try:
        if not resident:
            raise StopIteration
        b.upload(q, lens[:2 * n_pairs])
        b.enable_counters(True)
        b.align_pair(3, 8000, 2, 1, 500, 1000)
        b.sync()
        cnt = b.counters()
        b.enable_counters(False)
        dev_ms = search_ms = ce_ms = 0.0
        t0 = time.perf_counter()
        for _ in range(2):
            b.align_pair(3, 8000, 2, 1, 500, 1000)
            b.sync()
            tot, s_ms, _k = b.timing()
            dev_ms += tot
            search_ms += s_ms
            ce_ms += b.ce_timing()[0]
        dt = (time.perf_counter() - t0) / 2
        full = b.fetch_pairs()
        pairs = full[0]
        # SURVEY.md 8d's formula with 12 B per reported occurrence (strict) and the same extended by what lookup tables and check-and-extend read
        # (the hit term is left out of the strict figure here: the stage's output is pairings, its occurrence lists stay on the device)
        bytes_strict = 32.0 * cnt["rankQueries"] + 2 * n_pairs * (4.0 * wpq + 4.0)
        bytes_bwt = bytes_strict + 12.0 * cnt["records"] + 16.0 * cnt["ltLookups"] + 4.0 * cnt["saReads"] + 16.0 * cnt["ceWindows"]
        kern_s = (search_ms + ce_ms) / 2 / 1e3
        res["bwt_stage_resident"] = {"ms_per_pass": dt * 1e3, "device_ms_per_pass": dev_ms / 2, "search_kernels_ms_per_pass": (search_ms + ce_ms) / 2,
                                     "M_pairs_per_s": n_pairs / dt / 1e6, "properly_paired": int((pairs["phase"] > 0).sum()),
                                     "rank_queries": int(cnt["rankQueries"]), "algorithmic_bytes": bytes_bwt,
                                     "roofline": {"bound": "hbm", "achieved": bytes_strict / kern_s / 1e9 if kern_s > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                  "frac": bytes_strict / kern_s / 1e9 / HBM_PEAK_GBS if kern_s > 0 else 0.0,
                                                  "is": "SURVEY.md 8d unextended without the hit term: 32 B x rank queries + packed reads, over the search kernels' time",
                                                  "frac_extended": bytes_bwt / kern_s / 1e9 / HBM_PEAK_GBS if kern_s > 0 else 0.0}}
        # oracle check, stratified like the single-end one: pairs with a hundred and more valid pairings (both ends inside a repeat family: paired by one wave
        # each, s3dp_pair.hip), pairs the ladder left unpaired, pairs with a few pairings, and the first pairs -- 2,048 each; compared: every field of the pair
        # result (phase, optimal pairing, its index statistics, X0 / X1) and the XA lists in pairing order
        xoff, xa = full[1], full[2]
        rng = np.random.default_rng(6)
        npairs_arr, ph = pairs["numPairs"], pairs["phase"]
        strata = (("hundred_and_more_pairings", np.nonzero(npairs_arr >= 100)[0]), ("no_proper_pairing", np.nonzero(ph == 0)[0]),
                  ("two_to_99_pairings", np.nonzero((npairs_arr >= 2) & (npairs_arr < 100))[0]), ("first_pairs", None))
        # (round 5: 2,048 per stratum -- 8,192 pairs -- in blocks of 32 pairs on every granted core; round 4 checked 1,024 on one core in the same five seconds)
        per_stratum = 2048
        chosen, tags = [], {}
        for tag, idx in strata:
            if tag == "first_pairs":
                idx = np.arange(min(per_stratum, n_pairs))
            elif len(idx) > per_stratum:
                idx = np.sort(rng.choice(idx, size=per_stratum, replace=False))
            idx = idx[:len(idx) // 32 * 32]
            tags[tag] = int(len(idx))
            chosen.append(idx)
        sel_pairs = np.concatenate(chosen)
        t0 = time.perf_counter()
        oix = s3o.Index(host_index=hi)
        bad = np.zeros(len(sel_pairs), dtype=bool)

        def check_block(k0):
            blk = sel_pairs[k0:k0 + 32]
            rows = np.stack([2 * blk, 2 * blk + 1], axis=1).ravel()
            want = oix.align_pair(select_packed(q, wpq, rows), np.ascontiguousarray(lens[rows]), wpq, 3, 8000, 2, 1, 500, 1000)
            got = pairs[blk]
            hit = want[0]["phase"] > 0
            b_ = got["phase"] != want[0]["phase"]
            for f in got.dtype.names:
                if f not in ("pad", "phase"):
                    b_ |= hit & (got[f] != want[0][f])
            wxoff, wxa = want[1], want[2]
            for k, pidx in enumerate(blk):
                a0, a1 = int(xoff[pidx]), int(xoff[pidx + 1])
                w0, w1 = int(wxoff[k]), int(wxoff[k + 1])
                if a1 - a0 != w1 - w0 or xa[a0:a1].tobytes() != wxa[w0:w1].tobytes():
                    b_[k] = True
            return k0, b_
        import concurrent.futures as cf
        with cf.ThreadPoolExecutor(max_workers=granted_cpus()) as ex:
            for k0, b_ in ex.map(check_block, range(0, len(sel_pairs), 32)):
                bad[k0:k0 + len(b_)] = b_
        oix.close()
        res["oracle_check"] = {"pairs": int(len(sel_pairs)), "strata": tags, "seconds": time.perf_counter() - t0, "differing_pairs": [int(x) for x in sel_pairs[bad][:8]],
                               "what": "the BWT stage's pair results (phase, optimal pairing, its index statistics, X0 / X1, the XA lists in pairing order) against the CPU oracle, on pairs "
                                       "chosen where the pairing machinery works", "bit_exact": bool(not bad.any())}
    except StopIteration:
        res.pop("bwt_stage_resident_pairs", None)
    finally:
        if b is not None:
            b.close()
    n_pairs = n_all
    # (the command-line part is a function of its own: bench.py's main run calls it after its own resident index is closed -- two copies of a 120 GB index
    #  on one GPU made the child's start-up and teardown a second longer than when it has the device to itself, which is how it is used)
    def cli_part():
        # ---- end to end: the command line with the DP stages on ----
        work = workdir or tempfile.mkdtemp(prefix="s3dp_c3_", dir="/tmp")
        os.makedirs(work, exist_ok=True)
        try:
            t0 = time.perf_counter()
            prefix = index_prefix or os.path.join(work, "g.index")
            if not os.path.exists(prefix + ".sa"):                              # (legs that share an index directory write the files once)
                hi.write_files(prefix, with_lkt=True)
            f1, f2 = os.path.join(work, "r1.fq"), os.path.join(work, "r2.fq")
            write_fastq_pairs(reads_np, L, f1, f2)
            os.sync()                                                          # (files written a moment ago: their write-back is not the program's time)
            res["setup_s"] = time.perf_counter() - t0
            exe = os.path.join(ROOT, "soap3-dp_amd", "soap3-dp")
            # (the command line runs as a child BESIDE this process's resident index: what torch's allocator caches on top of that would be missing from the
            #  child's HBM, and it sizes its lookup tables by what is free)
            import torch
            torch.cuda.empty_cache()
            t0 = time.perf_counter()
            # (a rank of an N > 1 run: its own device, its share of the host's threads -- one process per GPU, SOAP3-DP.cu:219-220)
            cli_env = dict(os.environ, S3DP_HOST_THREADS=str(host_threads)) if host_threads else None
            out = subprocess.run([exe, "pair", prefix, f1, f2, "-L", str(L), "-u", "500", "-v", "1", "-b", "2", "-o", os.path.join(work, "out")] +
                                 (["-c", str(device_no)] if device_no is not None else []), capture_output=True, text=True, env=cli_env)
            wall = time.perf_counter() - t0
            if out.returncode != 0:
                raise RuntimeError("soap3-dp pair failed: " + out.stderr[-500:])
            txt = out.stdout
            if echo:
                log("command line said:\n" + txt + out.stderr)
            ph = {k: float(v) for k, v in re.findall(r"\[Main\] (Loading time|Read load time|Alignment \+ output time|Total elapsed time|DP stages time) :\s+([0-9.]+)", txt)}
            counts = {k: int(v) for k, v in re.findall(r"\[Main\] (Number of pairs aligned by new default DP|Number of pairs aligned by DP|Number of pairs aligned by deep DP|"
                                                       r"Pairs left without a proper pairing|Overall number of reads aligned): (\d+)", txt)}
            stage = re.search(r"DP stages time :\s+[0-9.]+ seconds \((.*)\)", txt)
            ltc = re.search(r"Lookup tables in device memory : (\d+) characters", txt)
            devl = re.search(r"Device time : BWT stages ([0-9.]+) ms \(search kernels ([0-9.]+) ms\), DP kernels ([0-9.]+) ms, (\d+) problems, (\d+) cells, ([0-9.]+) GCUPS", txt)
            align_s = ph.get("Alignment + output time", wall)
            e2e = {"wall_s": wall, "phases_s": ph, "counts": counts, "dp_stage_seconds": stage.group(1) if stage else None,
                   "M_pairs_per_s_reads_in_sam_out": n_pairs / max(align_s, ph.get("Read load time", 0.0)) / 1e6}
            ch = re.search(r"Chunk loop time : alignment calls ([0-9.]+), fetching results ([0-9.]+), formatting \+ writing ([0-9.]+), DP stages ([0-9.]+)", txt)
            if ch:
                e2e["chunk_loop_s"] = {"alignment_calls": float(ch.group(1)), "fetching_results": float(ch.group(2)), "formatting_and_writing": float(ch.group(3)), "dp_stages": float(ch.group(4))}
            if devl:
                dp_ms, cells = float(devl.group(3)), int(devl.group(5))
                e2e["device_ms"] = {"bwt_stages": float(devl.group(1)), "bwt_search_kernels": float(devl.group(2)), "dp_kernels": dp_ms}
                # vector-ALU peak for the DP's cell update: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-ops/s; the paired fill kernel spends about 31 vector
                # instructions per step of 64 lanes x 2 cells = 15.5 per 64 cells -> a ceiling of about 2540 G cell updates per second
                e2e["dp"] = {"problems": int(devl.group(4)), "cells": cells, "GCUPS": cells / (dp_ms / 1e3) / 1e9 if dp_ms > 0 else 0.0,
                             "valu_ceiling_GCUPS": 256 * 4 * 16 * 2.4 / 15.5, "frac_of_valu_ceiling": (cells / (dp_ms / 1e3) / 1e9) / (256 * 4 * 16 * 2.4 / 15.5) if dp_ms > 0 else 0.0}
            if ltc:
                e2e["lookup_chars"] = int(ltc.group(1))
            res["end_to_end"] = e2e
            if not truth:
                return res
            # ---- truth check of the three SAM files: where did every mate land? ----
            import pandas as pd
            seen = np.zeros(2 * n_pairs, dtype=bool)
            at_truth = np.zeros(2 * n_pairs, dtype=bool)
            mapped = np.zeros(2 * n_pairs, dtype=bool)
            by_file, in_file = {}, {}
            for name in ("out.gout.1", "out.dpout.1", "out.unpair"):
                p = os.path.join(work, name)
                if not os.path.exists(p) or os.path.getsize(p) == 0:
                    continue
                cut = subprocess.run("grep -v '^@' %s | cut -f1,2,4" % p, shell=True, capture_output=True)
                if not cut.stdout:
                    continue
                import io
                df = pd.read_csv(io.BytesIO(cut.stdout), sep="\t", header=None, names=["q", "flag", "pos"], dtype={"q": np.int64, "flag": np.int64, "pos": np.int64})
                prim = df[(df.flag & 0x900) == 0]
                row = prim.q.values * 2 + ((prim.flag.values & 0x80) != 0)
                isMapped = (prim.flag.values & 4) == 0
                good = isMapped & (np.abs(prim.pos.values - 1 - tpos[row]) <= 12) & ((((prim.flag.values & 0x10) != 0).astype(np.uint8)) == tstrand[row])
                seen[row] = True
                mapped[row] |= isMapped
                at_truth[row] |= good
                in_file[name] = np.unique(row // 2)
                by_file[name] = {"records": int(len(df)), "mates": int(len(prim)), "mates_mapped": int(isMapped.sum()), "mates_at_simulated_place": int(good.sum())}
            both = at_truth[0::2] & at_truth[1::2]
            res["truth_check"] = {"pairs": n_pairs, "mates_reported": int(seen.sum()), "mates_mapped": int(mapped.sum()), "mates_at_simulated_place": int(at_truth.sum()),
                                  "pairs_with_both_mates_at_simulated_place": int(both.sum()),
                                  "mates_with_indel": int(tindel.sum()), "mates_with_indel_at_simulated_place": int((at_truth & tindel).sum()),
                                  "pairs_damaged_in_both_mates": int(tboth.sum()), "of_those_with_both_mates_at_simulated_place": int((both & tboth).sum()),
                                  "by_file": by_file,
                                  "is": "primary record of every mate against the simulated place (tolerance 12 bases for clips and indels at the ends, strand must match); "
                                        "on a genome that is half repeats a mate inside a repeat family has equally good places elsewhere, so this is a rate, "
                                        "not a pass / fail -- all mates must be reported",
                                  "ok": bool(seen.all())}
            if dp_check > 0:
                try:
                    res["dp_oracle_check"] = dp_reference_check(work, prefix, hi, reads_np, L, in_file, tindel, tboth, dp_check, exe)
                except Exception as e:
                    res["dp_oracle_check"] = {"error": repr(e)}
            if single_too:
                # ---- the first mates alone as single reads in the DEFAULT mode (DP stage on; above 120 bases: 100-base seeds verified on the device) -- the
                # reference's everyday single-end call, which no other leg runs at this size on this genome (58 s for 10 M reads until the end of round 4)
                try:
                    for name in ("out.gout.1", "out.dpout.1", "out.unpair"):
                        if os.path.exists(os.path.join(work, name)):
                            os.remove(os.path.join(work, name))
                    os.sync()
                    t0 = time.perf_counter()
                    se = subprocess.run([exe, "single", prefix, f1, "-L", str(L), "-b", "2", "-o", os.path.join(work, "se")], capture_output=True, text=True)
                    se_wall = time.perf_counter() - t0
                    if se.returncode != 0:
                        raise RuntimeError("soap3-dp single failed: " + se.stderr[-500:])
                    sph = {k: float(v) for k, v in re.findall(r"\[Main\] (Loading time|Read load time|Alignment \+ output time|Total elapsed time) :\s+([0-9.]+)", se.stdout)}
                    sal = re.search(r"Overall number of reads aligned: (\d+)", se.stdout)
                    sdp = re.search(r"Number of reads aligned by DP: (\d+)", se.stdout)
                    s_align = sph.get("Alignment + output time", se_wall)
                    res["single_default_mode"] = {"reads": n_pairs, "read_len": L, "options": "single -L %d (default mode: BWT stage with 100-base seeds + single-end DP)" % L,
                                                  "wall_s": se_wall, "phases_s": sph, "M_reads_per_s_reads_in_sam_out": n_pairs / max(s_align, sph.get("Read load time", 0.0)) / 1e6,
                                                  "reads_aligned": int(sal.group(1)) if sal else None, "reads_aligned_by_dp": int(sdp.group(1)) if sdp else None}
                    if single_dp_check > 0:
                        try:
                            res["single_default_mode"]["dp_oracle_check"] = single_dp_reference_check(work, prefix, hi, reads_np, L, os.path.join(work, "se"), tindel, tboth,
                                                                                                      single_dp_check, exe)
                        except Exception as e:
                            res["single_default_mode"]["dp_oracle_check"] = {"error": repr(e)}
                except Exception as e:
                    res["single_default_mode"] = {"error": repr(e)}
        finally:
            if not workdir:
                shutil.rmtree(work, ignore_errors=True)
    
        return res
    if defer_cli:
        return res, cli_part
    return cli_part()


def se150_leg(s3dp, ix, hi, codes, dev, n_reads):
    """SURVEY.md 8d's second read shape: 150 bp single-end reads, the same all-best ladder, kernel-only reads/s (resident batch, device time)"""
    import torch
    import s3o
    L = 150
    reads2d, tpos, tstrand, tsub = make_reads(codes, n_reads, L, 515, dev)
    wpq = s3dp.word_per_query(L)
    q = pack_on_device(reads2d, wpq)
    del reads2d
    torch.cuda.empty_cache()
    lens = np.full(n_reads, L, dtype=np.uint32)
    b = s3dp.Batch(ix, n_reads, wpq)
    try:
        b.upload(q, lens)
        b.align_single_allbest(4, 1000)
        b.sync()
        dev_ms = search_ms = ce_ms = 0.0
        t0 = time.perf_counter()
        for _ in range(2):
            b.align_single_allbest(4, 1000)
            b.sync()
            tot, s_ms, _k = b.timing()
            dev_ms += tot
            search_ms += s_ms
            ce_ms += b.ce_timing()[0]
        dt = (time.perf_counter() - t0) / 2
        phase, werr, offs, pos, strand, mis = b.fetch_alignments()
        m = 512
        oix = s3o.Index(host_index=hi)
        oph, owe, ooff, opos, ostr, omis, _ = oix.align_single_allbest(q, lens, wpq, 4, 1000, first=0, count=m)
        oix.close()
        ok = (phase[:m] == oph).all() and (offs[:m + 1] == ooff).all() and (pos[:int(ooff[-1])] == opos).all() and (mis[:int(ooff[-1])] == omis).all()
        return {"reads": n_reads, "read_len": L, "ms_per_step": dt * 1e3, "device_ms_per_step": dev_ms / 2, "search_kernels_ms_per_step": (search_ms + ce_ms) / 2,
                "M_reads_per_s": n_reads / dt / 1e6, "M_reads_per_s_kernel_only": n_reads / ((search_ms + ce_ms) / 2 / 1e3) / 1e6,
                "aligned_reads": int((phase > 0).sum()), "oracle_check": {"reads": m, "bit_exact": bool(ok)},
                "options": "single -L 150 -s 4 -h 2 (DP off), same substitution mixture as the headline"}
    finally:
        b.close()


def granted_cpus():
    """CPUs this process may use: the hardware's, cut to the control group's quota and to the affinity mask"""
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return cores


def c4_leg(s3dp, hi, codes, dev, local, rank, world, dist, a, index_prefix, coll_dev):
    """BASELINE.json configs[3] on N GPUs of one node: every rank takes its own shard of a.c4_pairs 2 x 150 bp pairs (own seed) through the drop-in command
    line on ITS device -- one process per GPU over its shard, the reference's model (SOAP3-DP.cu:219-220, README.md:523-536) --, index replicated, host
    threads divided by the number of ranks; the ranks start together (barrier), the job's time is the slowest rank's, its counts are the sums of the shards'
    (`[Main] Number of pairs ...`, SOAP3-DP.cu:816-830), reduced over RCCL."""
    from s3dp import dist as sd
    import torch
    threads = max(2, granted_cpus() // world)
    share = os.environ.get("S3DP_BENCH_SHARE_GPU") == "1"
    err, leg, cli = None, {}, None
    try:                                                                # (a rank that fails must still reach the barrier and the collectives below)
        leg, cli = c3_leg(s3dp, None, hi, codes, dev, a.c4_pairs, defer_cli=True, resident=False, index_prefix=index_prefix, seed_offset=7919 * (rank + 1),
                          device_no=0 if share else local, host_threads=threads, truth=(rank == 0))
    except Exception as e:
        err = repr(e)
    torch.cuda.empty_cache()
    dist.barrier()
    t0 = time.perf_counter()
    try:
        if cli is not None:
            cli()
    except Exception as e:
        err = repr(e)
    wall = time.perf_counter() - t0
    e2e = leg.get("end_to_end") or {}
    if "error" in e2e:
        err = e2e["error"]
    align_s = (e2e.get("phases_s") or {}).get("Alignment + output time", 0.0) if not err else 0.0
    counts = dict((e2e.get("counts") or {})) if not err else {}
    mine = {"pairs": a.c4_pairs if not err else 0, "failed": 1 if err else 0,
            "aligned_reads": counts.get("Overall number of reads aligned", 0), "pairs_default_dp": counts.get("Number of pairs aligned by DP", 0),
            "pairs_new_default_dp": counts.get("Number of pairs aligned by new default DP", 0), "pairs_deep_dp": counts.get("Number of pairs aligned by deep DP", 0),
            "pairs_left": counts.get("Pairs left without a proper pairing", 0), "align_ms": int(align_s * 1e3), "wall_ms": int(wall * 1e3)}
    sums, per_rank = sd.sum_dict_over_ranks(mine, device=coll_dev)
    slowest = sd.max_over_ranks(align_s, device=coll_dev)
    slowest_wall = sd.max_over_ranks(wall, device=coll_dev)
    if rank != 0:
        return None
    res = {"pairs_per_gpu": a.c4_pairs, "gpus": world, "host_threads_per_rank": threads, "options": leg.get("options"), "mixture": leg.get("mixture"),
           "what": "every rank: its own shard through `soap3-dp pair ... -c <its device>` (FASTQ in, three SAM files out, every DP stage on), started together; "
                   "seconds = the slowest rank's alignment + output time (index load apart, as in the C3 leg); counts = sums over the ranks",
           "end_to_end": {"phases_s": {"Alignment + output time": slowest, "Loading time": (e2e.get("phases_s") or {}).get("Loading time")}, "wall_s": slowest_wall,
                          "M_pairs_per_s_reads_in_sam_out": sums["pairs"] / slowest / 1e6 if slowest > 0 else 0.0,
                          "counts": {k: v for k, v in sums.items() if k not in ("align_ms", "wall_ms")},
                          "per_device": [{"M_pairs_per_s": r["pairs"] / (r["align_ms"] / 1e3) / 1e6 if r["align_ms"] else 0.0, **r} for r in per_rank],
                          "counts_sum_equals_sum_of_shards": all(sums[k] == sum(r[k] for r in per_rank) for k in sums),
                          "rank0": {k: e2e.get(k) for k in ("device_ms", "dp", "chunk_loop_s", "dp_stage_seconds", "lookup_chars")}},
           "truth_check": dict(leg.get("truth_check") or {}, of="rank 0's shard")}
    if sums["failed"]:
        res["end_to_end"]["error"] = "%d rank(s) failed: %s" % (sums["failed"], err)
    return res


def extras(s3dp, ix, hi, codes, dev, rank, a):
    """secondary figures (not the headline metric): BASELINE configs[2] -- 2 x 150 bp pairs with the DP stages on (c3_pairs), 150 bp single-end reads
    (se_150bp), and the semi-global DP kernel on default-DP windows"""
    import torch
    res = {}
    L = 150
    wpq = s3dp.word_per_query(L)
    import tempfile
    shared = tempfile.mkdtemp(prefix="s3dp_bench_ix_", dir="/tmp") if (a.c3_pairs > 0 or a.c5_pairs > 0) else None
    res["_index_dir"] = shared
    if a.c3_pairs > 0:
        try:
            res["c3_pairs"], res["_c3_cli"] = c3_leg(s3dp, ix, hi, codes, dev, a.c3_pairs, defer_cli=True, index_prefix=os.path.join(shared, "g.index"), single_too=True,
                                                     dp_check=a.dp_check if a.c5_pairs <= 0 else 0,     # (C5's mixture holds C3's: the DP parity leg runs once, there)
                                                     single_dp_check=a.dp_check)
        except Exception as e:
            res["c3_pairs"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    if a.c5_pairs > 0:
        # BASELINE.json configs[4] on one GPU: the C3 mixture plus 5 % of the pairs damaged in both mates (deep DP, DV-DPForBothUnalign.cu), command line only
        try:
            res["c5_pairs"], res["_c5_cli"] = c3_leg(s3dp, ix, hi, codes, dev, a.c5_pairs, defer_cli=True, both_frac=0.05, resident=False,
                                                      index_prefix=os.path.join(shared, "g.index"), dp_check=a.dp_check)
        except Exception as e:
            res["c5_pairs"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    if a.se150_reads > 0:
        try:
            res["se_150bp"] = se150_leg(s3dp, ix, hi, codes, dev, a.se150_reads)
        except Exception as e:
            res["se_150bp"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    # DP kernel: 150 bp reads against 648-column windows cut from the genome at the reads' places (default-DP shape), device time
    n_dp = 32768
    occ = np.zeros(n_dp, dtype=s3dp.OCC_DTYPE)
    reads2d = make_pairs(codes, n_dp, L, 777, dev)
    q = pack_on_device(reads2d, wpq)
    del reads2d
    lens = np.full(2 * n_dp, L, dtype=np.uint32)
    bb = s3dp.Batch(ix, 2 * n_dp, wpq)
    dp = None
    try:
        bb.upload(q, lens)
        bb.align_pair(3, 8000, 2, 1, 500, 1000)
        pr = bb.fetch_pairs()[0]
        k = 0
        for p in range(n_dp):                                              # the first end's hit of every properly paired pair anchors a DP of the second end
            if pr["phase"][p] and k < n_dp:
                occ[k] = (2 * p, pr["bestA1"][p], pr["bestS1"][p], pr["bestM1"][p], (0, 0))
                k += 1
        dp = s3dp.DpAligner(n_dp, 152, 500 - 1 + 152 + 1, scheme=1)
        dp.half_end(bb, lens, occ[:k], 1, 500, (45, 45), 49, 49)
        t0 = time.perf_counter()
        resd, cig = dp.half_end(bb, lens, occ[:k], 1, 500, (45, 45), 49, 49)
        wall = time.perf_counter() - t0
        ms = dp.timing()
        cells = float(k) * L * 499.0
        res["default_dp_half_end"] = {"problems": k, "read_len": L, "window": 499, "kernel_ms": ms, "GCUPS_kernel": cells / (ms / 1e3) / 1e9,
                                      "engine_call_ms": getattr(dp, "last_call_ms", None),
                                      "engine_call_is": "s3_dp_half_end: pack on the device, fill + traceback kernels, decode, CIGAR text; wall clock of the C call",
                                      "wall_ms_incl_python_binding": wall * 1e3, "aligned_by_dp": int((resd["whichFromDP"] < 2).sum())}
    except Exception as e:
        res["default_dp_half_end"] = {"error": repr(e)}
    finally:
        if dp is not None:
            dp.close()
        bb.close()
    return res


def cpu_baseline(hi, q, lens, wpq, mm, max_out, budget_s):
    """oracle (CPU restatement of the reference path) timed on a bounded sample with every host core: each thread aligns blocks of 256 reads of the
    same batch, same ladder, until the time budget is used up (reads in repeat families cost thousands of times what unique reads cost, so the
    sample is bounded by time, not by a read count fixed in advance)"""
    import concurrent.futures as cf
    import itertools
    import threading
    import s3o
    cores = os.cpu_count() or 1
    try:                                                    # the container's CPU quota, when it has one: threads beyond it only take turns
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    oix = s3o.Index(host_index=hi)
    n_all = len(lens)
    block = 256
    n_blocks = n_all // block
    counter = itertools.count()
    lock = threading.Lock()
    t0 = time.perf_counter()
    deadline = t0 + budget_s

    def work(_):
        done, ranks = 0, 0
        while time.perf_counter() < deadline:
            with lock:
                k = next(counter)
            if k >= n_blocks:
                break
            ranks += oix.align_single_allbest(q, lens, wpq, mm, max_out, first=k * block, count=block)[6].rankQueries
            done += block
        return done, ranks

    with cf.ThreadPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    oix.close()
    total = sum(r[0] for r in res)
    ranks = sum(r[1] for r in res)
    return {"value": total / dt / 1e6, "unit": "M reads/s", "cores": cores, "kind": "port",
            "sample": "the first %d reads of the same batch (blocks of %d dealt to %d threads until %.0f s were used), same ladder, %.1f s wall; "
                      "oracle/s3_oracle.c" % (total, block, cores, budget_s, dt),
            "reads_per_s_per_core": total / dt / cores, "rank_queries_per_read": ranks / max(1, total)}


def cpu_reference_leg(n_reads=200_000):
    """The reference's OWN CPU path timed beside the port (cpu_baseline.kind "reference"): oracle/_ref/ref_harness -- the reference's hostKernel
    (CPUfunctions.cpp:1498: every (read, case) resolved by its CPU twin BGS-HostAlgnmtAlgo2.cpp, occurrences decoded, SAM records written), compiled from the
    reference's sources by oracle/ref.mk -- on a simulated 3 Mbp genome with repeat families and N runs indexed by the reference's own builder
    (oracle/_ref/soap3-dp-builder + BGS-Build; GRCh38-scale files would take the reference builder hours), `single -L 100 -s 4 -b 2`, one CPU thread; and
    the port (oracle/s3_oracle.c, what cpu_baseline times on the headline workload) on the same reads and the same index files, one thread: the ratio
    calibrates the port.  None when the prebuilt reference binaries are not there."""
    import re
    import shutil
    import subprocess
    import tempfile
    import s3dp
    import s3o
    ref = os.path.join(ROOT, "oracle", "_ref")
    need = [os.path.join(ref, x) for x in ("ref_harness", "soap3-dp-builder", "BGS-Build", "soap3-dp-builder.ini")]
    if not all(os.path.exists(x) for x in need):
        return None
    w = tempfile.mkdtemp(prefix="s3dp_refcpu_", dir="/tmp")
    try:
        sim = os.path.join(ROOT, "tools", "simulate.py")
        subprocess.check_call([sys.executable, sim, "--genome-len", "3000000", "--chroms", "4", "--n-runs", "6", "--repeats", "24", "--genome-seed", "101", "--fasta",
                               os.path.join(w, "g.fa"), "--reads", str(n_reads), "--read-len", "100", "--read-seed", "7", "--fastq", os.path.join(w, "se.fq")],
                              stdout=subprocess.DEVNULL)
        shutil.copy(os.path.join(ref, "soap3-dp-builder.ini"), os.path.join(w, "soap3-dp-builder.ini"))
        subprocess.check_call([os.path.join(ref, "soap3-dp-builder"), "g.fa"], cwd=w, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.check_call([os.path.join(ref, "BGS-Build"), "g.fa.index"], cwd=w, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        ini = os.path.join(ROOT, "soap3-dp_amd", "host", "soap3-dp.ini")
        out = subprocess.run([os.path.join(ref, "ref_harness"), "sam", ini, "single", "g.fa.index", "se.fq", "-L", "100", "-s", "4", "-b", "2", "-o", "ref"], cwd=w,
                             capture_output=True, text=True, env=dict(os.environ, REF_HARNESS_TIMING="1"), timeout=1200)
        m = re.search(r"\[ref_harness\] hostKernel: ([0-9.]+) seconds in (\d+) passes over (\d+) reads", out.stderr + out.stdout)
        if out.returncode != 0 or not m:
            return {"error": "ref_harness failed: " + (out.stderr or out.stdout)[-300:]}
        ref_s = float(m.group(1))
        hi = s3dp.HostIndex.from_files(os.path.join(w, "g.fa.index"))
        reads = s3o.read_fastq(os.path.join(w, "se.fq"))
        q, lens, wpq = s3dp.pack_reads(reads, 100)
        oix = s3o.Index(host_index=hi)
        t0 = time.perf_counter()
        res = oix.align_single_allbest(q, lens, wpq, 4, 1000)
        port_s = time.perf_counter() - t0
        oix.close()
        return {"kind": "reference", "what": "oracle/_ref/ref_harness: the reference's hostKernel (CPUfunctions.cpp:1498) compiled from /root/reference by oracle/ref.mk, "
                                             "wall clock inside hostKernel summed over the ladder's passes, one CPU thread",
                "sample": "%d simulated 100 bp reads (tools/simulate.py, read seed 7) on a 3 Mbp genome with repeat families and N runs (genome seed 101), index files by the "
                          "reference's soap3-dp-builder + BGS-Build, single -L 100 -s 4 -b 2" % n_reads,
                "cores": 1, "value": n_reads / ref_s / 1e6, "unit": "M reads/s", "seconds": ref_s,
                "port_same_input": {"kind": "port", "cores": 1, "value": n_reads / port_s / 1e6, "unit": "M reads/s", "seconds": port_s,
                                    "what": "oracle/s3_oracle.c (ladder + decode, no SAM text) on the same reads and index files, one thread",
                                    "occurrences": int(res[2][-1])},
                "port_over_reference": ref_s / port_s}
    except Exception as e:
        return {"error": repr(e)}
    finally:
        shutil.rmtree(w, ignore_errors=True)


def select_packed(q, wpq, rows):
    """the packed words of reads `rows` (a multiple of 32 of them) as a batch of their own: the layout interleaves 32 reads word by word"""
    m = len(rows)
    assert m % 32 == 0
    q3 = q.reshape(-1, wpq, 32)
    return np.ascontiguousarray(q3[rows // 32, :, rows % 32].reshape(m // 32, 32, wpq).transpose(0, 2, 1)).ravel()


def stratified_oracle_check(hi, q, lens, wpq, mm, max_out, phase, werr, offs, pos, strand, mis, want, budget_s):
    """oracle check where the machinery is: reads whose occurrence list was CUT at MaxOutputPerRead (the cut follows the CPU path's search order: units,
    adoption and the running total must all agree with it), reads with hundreds of occurrences below the cut (listed as heavy, taken apart into units),
    reads with a handful, and the first reads of the batch -- a quarter of `want` each, in blocks of 32 on every granted core until `budget_s` seconds
    are used (a read in a repeat family costs the oracle milliseconds).  Bit-exact: phase, WithError, occurrence count, every (position, strand,
    mismatches) in order."""
    import concurrent.futures as cf
    import s3o
    n = len(lens)
    counts = np.diff(offs.astype(np.int64))
    quarter = max(32, want // 4 // 32 * 32)
    strata = (("cut_at_max_output", np.nonzero(counts >= max_out)[0]), ("hundreds_of_occurrences", np.nonzero((counts >= 100) & (counts < max_out))[0]),
              ("two_to_99_occurrences", np.nonzero((counts >= 2) & (counts < 100))[0]), ("first_reads", np.arange(min(n, quarter))))
    rng = np.random.default_rng(4)
    chosen, tags = [], {}
    for tag, idx in strata:
        if tag != "first_reads" and len(idx) > quarter:
            idx = np.sort(rng.choice(idx, size=quarter, replace=False))
        idx = idx[:len(idx) // 32 * 32]
        tags[tag] = len(idx)
        chosen.append(idx)
    rows = np.concatenate(chosen) if chosen else np.zeros(0, dtype=np.int64)
    if not len(rows):
        return {"reads": 0, "bit_exact": True, "strata": tags}
    sel = select_packed(q, wpq, rows)
    ls = np.ascontiguousarray(lens[rows])
    oix = s3o.Index(host_index=hi)
    cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    # the strata take turns, so that a run cut short by the budget still covers all of them
    per = [[] for _ in strata]
    at = 0
    for k, idx in enumerate(chosen):
        per[k] = list(range(at, at + len(idx), 32))
        at += len(idx)
    import itertools
    order = [b for tup in itertools.zip_longest(*per) for b in tup if b is not None]
    t0 = time.perf_counter()
    deadline = t0 + budget_s
    bad, done = [], []

    def work(b0):
        if time.perf_counter() > deadline:
            return
        oph, owe, ooff, opos, ostr, omis, _ = oix.align_single_allbest(sel, ls, wpq, mm, max_out, first=b0, count=32)
        for j in range(32):
            r = int(rows[b0 + j])
            lo, hi_ = int(offs[r]), int(offs[r + 1])
            o0, o1 = int(ooff[j]), int(ooff[j + 1])
            if not (phase[r] == oph[j] and (werr[r] == owe[j]).all() and hi_ - lo == o1 - o0 and (pos[lo:hi_] == opos[o0:o1]).all() and
                    (strand[lo:hi_] == ostr[o0:o1]).all() and (mis[lo:hi_] == omis[o0:o1]).all()):
                bad.append(r)
        done.append(b0)

    with cf.ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(work, order))
    oix.close()
    checked = np.zeros(len(rows), dtype=bool)
    for b0 in done:
        checked[b0:b0 + 32] = True
    by = {}
    at = 0
    for (tag, _), idx in zip(strata, chosen):
        by[tag] = {"chosen": int(len(idx)), "checked": int(checked[at:at + len(idx)].sum())}
        at += len(idx)
    return {"reads": int(checked.sum()), "bit_exact": not bad, "differing_reads": [int(x) for x in bad[:8]], "strata": by, "seconds": time.perf_counter() - t0,
            "what": "reads chosen where the heavy-read machinery works (cut at MaxOutputPerRead / hundreds of occurrences / a few / the first reads), against the CPU oracle: "
                    "phase, WithError, every occurrence in order"}


def truth_check(phase, offs, pos, strand, mis, tpos, tstrand, tsub, max_out):
    """The simulated place of every read against what the ladder reported.  All-best reports the best stratum only, cut at MaxOutputPerRead,
    so for every read with at most 4 substitutions: it must be aligned; unless its list was cut at MaxOutputPerRead (the cut follows the search
    order, not the mismatch count, so a cut list may lack the best hits -- reference behaviour, CPUfunctions.cpp:1612-1615) its best hit has at
    most as many mismatches as were planted, and when it has exactly that many the true (position, strand) must be in the list."""
    n = len(phase)
    counts = np.diff(offs.astype(np.int64))
    ridx = np.repeat(np.arange(n, dtype=np.int64), counts)
    no = int(offs[-1])
    want_strand = np.where(tstrand == 1, 2, 1).astype(np.uint8)          # reference coding: 1 = read as given, 2 = reverse complement
    hit_true = (pos[:no] == tpos[ridx].astype(np.uint32)) & (strand[:no] == want_strand[ridx])
    found = np.bincount(ridx[hit_true], minlength=n) > 0
    best = np.full(n, 255, dtype=np.int64)
    np.minimum.at(best, ridx, mis[:no].astype(np.int64))
    aligned = counts > 0
    cut = counts >= max_out
    unaligned = int((~aligned).sum())
    worse = int((aligned & ~cut & (best > tsub)).sum())
    must = aligned & (best == tsub) & ~cut
    missing = int((must & ~found).sum())
    return {"reads": int(n), "unaligned": unaligned, "best_worse_than_planted": worse, "true_place_missing": missing,
            "true_place_reported": int((must & found).sum()), "better_hit_elsewhere": int((aligned & (best < tsub)).sum()),
            "cut_at_max_output": int(cut.sum()), "multi_mapped_reads": int((counts > 1).sum()),
            "ok": bool(unaligned == 0 and worse == 0 and missing == 0)}


def measured_peaks(dev, local):
    """the two denominators, measured on this GPU at the start of the run: streaming copy bandwidth (read + write bytes per second of a
    device-to-device copy of 2 GiB, HIP events) and the random 64-byte-line gather rate (s3_probe_gather, the product library's own probe)"""
    import torch
    import s3dp
    n = 1 << 31
    x = torch.empty(n, dtype=torch.uint8, device=dev)
    y = torch.empty(n, dtype=torch.uint8, device=dev)
    x.zero_()
    y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 0.0
    for _ in range(3):
        e0.record()
        for _ in range(4):
            y.copy_(x)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * n * 4 / (e0.elapsed_time(e1) / 1e3) / 1e9)
    del x, y
    torch.cuda.empty_cache()
    own = 0.0
    for _ in range(3):                       # the library's own copy kernel (16-byte loads and stores, s3_probe_copy), 4 GiB, read + write bytes
        own = max(own, 2.0 * (1 << 32) / (s3dp.probe_copy(1 << 32, repeats=4, device=local) / 1e3) / 1e9)
    ms, lines = s3dp.probe_gather(int(6e9), blocks=2048, iters=128, ilp=4, loads_per_line=1, device=local)
    return {"copy_GBps": max(best, own), "copy_GBps_own_kernel": own, "copy_GBps_torch_copy": best, "random_Glines_per_s": lines / ms / 1e6,
            "how": "device-to-device copy, read + write bytes: s3_probe_copy (own kernels, 16 B per lane and access, the fastest of six shapes -- grid-stride or one "
                   "workgroup per piece, 1/4/8 loads in flight, non-temporal or ordinary; 4 GiB, best of 3 x 4) and "
                   "torch.Tensor.copy_ (2 GiB, best of 3 x 4); s3_probe_gather over a 6 GB table, 4 independent 64-byte lines in flight per lane"}


def walk_mode_rank_queries(s3dp, hi, local, q, lens, wpq, a, offs):
    """rank queries of the CPU path's own walk on this input, counted by the kernels in walk mode (see run_workload).  With `offs` (the occurrence
    offsets of the timed run, rank 0) the count is pinned to the oracle: the first reads that stay below MaxOutputPerRead are counted by both and the
    numbers must be EQUAL; on reads that reach the limit the kernels' walk can only run a little past the CPU path's stop (it books check-and-extend
    candidates when they are verified, not when they are found), which a second sample -- the first reads whatever they reach -- quantifies."""
    import s3o
    keep = {k: os.environ.get(k) for k in ("S3DP_HEAVY", "S3DP_EARLY_CE", "S3DP_LT_EXTRA")}
    os.environ.update({"S3DP_HEAVY": "0", "S3DP_EARLY_CE": "0", "S3DP_LT_EXTRA": "0"})
    try:
        t0 = time.perf_counter()
        ixw = s3dp.Index(hi, device=local)
        out = {"usable": True}

        def both(rows, tag):
            m = len(rows)
            q3 = q.reshape(-1, wpq, 32)
            sel = np.ascontiguousarray(q3[rows // 32, :, rows % 32].reshape(m // 32, 32, wpq).transpose(0, 2, 1)).ravel()    # the packed layout interleaves 32 reads
            ls = np.ascontiguousarray(lens[rows])
            bs = s3dp.Batch(ixw, m, wpq)
            bs.upload(sel, ls)
            bs.enable_counters(True)
            bs.align_single_allbest(a.mismatch, a.max_out)
            bs.sync()
            got = int(bs.counters()["rankQueries"])
            bs.close()
            oix = s3o.Index(host_index=hi)
            st = oix.align_single_allbest(sel, ls, wpq, a.mismatch, a.max_out)[-1]
            oix.close()
            out[tag] = {"reads": m, "kernels": got, "oracle": int(st.rankQueries), "equal": bool(got == st.rankQueries)}
            return out[tag]

        if offs is not None and a.check > 0:
            m = min(a.check, a.reads) // 32 * 32
            counts = np.diff(offs.astype(np.int64))
            below = np.nonzero(counts < a.max_out)[0][:m]
            below = below[:len(below) // 32 * 32]
            if len(below) and not both(below, "sample_below_limit")["equal"]:
                out["usable"] = False                  # (reported, not used: the line then falls back to the CPU sample's scaled count)
                log("walk-mode rank-query count differs from the oracle's on the sample: %r" % out["sample_below_limit"])
            both(np.arange(m), "sample_first_reads")
        bw = s3dp.Batch(ixw, a.reads, wpq)
        bw.upload(q, lens)
        bw.enable_counters(True)
        bw.align_single_allbest(a.mismatch, a.max_out)
        bw.sync()
        out["rank_queries"] = int(bw.counters()["rankQueries"])
        out["seconds"] = time.perf_counter() - t0
        bw.close()
        ixw.close()
        return out
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def run_workload(a, s3dp, builder, dev, local, rank, world, dist, peaks, repeat_frac, label, index_dir=None, want_cpu=True):
    """one workload = genome + index + reads + counted pass + timed region + checks; returns (json fragment, objects kept for the extras)"""
    import torch
    from s3dp import dist as sd
    t0 = time.perf_counter()
    model = getattr(a, "repeat_model", "grch38-like")
    codes, families = make_genome(a.genome_bp, 1, dev, repeat_frac, model)
    if repeat_frac > 0 and model != "grch38-like":
        label = label + "-" + model
    log("[%s] genome %d bp (repeat_frac %.2f) generated in %.1f s" % (label, a.genome_bp, repeat_frac, time.perf_counter() - t0))
    t0 = time.perf_counter()
    reads2d, tpos, tstrand, tsub = make_reads(codes, a.reads, a.read_len, 2026 + rank, dev)
    wpq = s3dp.word_per_query(a.read_len)
    q = pack_on_device(reads2d, wpq)
    lens = np.full(a.reads, a.read_len, dtype=np.uint32)
    tpos, tstrand, tsub = tpos.cpu().numpy(), tstrand.cpu().numpy(), tsub.cpu().numpy().astype(np.int64)
    del reads2d
    log("[%s] %d reads generated + packed in %.1f s" % (label, a.reads, time.perf_counter() - t0))
    # index: built ONCE (rank 0, on its GPU), written in the reference's file formats, loaded by every rank; --index-dir keeps the files for the
    # next run (profiling runs load them instead of tracing the thousands of small kernels of the suffix sort)
    t0 = time.perf_counter()
    keep_dir = a.index_dir or index_dir
    prefix = os.path.join(keep_dir, "bench_%s_%d.index" % (label, a.genome_bp)) if keep_dir else None
    gsum = genome_checksum(codes)
    reuse = False
    if prefix and a.index_dir and os.path.exists(prefix + ".sa") and os.path.exists(prefix + ".genome_sum"):
        reuse = open(prefix + ".genome_sum").read().strip() == str(gsum)
        if not reuse:
            log("[%s] index files under %s belong to another genome: rebuilding" % (label, keep_dir))
    if reuse:
        hi = s3dp.HostIndex.from_files(prefix)
    elif world > 1 or prefix:
        if rank == 0:
            os.makedirs(keep_dir, exist_ok=True)
            hi = builder.build_host_index(codes.cpu().numpy(), device=dev, verbose=True)
            hi.write_files(prefix, with_lkt=True)
            with open(prefix + ".genome_sum", "w") as f:
                f.write("%d\n" % gsum)
        if world > 1:
            dist.barrier()
            if rank != 0:
                if open(prefix + ".genome_sum").read().strip() != str(gsum):
                    raise SystemExit("bench.py: rank %d generated a different genome than rank 0" % rank)
                hi = s3dp.HostIndex.from_files(prefix, shared=True)       # one set of pages per node, not one copy per rank
    else:
        hi = builder.build_host_index(codes.cpu().numpy(), device=dev, verbose=(rank == 0))
    torch.cuda.empty_cache()
    log("[%s] index built in %.1f s" % (label, time.perf_counter() - t0))
    if a.build_only:
        raise SystemExit(0)
    t0 = time.perf_counter()
    ix = s3dp.Index(hi, device=local)
    log("[%s] index resident: %.2f GB in HBM, upload %.1f s" % (label, ix.device_bytes() / 1e9, time.perf_counter() - t0))
    b = s3dp.Batch(ix, a.reads, wpq)
    b.upload(q, lens)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- one counted pass: algorithmic work (N_rank etc.) ------------------------------------------------
    b.enable_counters(True)
    b.align_single_allbest(a.mismatch, a.max_out)
    b.sync()
    cnt = b.counters()
    kcnt = b.kernel_stats()                     # the same tallies per kernel (first pass, trunk pass, unit pass, check-and-extend)
    phase, werr, offs, pos, strand, mis = b.fetch_alignments()
    aligned = int((phase > 0).sum())
    n_occ = int(offs[-1])
    # the same pass with ONE first-mismatch position per unit (S3DP_UNIT_GROUP=1; the single-end ladders take groups of two): what the trunk and unit passes
    # need to issue -- a unit of a group walks the trunk's path through its group again, and `rank_queries_needed` must not credit those steps
    kneed = None
    if os.environ.get("S3DP_UNIT_GROUP") != "1":
        os.environ["S3DP_UNIT_GROUP"] = "1"
        try:
            b.align_single_allbest(a.mismatch, a.max_out)
            b.sync()
            kneed = b.kernel_stats()
        finally:
            del os.environ["S3DP_UNIT_GROUP"]
    b.enable_counters(False)

    # ---- timed region: resident batch ----------------------------------------------------------------------
    for _ in range(a.warmup):
        b.align_single_allbest(a.mismatch, a.max_out)
        b.sync()
    barrier()
    t0 = time.perf_counter()
    search_ms = ce_ms = dev_ms = 0.0
    launches = 0
    kms = {k: 0.0 for k in ("first", "trunk", "unit", "ce")}
    for _ in range(a.steps):
        b.align_single_allbest(a.mismatch, a.max_out)
        b.sync()
        tot_ms, s_ms, k = b.timing()        # HIP events on the batch stream
        for name, st in b.kernel_stats().items():
            kms[name] += st["ms"]
        search_ms += s_ms
        dev_ms += tot_ms
        ce_ms += b.ce_timing()[0]
        launches += k
    barrier()
    dt = time.perf_counter() - t0
    dt = sd.max_over_ranks(dt, device=COLL_DEV[0] if COLL_DEV else dev)
    # the only collective of the path: the final reduce of the counters (RCCL all-reduce)
    tot = sd.reduce_counters({"reads": a.reads, "readsAligned": aligned, "alignments": n_occ, "rankQueries": cnt["rankQueries"],
                              "records": cnt["records"]}, device=COLL_DEV[0] if COLL_DEV else dev)
    value = tot["reads"] * a.steps / dt / 1e6
    log("[%s] resident: %.2f M reads/s, %.1f ms per step (stepping %.1f ms, check-and-extend %.1f ms, device %.1f ms); %.0f rank queries per read" %
        (label, value, dt / a.steps * 1e3, search_ms / a.steps, ce_ms / a.steps, dev_ms / a.steps, cnt["rankQueries"] / a.reads))

    # ---- streamed: upload of the packed reads and download of the results inside the timed region -------------
    streamed = None
    if a.streaming:
        b.close()                                   # the stream's slots bring their own batches
        mism = []

        def chk(first, ph, of, po, sr, mi):         # the streamed results must be the resident batch's
            m = len(ph)
            lo, hi = int(offs[first]), int(offs[first + m])
            if not ((ph == phase[first:first + m]).all() and (of.astype(np.int64) == (offs[first:first + m + 1].astype(np.int64) - lo)).all() and
                    (po == pos[lo:hi]).all() and (sr == strand[lo:hi]).all() and (mi == mis[lo:hi]).all()):
                mism.append(first)
        def one_stream(batch_reads, depth):
            st = s3dp.stream_bench(ix, q, lens, wpq, a.mismatch, a.max_out, batch_reads, min(a.steps, 5), depth=depth, check=chk)
            if mism:
                raise SystemExit("bench.py: streamed results differ from the resident batch's (batches at reads %r)" % mism[:5])
            st["same_results_as_resident"] = True
            st["M_reads_per_s"] = sd.reduce_counters({"reads": int(st["reads"])}, device=COLL_DEV[0] if COLL_DEV else dev)["reads"] / sd.max_over_ranks(st["seconds"], device=COLL_DEV[0] if COLL_DEV else dev) / 1e6
            st["vs_resident"] = st["M_reads_per_s"] / value
            log("[%s] streamed, %d reads per batch, %d in flight: %.2f M reads/s (%.2f of resident)" % (label, batch_reads, depth, st["M_reads_per_s"], st["vs_resident"]))
            return st
        streamed = one_stream(a.stream_batch, a.stream_depth)
        if a.stream_batch != 1 << 20 and a.reads >= 4 << 20:
            # the reference's own batch (1 Mi reads, SOAP3-DP.cu:394), which is also the command line's chunk: six batches in flight
            streamed["batches_of_1Mi"] = one_stream(1 << 20, 6)

    # ---- roofline of the search (stepping kernel + check-and-extend kernel: the two halves of one search pass) ----------------
    # algorithmic bytes: SURVEY.md 8d's formula (strict), and the same extended by what the lookup tables and check-and-extend read: 16 B per lookup
    # (one long-table entry, or the four 4-byte entries of the 13-mer tables), 4 B per suffix-array entry, 16 B per 32-character compare window
    # (8 B of packed text + 8 B of the read)
    # N_rank: what the kernels counted, or -- when that is MORE than the reference's own algorithm needs on this input (reads taken apart into
    # units walk shared parts again; a lane cannot stop where the CPU path's running total stops it) -- the count of the CPU restatement, scaled
    # from its sample ("N_rank counted exactly by the build's CPU restatement on the same input")
    cpu = None
    if rank == 0 and world == 1 and a.cpu_budget > 0 and want_cpu:
        cpu = cpu_baseline(hi, q, lens, wpq, a.mismatch, a.max_out, a.cpu_budget)
    # N_rank, counted (not extrapolated): the reference algorithm's own number of rank queries on THIS input is what the kernels issue when they
    # walk like the CPU path -- one lane per read, no early check-and-extend, the file format's 13-mer tables (S3DP_HEAVY=0 S3DP_EARLY_CE=0
    # S3DP_LT_EXTRA=0; the GPU tests assert that this count equals the oracle's).  One such pass over all reads, outside the timed region; on the
    # checked sample the same pass must give the oracle's count exactly.
    walk = walk_mode_rank_queries(s3dp, hi, local, q, lens, wpq, a, offs if rank == 0 else None) if a.exact_rank else None
    n_rank = float(cnt["rankQueries"])
    if walk is not None and walk["usable"]:
        n_rank = min(n_rank, float(walk["rank_queries"]))
    elif cpu and cpu["rank_queries_per_read"] * a.reads < n_rank:
        n_rank = cpu["rank_queries_per_read"] * a.reads
    # headline (nothing added): 32 B per rank query of the reference algorithm + the packed reads + 12 B per OUTPUT hit (the occurrences the
    # ladder reports); the extended figure, a named secondary: 12 B per internal record instead and what lookup tables and check-and-extend read
    bytes_strict = 32.0 * n_rank + a.reads * (4.0 * wpq + 4.0) + 12.0 * n_occ
    bytes_step = 32.0 * n_rank + a.reads * (4.0 * wpq + 4.0) + 12.0 * cnt["records"] + 16.0 * cnt["ltLookups"] + 4.0 * cnt["saReads"] + 16.0 * cnt["ceWindows"]
    search_s_per_step = (search_ms + ce_ms) / 1e3 / max(1, a.steps)
    achieved = bytes_step / search_s_per_step / 1e9 if search_s_per_step > 0 else 0.0
    achieved_strict = bytes_strict / search_s_per_step / 1e9 if search_s_per_step > 0 else 0.0
    # HBM traffic of the two kernels comes from a separate rocprofv3 --pmc FETCH_SIZE pass (it cannot be collected from inside this process);
    # the committed measurement is attached when it was taken on exactly this workload, per launch like `achieved`
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            for w in json.load(f)["workloads"]:
                if all(w[k] == v for k, v in (("genome_bp", a.genome_bp), ("reads_per_gpu", a.reads), ("read_len", a.read_len),
                                              ("max_mismatch", a.mismatch), ("max_output_per_read", a.max_out), ("repeat_frac", repeat_frac))):
                    traffic = w["fetch_bytes_per_step"] / w["launches_per_step"]
    except (OSError, ValueError, KeyError):
        traffic = None
    n_launch = max(1, launches // max(1, a.steps))
    peak = HBM_PEAK_GBS
    # per kernel: what IT issued (counted pass), its own duration (HIP events around its launches, timed steps), its fetched bytes (the committed
    # FETCH_SIZE pass, split by kernel name) -- so that the dominant kernel's fraction can be recomputed from this line alone
    per_kernel, traffic_roles = {}, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            for w in json.load(f)["workloads"]:
                if traffic is not None and w.get("repeat_frac") == repeat_frac and w.get("reads_per_gpu") == a.reads:
                    traffic_roles = w.get("per_role")
    except (OSError, ValueError, KeyError):
        traffic_roles = None
    for name in ("first", "trunk", "unit", "ce"):
        kc, ms = kcnt[name], kms[name] / max(1, a.steps)
        b8d = 32.0 * kc["rankQueries"] + (a.reads * (4.0 * wpq + 4.0) if name == "first" else 0.0)
        bext = b8d + 16.0 * kc["ltLookups"] + 4.0 * kc["saReads"] + 16.0 * kc["ceWindows"]
        fetched = traffic_roles[name]["bytes"] if traffic_roles and name in traffic_roles else None
        # needed: the first pass without what it spent on reads it then listed (the second pass starts those over); trunk and unit pass with one position per
        # unit (kneed); never more than issued.  `frac` is priced on the needed count, `frac_issued` on what the kernel issued.
        need = kc["rankQueries"] - (kc.get("listedRankQueries", 0) if name == "first" else 0)
        if kneed is not None and name in ("trunk", "unit"):
            need = min(need, kneed[name]["rankQueries"])
        b8d_issued = b8d
        b8d = 32.0 * need + (a.reads * (4.0 * wpq + 4.0) if name == "first" else 0.0)
        bext += b8d - b8d_issued
        per_kernel[name] = {"ms_per_step": ms, "launches_per_step": kc["launches"], "rank_queries": kc["rankQueries"], "rank_queries_needed": need,
                            "lt_lookups": kc["ltLookups"], "frac_issued": b8d_issued / ms / 1e6 / peak if ms > 0 else 0.0,
                            "sa_reads": kc["saReads"], "ce_windows": kc["ceWindows"], "algorithmic_bytes": b8d, "algorithmic_bytes_extended": bext,
                            "achieved_GBps": b8d / ms / 1e6 if ms > 0 else 0.0, "frac": b8d / ms / 1e6 / peak if ms > 0 else 0.0,
                            "frac_extended": bext / ms / 1e6 / peak if ms > 0 else 0.0, "fetched_bytes": fetched,
                            "fetched_GBps": fetched / ms / 1e6 if fetched and ms > 0 else None,
                            "fetched_Glines_per_s": fetched / 64.0 / ms / 1e6 if fetched and ms > 0 else None}
    per_kernel["first"]["rank_queries_on_reads_then_listed"] = kcnt["first"].get("listedRankQueries", 0)
    # where ceKernel's rejected candidates are decided (s3_kernel_stats.ceRejected; counted pass)
    rej = kcnt["ce"].get("ceRejected")
    if rej and sum(rej):
        tot_rej = float(sum(rej))
        hist = {}
        for d, dn in enumerate(("backward_step", "forward_step")):
            for bud, bn in enumerate(("budget_0", "budget_1", "budget_2_or_more")):
                w = rej[(d * 3 + bud) * 3:(d * 3 + bud) * 3 + 3]
                hist["%s.%s" % (dn, bn)] = {"within_32_adjacent_chars": w[0], "rest_of_that_step": w[1], "a_later_step": w[2]}
        per_kernel["ce"]["rejected"] = {"candidates": kcnt["ce"]["saReads"], "accepted": kcnt["ce"].get("accepted", 0), "rejected": int(tot_rej),
                                        "share_within_32_adjacent_chars": sum(rej[0::3]) / tot_rej,
                                        "share_within_32_adjacent_chars_backward_steps_only": sum(rej[0:9:3]) / tot_rej,
                                        "share_handed_over_in_backward_steps": sum(rej[0:9]) / tot_rej, "by_direction_and_budget": hist}
    traffic_commit = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            traffic_commit = json.load(f).get("commit")
    except (OSError, ValueError):
        pass
    roof = {"bound": "hbm", "achieved": achieved_strict, "peak": peak, "unit": "GB/s", "frac": achieved_strict / peak, "traffic": traffic,
            "achieved_is": "SURVEY.md 8d, nothing added: (32 B x rank queries of the reference algorithm on this input + packed reads + 12 B x reported occurrences) per step "
                           "/ time inside the search kernels per step",
            "peak_is": "HBM3E peak of the MI355X (MI355X_MICROARCH.md); peak_measured: the best device-to-device copy on this GPU at the start of the run "
                       "(own 16-byte copy kernel and torch copy, read + write bytes)",
            "peak_measured": peaks["copy_GBps"], "frac_of_measured": achieved_strict / peaks["copy_GBps"],
            "peak_spec": HBM_PEAK_GBS, "frac_of_spec": achieved_strict / HBM_PEAK_GBS,
            "achieved_extended": achieved, "frac_extended": achieved / peak,
            "extended_is": "secondary: 12 B per internal record instead of per reported occurrence, + 16 B per lookup-table entry + 4 B per suffix-array entry + 16 B per 32-character compare window",
            "traffic_unit": "bytes per launch pair (FETCH_SIZE of the search kernels from a separate rocprofv3 --pmc pass, profiles/pmc_traffic.json: a committed measurement of this workload, "
                            "not taken by this run)",
            "traffic_measured_at_commit": traffic_commit if traffic else None,
            "algorithmic_bytes_per_launch": bytes_strict / n_launch, "algorithmic_bytes_per_launch_extended": bytes_step / n_launch,
            "kernel": "s3::firstKernel + s3::unitKernel + s3::ceKernel (the stepping kernels and the check-and-extend kernel: the halves of one search pass)", "launches_per_step": n_launch,
            "kernel_ms_per_step": (search_ms + ce_ms) / max(1, a.steps),
            "stepping_kernel_ms_per_step": search_ms / max(1, a.steps),
            "check_extend_kernel_ms_per_step": ce_ms / max(1, a.steps), "device_ms_per_step": dev_ms / max(1, a.steps),
            "sa_reads_per_step": cnt["saReads"], "ce_windows_per_step": cnt["ceWindows"], "lt_lookups_per_step": cnt["ltLookups"],
            "rank_queries_per_step": cnt["rankQueries"], "rank_queries_algorithmic": n_rank,
            "rank_queries_algorithmic_is": ("min(what the timed kernels issued, what the kernels issue when they walk every read like the CPU path: "
                                            "one device pass over all reads, counted)") if walk is not None and walk["usable"] else
                                           "min(what the kernels issued, the CPU restatement's count scaled from its sample)",
            "walk_mode_pass": walk,
            "per_kernel": per_kernel,
            "per_kernel_is": "first = s3::firstKernel (first pass), trunk = s3::firstKernel (trunk-pass role), unit = s3::unitKernel, ce = s3::ceKernel; counts: what the kernel "
                             "issued in the counted pass; algorithmic_bytes = 32 B x its rank queries (+ the packed reads for the first pass), SURVEY.md 8d without the hit "
                             "term (12 B x reported occurrences belong to the whole pass); _extended adds 16 B per lookup entry, 4 B per suffix-array entry, 16 B per "
                             "32-character compare window (all of ceKernel's work); rank_queries_needed: first pass -- without what it spent on reads it then listed; trunk / "
                             "unit pass -- with one first-mismatch position per unit (a group's unit walks the trunk's path again); frac on needed, frac_issued on issued; "
                             "ms = HIP events around the kernel's own launches, mean over the timed steps; "
                             "fetched_bytes = FETCH_SIZE of its dispatches in the committed rocprofv3 pass (profiles/pmc_traffic.json)",
            "dominant_kernel": max(per_kernel, key=lambda k: per_kernel[k]["ms_per_step"])}
    if traffic:
        # the path is made of random accesses with 4-32 B payloads; the GPU moves 64 B per access (tools/hbm_probe.py), so the rate of fetched
        # 64-byte lines against the probe measured in this run is what tells how busy the memory system is
        lines = traffic * n_launch / 64.0 / search_s_per_step / 1e9
        roof["random_line_rate"] = {"achieved_Glines_per_s": lines, "probe_Glines_per_s": peaks["random_Glines_per_s"],
                                    "frac": lines / peaks["random_Glines_per_s"]}
    out = {"value": value, "ms_per_step": dt / a.steps * 1e3,
           "config": {"workload": label, "genome_bp": a.genome_bp, "reads_per_gpu": a.reads, "read_len": a.read_len, "max_mismatch": a.mismatch,
                      "max_output_per_read": a.max_out, "repeat_frac": repeat_frac, "repeat_model": model if repeat_frac > 0 else None, "repeat_families": families,
                      "parallelism": "reads sharded x%d, index replicated" % world,
                      "lookup_chars": int(s3dp.lib().s3_index_lookup_chars(ix.h))},
           "roofline": roof, "aligned_reads": tot["readsAligned"], "occurrences": tot["alignments"],
           "rank_queries_per_read": cnt["rankQueries"] / a.reads}
    if streamed:
        out["streamed"] = streamed
    if cpu:
        if a.cpu_reference:
            t0 = time.perf_counter()
            cpu["reference"] = cpu_reference_leg(a.cpu_reference)
            log("[%s] reference CPU path timed beside the port in %.1f s: %r" % (label, time.perf_counter() - t0,
                                                                                 {k: v for k, v in (cpu["reference"] or {}).items() if k in ("value", "port_over_reference", "error")}))
        out["cpu_baseline"] = cpu
    if rank == 0:
        out["truth_check"] = truth_check(phase, offs, pos, strand, mis, tpos, tstrand, tsub, a.max_out)
        if a.check > 0:
            out["oracle_check"] = stratified_oracle_check(hi, q, lens, wpq, a.mismatch, a.max_out, phase, werr, offs, pos, strand, mis, min(a.check, a.reads), a.check_budget)
            if not out["oracle_check"]["bit_exact"]:
                raise SystemExit("bench.py: HIP result differs from the CPU oracle (%s): reads %r" % (label, out["oracle_check"]["differing_reads"]))
        if not out["truth_check"]["ok"]:
            raise SystemExit("bench.py: truth check failed (%s): %r" % (label, out["truth_check"]))
    if not a.streaming:
        b.close()
    return out, dict(ix=ix, hi=hi, codes=codes, q=q, lens=lens, wpq=wpq, prefix=prefix)


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher around it: start N ranks of this same script, one per GPU, as CHILD processes of
    torch.distributed.run (the reference's model is one process per GPU, SOAP3-DP.cu:219-220, README.md:523-536), pass their stderr through, print
    rank 0's JSON line, return their exit code.  This process only counts the devices (torch.cuda.device_count() does not initialise the HIP runtime
    on this image) and waits: a process that has touched a GPU is never replaced by another program."""
    import socket
    import subprocess
    if not a.launch_selftest and os.environ.get("S3DP_BENCH_SHARE_GPU") != "1":
        import torch
        have = torch.cuda.device_count()
        if have < a.gpus:
            print("bench.py: --gpus %d but this node shows %d HIP device%s" % (a.gpus, have, "" if have == 1 else "s"), file=sys.stderr)
            return 2
    with socket.socket() as sk:                                         # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [x for x in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in p.stdout:                                                 # rank 0 prints ONE JSON line; anything else on stdout goes to stderr
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.rstrip("\n")
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the ranks ended without a result line", file=sys.stderr)
        rc = 1
    return rc


def launch_selftest_rank(a):
    """one rank of `bench.py --gpus N --launch-selftest 1` (CPU, gloo): the N > 1 path's plumbing -- environment from the launcher, process group,
    the shard bounds, the counters' reduce, barrier + max-over-ranks timing, ONE line from rank 0 -- around a stand-in for the timed region"""
    import torch.distributed as dist
    from s3dp import dist as sd
    world, rank, local = sd.world()
    if a.gpus != world:
        print("bench.py: --gpus %d but WORLD_SIZE is %d" % (a.gpus, world), file=sys.stderr)
        return 2
    dist.init_process_group(backend="gloo")
    total = a.reads * world                                             # weak scaling: a.reads per rank
    lo, hi = sd.shard_bounds(total, rank, world)
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))                                       # (the slowest rank sets the time)
    dist.barrier()
    dt = sd.max_over_ranks(time.perf_counter() - t0)
    tot = sd.reduce_counters({"reads": hi - lo, "readsAligned": hi - lo, "alignments": (hi - lo) * 2})
    if rank == 0:
        print(json.dumps({"metric": "launch selftest (no alignment done)", "value": tot["reads"] / dt / 1e6, "unit": "M reads/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "u32", "data": "none", "config": {"workload": "launch selftest"}, "selftest": True,
                          "reads": tot["reads"], "alignments": tot["alignments"], "max_seconds": dt}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def summary_of(out):
    """the figures a reader of the stored tail needs: headline, per-kernel roofline of the dominant kernel, C3 / C5 legs, robustness legs"""
    def g(d, *path, default=None):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return default
            d = d[k]
        return d
    roof = out.get("roofline", {})
    pk = roof.get("per_kernel") or {}
    sm = {"headline_M_reads_per_s": out.get("value"), "ms_per_step": out.get("ms_per_step"), "n_gpus": out.get("n_gpus"), "roofline_frac": roof.get("frac"),
          "per_kernel": {k: {f: v.get(f) for f in ("ms_per_step", "rank_queries", "sa_reads", "ce_windows", "algorithmic_bytes", "algorithmic_bytes_extended", "fetched_bytes",
                                                    "frac", "frac_extended")} for k, v in pk.items()}}
    lines = ["headline: %.2f M reads/s, %.1f ms per step, roofline.frac %.4f (8d bytes over the search kernels' time)" %
             (out.get("value") or 0.0, out.get("ms_per_step") or 0.0, roof.get("frac") or 0.0)]
    if pk:
        lines.append("per kernel (ms per step | rank queries | 8d bytes | fetched bytes | frac of 8 TB/s, extended): " + "; ".join(
            "%s %.1f | %.3g | %.3g | %s | %.3f, %.3f" % (k, v["ms_per_step"], v["rank_queries"], v["algorithmic_bytes"],
                                                          ("%.3g" % v["fetched_bytes"]) if v.get("fetched_bytes") else "-", v["frac"], v.get("frac_extended") or 0.0)
            for k, v in pk.items()))
    ex = out.get("extras") if isinstance(out.get("extras"), dict) else {}
    for key, tag in (("c3_pairs", "C3"), ("c5_pairs", "C5"), ("c4_pairs", "C4")):
        leg = ex.get(key)
        if not isinstance(leg, dict):
            continue
        e2e = leg.get("end_to_end") or {}
        if "error" in leg or "error" in e2e or not e2e:
            lines.append("%s: %s" % (tag, (leg.get("error") or e2e.get("error") or "no end-to-end figure")[:200]))
            continue
        pairs = g(e2e, "counts", "pairs") or g(leg, "truth_check", "pairs", default=0) or 0      # (C4: the sum over the ranks)
        secs = g(e2e, "phases_s", "Alignment + output time", default=e2e.get("wall_s"))
        rec = {"pairs": pairs, "seconds": secs, "M_pairs_per_s": e2e.get("M_pairs_per_s_reads_in_sam_out"), "loading_s": g(e2e, "phases_s", "Loading time"),
               "bwt_stage_frac": g(leg, "bwt_stage_resident", "roofline", "frac"), "dp_GCUPS": g(e2e, "dp", "GCUPS") or g(e2e, "rank0", "dp", "GCUPS"),
               # the same default-DP problems through the engine ALONE (no search kernels beside it): extras.default_dp_half_end
               "dp_isolated_GCUPS": g(ex, "default_dp_half_end", "GCUPS_kernel") if key == "c3_pairs" else None,
               "device_ms": e2e.get("device_ms"), "counts": e2e.get("counts"),
               "both_mates_at_simulated_place": g(leg, "truth_check", "pairs_with_both_mates_at_simulated_place")}
        dpc = leg.get("dp_oracle_check")
        if isinstance(dpc, dict):
            rec["dp_oracle_check"] = {k: dpc.get(k) for k in ("pairs", "records", "bit_exact", "differing_pairs", "strata", "routes_of_the_sample", "reference_stages", "seconds", "error", "skipped") if dpc.get(k) is not None}
        if "per_device" in e2e:
            rec["per_device"] = e2e["per_device"]
        sm[key] = rec
        lines.append("%s: %d pairs %.2f s = %.2f M pairs/s (FASTQ in, SAM out; index load %.2f s apart), BWT frac %s, DP %s GCUPS in the run%s" %
                     (tag, pairs, secs or 0.0, rec["M_pairs_per_s"] or 0.0, rec["loading_s"] or 0.0,
                      ("%.3f" % rec["bwt_stage_frac"]) if rec["bwt_stage_frac"] else "-", ("%.0f" % rec["dp_GCUPS"]) if rec["dp_GCUPS"] else "-",
                      (", %.0f isolated" % rec["dp_isolated_GCUPS"]) if rec["dp_isolated_GCUPS"] else ""))
        if isinstance(dpc, dict):
            lines.append("%s DP parity against the reference's own host pipeline on the same index files: %s" % (tag, ("%d pairs by route %r, %d records, bit-exact: %s (%d pairs differ)" % (
                dpc.get("pairs", 0), dpc.get("strata"), dpc.get("records", 0), dpc.get("bit_exact"), dpc.get("differing_pairs", 0))) if "records" in dpc else (dpc.get("error") or dpc.get("skipped"))))
        sd = leg.get("single_default_mode")
        if isinstance(sd, dict) and "M_reads_per_s_reads_in_sam_out" in sd:
            sm[key + "_single_default_mode"] = {"reads": sd["reads"], "M_reads_per_s": sd["M_reads_per_s_reads_in_sam_out"], "wall_s": sd["wall_s"]}
            lines.append("%s first mates, single -L %d default mode: %.2f M reads/s FASTQ in, SAM out" % (tag, sd["read_len"], sd["M_reads_per_s_reads_in_sam_out"]))
            sdc = sd.get("dp_oracle_check")
            if isinstance(sdc, dict):
                sm[key + "_single_default_mode"]["dp_oracle_check"] = {k: sdc.get(k) for k in ("reads", "records", "bit_exact", "differing_reads", "strata", "error", "skipped") if sdc.get(k) is not None}
                lines.append("%s first mates, single-end DP parity against the reference's own host pipeline: %s" % (tag, ("%d reads by route %r, %d records, bit-exact: %s (%d reads differ)" % (
                    sdc.get("reads", 0), sdc.get("strata"), sdc.get("records", 0), sdc.get("bit_exact"), sdc.get("differing_reads", 0))) if "records" in sdc else (sdc.get("error") or sdc.get("skipped"))))
    for model, r in (ex.get("repeat_models") or {}).items():
        if isinstance(r, dict) and "M_reads_per_s" in r:
            sm.setdefault("repeat_models", {})[model] = {"M_reads_per_s": r["M_reads_per_s"], "ms_per_step": r["ms_per_step"]}
            lines.append("repeat model %s: %.2f M reads/s (%.1f ms per step)" % (model, r["M_reads_per_s"], r["ms_per_step"]))
    u = ex.get("uniform_genome")
    if isinstance(u, dict) and "value" in u:
        sm["uniform_genome_M_reads_per_s"] = u["value"]
        lines.append("uniform genome: %.1f M reads/s" % u["value"])
    st = out.get("streamed")
    if isinstance(st, dict):
        sm["streamed_vs_resident"] = {"4Mi_x3": st.get("vs_resident"), "1Mi_x6": g(st, "batches_of_1Mi", "vs_resident")}
    sm["lines"] = ["[summary] " + ln for ln in lines]
    return sm

def _r(x, sig=6):
    """numbers of the compact line: six significant digits are more than any of them is measured to"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


COMPACT_LIMIT = 8192                                                    # bytes; tests/test_bench_line_cpu.py holds the line to it


def compact_line(out, full_path=None):
    """The ONE stdout line.  The full object (tens of kilobytes: every leg's counts, checks and texts) goes to a side file and to stderr; the line the
    driver parses carries the contract's keys, the roofline of the search kernels with every kernel's own counts, the CPU baseline and a summary of the
    other legs -- and stays under COMPACT_LIMIT bytes whatever the legs add (a 27 KB line was stored unparsed)."""
    def g(d, *path, default=None):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return default
            d = d[k]
        return d
    cfg = out.get("config") or {}
    roof = out.get("roofline") or {}
    cpu = out.get("cpu_baseline") or {}
    sm = out.get("summary") or summary_of(out)
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    wl = cfg.get("workload") or ""
    line["config"] = {"workload": wl if len(wl) <= 400 else wl[:397] + "...",
                      **{k: cfg.get(k) for k in ("genome_bp", "reads_per_gpu", "read_len", "max_mismatch", "max_output_per_read", "repeat_frac", "repeat_model",
                                                 "parallelism", "lookup_chars") if k in cfg}}
    pk = {}
    for k, v in (roof.get("per_kernel") or {}).items():
        pk[k] = {f: v.get(f) for f in ("ms_per_step", "rank_queries", "rank_queries_needed", "sa_reads", "ce_windows", "algorithmic_bytes", "fetched_bytes", "frac",
                                       "frac_issued", "frac_extended") if v.get(f) is not None}
        rj = v.get("rejected")
        if isinstance(rj, dict):
            pk[k]["rejected"] = {f: rj.get(f) for f in ("candidates", "accepted", "share_within_32_adjacent_chars", "share_within_32_adjacent_chars_backward_steps_only",
                                                         "share_handed_over_in_backward_steps")}
    line["roofline"] = {"bound": roof.get("bound"), "achieved": roof.get("achieved"), "peak": roof.get("peak"), "unit": roof.get("unit"), "frac": roof.get("frac"),
                        "traffic": roof.get("traffic"),
                        "kernel": "s3::firstKernel + s3::unitKernel + s3::ceKernel (one search pass; dominant: s3::%sKernel)" % (roof.get("dominant_kernel") or "unit"),
                        "launches_per_step": roof.get("launches_per_step"), "kernel_ms_per_step": roof.get("kernel_ms_per_step"),
                        "algorithmic_bytes_per_launch": roof.get("algorithmic_bytes_per_launch"),
                        "rank_queries_algorithmic": roof.get("rank_queries_algorithmic"), "rank_queries_issued": roof.get("rank_queries_per_step"),
                        "peak_measured": roof.get("peak_measured"), "frac_of_measured": roof.get("frac_of_measured"), "frac_extended": roof.get("frac_extended"),
                        "traffic_measured_at_commit": roof.get("traffic_measured_at_commit"),
                        "random_line_rate": roof.get("random_line_rate"), "per_kernel": pk}
    ref = cpu.get("reference") if isinstance(cpu.get("reference"), dict) else None
    line["cpu_baseline"] = {"value": cpu.get("value"), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
                            "sample": (cpu.get("sample") or "")[:200], "reads_per_s_per_core": cpu.get("reads_per_s_per_core")}
    if ref:
        line["cpu_baseline"]["reference"] = {"value": ref.get("value"), "unit": ref.get("unit"), "cores": ref.get("cores"), "kind": ref.get("kind"),
                                             "port_over_reference": ref.get("port_over_reference")} if "value" in ref else {"error": str(ref.get("error"))[:120]}
    line["parity"] = {"truth_check_ok": g(out, "truth_check", "ok"), "oracle_check_reads": g(out, "oracle_check", "reads"),
                      "oracle_check_bit_exact": g(out, "oracle_check", "bit_exact"), "streamed_same_as_resident": g(out, "streamed", "same_results_as_resident")}
    line["occurrences"] = out.get("occurrences")
    s2 = {}
    for key in ("c3_pairs", "c5_pairs", "c4_pairs"):
        leg = sm.get(key)
        if isinstance(leg, dict):
            s2[key] = {f: leg.get(f) for f in ("pairs", "seconds", "M_pairs_per_s", "bwt_stage_frac", "dp_GCUPS", "dp_isolated_GCUPS", "device_ms",
                                               "both_mates_at_simulated_place", "dp_oracle_check") if leg.get(f) is not None}
            if isinstance(s2[key].get("dp_oracle_check"), dict):
                s2[key]["dp_oracle_check"] = {f: v for f, v in s2[key]["dp_oracle_check"].items() if f in ("pairs", "records", "bit_exact", "differing_pairs", "error", "skipped")}
        else:
            ex = g(out, "extras", key)
            if isinstance(ex, dict):
                s2[key] = {"error": str(ex.get("error") or g(ex, "end_to_end", "error") or "no end-to-end figure")[:160]}
    for key in ("repeat_models", "uniform_genome_M_reads_per_s", "streamed_vs_resident", "c3_pairs_single_default_mode"):
        if sm.get(key) is not None:
            s2[key] = sm[key]
    if isinstance(g(s2, "c3_pairs_single_default_mode", "dp_oracle_check"), dict):
        s2["c3_pairs_single_default_mode"] = dict(s2["c3_pairs_single_default_mode"], dp_oracle_check={f: v for f, v in s2["c3_pairs_single_default_mode"]["dp_oracle_check"].items() if f != "strata"})
    if isinstance(g(out, "extras", "error"), str):
        s2["extras_error"] = out["extras"]["error"][:160]
    line["summary"] = s2
    line["wall_s_total"] = out.get("wall_s_total")
    line["full_object"] = full_path
    line = _r(line)
    txt = json.dumps(line, separators=(",", ":"))
    # a leg that grew must not cost the record: drop the least needed parts until the line fits
    for drop in (("summary", "c3_pairs", "device_ms"), ("summary", "c5_pairs", "device_ms"), ("summary", "c4_pairs"), ("roofline", "random_line_rate"),
                 ("summary", "c3_pairs_single_default_mode"), ("summary", "repeat_models"), ("summary",)):
        if len(txt) < COMPACT_LIMIT:
            break
        d = line
        for k in drop[:-1]:
            d = d.get(k) if isinstance(d, dict) else None
        if isinstance(d, dict):
            d.pop(drop[-1], None)
        txt = json.dumps(line, separators=(",", ":"))
    return txt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-bp", type=int, default=int(os.environ.get("S3DP_BENCH_GENOME_BP", 3_100_000_000)))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("S3DP_BENCH_READS", 10_000_000)), help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--mismatch", type=int, default=4)
    ap.add_argument("--max-out", type=int, default=1000)
    ap.add_argument("--repeat-frac", type=float, default=0.5, help="share of the genome planted with repeat families (GRCh38 is about one half)")
    ap.add_argument("--repeat-model", default="grch38-like", choices=sorted(REPEAT_MODELS), help="which repeat structure is planted (robustness runs; the headline is grch38-like)")
    ap.add_argument("--uniform", type=int, default=1, help="N = 1 only: also run the repeat-free genome of round 1 (reported under extras)")
    ap.add_argument("--streaming", type=int, default=1, help="also time the streamed form (uploads and downloads inside the timed region)")
    ap.add_argument("--stream-batch", type=int, default=1 << 22, help="reads per streamed batch (the reference reads 1 Mi at a time, SOAP3-DP.cu:394; a launch here lasts as long as its longest unit, so larger batches pay)")
    ap.add_argument("--stream-depth", type=int, default=3, help="batches in flight in the streamed form")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU-baseline work (0 = skip)")
    ap.add_argument("--cpu-reference", type=int, default=200_000, help="reads of the leg that times the reference's own CPU path (oracle/_ref/ref_harness) beside the port (0: skip)")
    ap.add_argument("--check", type=int, default=16384, help="reads verified against the oracle after the run (a quarter each: cut at MaxOutputPerRead, hundreds of occurrences, a few, the first reads)")
    ap.add_argument("--check-budget", type=float, default=20.0, help="seconds the oracle check may take (it stops taking new blocks of 32 reads after that)")
    ap.add_argument("--index-dir", default=None, help="keep / reuse the index files of the benchmark genome in this directory")
    ap.add_argument("--build-only", type=int, default=0, help="stop after the index files are written (with --index-dir)")
    ap.add_argument("--c3-pairs", type=int, default=10_000_000, help="extras: BASELINE.json configs[2] -- 2 x 150 bp pairs through the command line with the DP stages on (0: skip)")
    ap.add_argument("--c5-pairs", type=int, default=10_000_000, help="extras: configs[4]'s mixture (5 %% of the pairs damaged in both mates: deep DP) through the command line (0: skip)")
    ap.add_argument("--c4-pairs", type=int, default=10_000_000, help="N > 1: BASELINE.json configs[3] -- pairs PER GPU, every rank's shard through the command line on its own device (0: skip)")
    ap.add_argument("--dp-check", type=int, default=2048, help="extras: pairs of the C3 / C5 runs, stratified by route, whose records are compared with the reference's own host "
                                                               "pipeline (oracle/_ref/ref_harness) on the same index files (0: skip)")
    ap.add_argument("--se150-reads", type=int, default=2_000_000, help="extras: 150 bp single-end reads, kernel-only rate (0: skip)")
    ap.add_argument("--exact-rank", type=int, default=1, help="count the reference algorithm's rank queries on this input with one walk-mode device pass")
    ap.add_argument("--robustness", default="young", help="extras: comma-separated repeat models (young, old, satellite-heavy) run at --robustness-reads reads each ('' = none)")
    ap.add_argument("--robustness-reads", type=int, default=10_000_000, help="reads of a robustness leg (the headline's count: a smaller batch pays more for the tails of its launches and would not compare)")
    ap.add_argument("--extras", type=int, default=1, help="after the timed region (N = 1): paired-end 2x150 BWT stage and DP kernel throughput")
    ap.add_argument("--launch-selftest", type=int, default=0, help="N > 1 plumbing only, no GPU: the ranks shard a count, reduce it over gloo and print the line "
                                                                    "(tests/test_bench_launcher_cpu.py)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a))                               # this process becomes the launcher: it never touches a GPU
    if a.launch_selftest:
        raise SystemExit(launch_selftest_rank(a))

    import torch
    import s3dp
    from s3dp import builder

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (bench.py --gpus N starts its own ranks when no launcher has set WORLD_SIZE)" % (a.gpus, world))
    if not torch.cuda.is_available() or s3dp.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    # S3DP_BENCH_SHARE_GPU=1 (test aid, tests/test_gpu_group.py): every rank on device 0, gloo instead of RCCL (RCCL wants one device per rank) -- the
    # whole N > 1 path of this file (index built once per node and mapped by the other ranks, barriers, max-over-ranks timing, the counters' reduce) on a
    # box with ONE GPU; the figures of such a run mean nothing
    share = os.environ.get("S3DP_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
        COLL_DEV.append(None)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    index_dir = None
    if world > 1:
        import tempfile
        import torch.distributed as dist_mod
        dist = dist_mod
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
        box = [tempfile.mkdtemp(prefix="s3dp_bench_") if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        index_dir = box[0]

    t_all = time.perf_counter()
    peaks = measured_peaks(dev, local)
    log("measured: copy %.0f GB/s, random 64-byte lines %.1f G/s" % (peaks["copy_GBps"], peaks["random_Glines_per_s"]))
    label = "repeat-rich" if a.repeat_frac > 0 else "uniform"
    res, keep = run_workload(a, s3dp, builder, dev, local, rank, world, dist, peaks, a.repeat_frac, label, index_dir)
    out = {"metric": "M reads/sec aligned (100 bp SE, GRCh38-scale synthetic genome)", "value": res["value"], "unit": "M reads/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u32", "data": "synthetic"}
    res["config"]["workload"] = ("synthetic genome of GRCh38's size with GRCh38-like repeat families planted (%s), full-SA 2BWT+GPU-2BWT index in HBM; "
                                 "100 bp single-end reads, 0-4 substitutions (55/25/12/5/3 %%), <=4 mismatch all-best ladder (-s 4 -h 2), DP off; "
                                 "BASELINE.json configs[1] with GRCh38 replaced by a synthetic genome of the same scale" %
                                 ("repeat_frac %.2f" % a.repeat_frac if a.repeat_frac > 0 else "none: uniform ACGT"))
    out.update({k: v for k, v in res.items() if k not in ("value", "ms_per_step")})
    out["measured_peaks"] = peaks

    if rank == 0:
        if world == 1 and a.extras:
            try:
                out["extras"] = extras(s3dp, keep["ix"], keep["hi"], keep["codes"], dev, rank, a)
            except Exception as e:                                      # secondary figures must never cost the headline line
                out["extras"] = {"error": repr(e)}
    keep["ix"].close()
    import gc
    if world > 1 and a.c4_pairs > 0 and keep.get("prefix"):
        # the paired workload on N GPUs (BASELINE.json configs[3]): every rank's shard through the command line on its own device, after the ranks have freed
        # their resident indexes (the children bring their own)
        keep.pop("q", None)
        gc.collect()
        torch.cuda.empty_cache()
        try:
            c4 = c4_leg(s3dp, keep["hi"], keep["codes"], dev, local, rank, world, dist, a, keep["prefix"], COLL_DEV[0] if COLL_DEV else dev)
        except Exception as e:
            c4 = {"error": repr(e)}
        if rank == 0:
            out.setdefault("extras", {})["c4_pairs"] = c4
    keep.clear()
    gc.collect()
    torch.cuda.empty_cache()
    ex = out.get("extras") if isinstance(out.get("extras"), dict) else {}
    shared_ix = ex.pop("_index_dir", None)
    for key, leg in (("_c3_cli", "c3_pairs"), ("_c5_cli", "c5_pairs")):
        cli = ex.pop(key, None)
        if cli is not None:
            try:
                cli()                                                   # fills extras.<leg>.end_to_end and .truth_check
            except Exception as e:
                ex[leg]["end_to_end"] = {"error": repr(e)}
            gc.collect()
            torch.cuda.empty_cache()
    if shared_ix:
        import shutil
        shutil.rmtree(shared_ix, ignore_errors=True)
    if rank == 0 and world == 1 and a.uniform and a.repeat_frac > 0:
        try:
            a2 = argparse.Namespace(**vars(a))
            a2.check = min(a.check, 1024)
            a2.streaming = 0
            ures, ukeep = run_workload(a2, s3dp, builder, dev, local, rank, world, dist, peaks, 0.0, "uniform", want_cpu=False)
            ukeep["ix"].close()
            out.setdefault("extras", {})["uniform_genome"] = ures
        except Exception as e:
            out.setdefault("extras", {})["uniform_genome"] = {"error": repr(e)}
    if rank == 0 and world == 1 and a.repeat_frac > 0 and a.robustness:
        # robustness of the heavy-read machinery's thresholds: other repeat structures of the same size at a reduced read count (own genome, own index each)
        out.setdefault("extras", {})["repeat_models"] = {}
        for model in [m for m in a.robustness.split(",") if m and m != a.repeat_model]:
            try:
                a3 = argparse.Namespace(**vars(a))
                a3.repeat_model, a3.reads, a3.streaming, a3.exact_rank, a3.check = model, min(a.reads, a.robustness_reads), 0, 0, min(a.check, 2048)
                a3.steps, a3.warmup, a3.check_budget = min(a.steps, 3), 1, min(a.check_budget, 10.0)
                mres, mkeep = run_workload(a3, s3dp, builder, dev, local, rank, world, dist, peaks, a.repeat_frac, "repeat-rich", want_cpu=False)
                mkeep["ix"].close()
                mkeep.clear()
                out["extras"]["repeat_models"][model] = {"M_reads_per_s": mres["value"], "ms_per_step": mres["ms_per_step"], "reads": a3.reads,
                                                         "truth_check": mres.get("truth_check"), "oracle_check": mres.get("oracle_check"),
                                                         "rank_queries_per_read": mres["rank_queries_per_read"], "occurrences": mres["occurrences"],
                                                         "kernel_ms_per_step": mres["roofline"]["kernel_ms_per_step"], "frac": mres["roofline"]["frac"],
                                                         "repeat_families": mres["config"]["repeat_families"]}
            except Exception as e:
                out["extras"]["repeat_models"][model] = {"error": repr(e)}
            gc.collect()
            torch.cuda.empty_cache()
    if rank == 0:
        out["wall_s_total"] = time.perf_counter() - t_all
        # The full object is tens of kilobytes: it goes to a side file (and to stderr); stdout gets ONE compact line (compact_line: under 8 KB) that the driver
        # parses, and the summary's lines are the last lines of stderr.
        out["summary"] = summary_of(out)
        full_path = os.environ.get("S3DP_BENCH_FULL", "bench_full.json")
        full_txt = json.dumps(out)
        try:
            with open(full_path, "w") as f:
                f.write(full_txt + "\n")
            if os.path.isdir("gpurun_out"):
                with open(os.path.join("gpurun_out", "bench_full.json"), "w") as f:
                    f.write(full_txt + "\n")
        except OSError as e:
            log("could not write %s: %r" % (full_path, e))
            full_path = None
        sys.stderr.write("[bench-full] " + full_txt + "\n")
        for ln in out["summary"]["lines"]:
            log(ln)
        print(compact_line(out, full_path), flush=True)
    if index_dir and rank == 0:
        import shutil
        dist.barrier()
        shutil.rmtree(index_dir, ignore_errors=True)
    elif dist is not None:
        dist.barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
