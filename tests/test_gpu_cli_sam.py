"""-m gpu: the drop-in `soap3-dp` command line (host C++ above the C-ABI) must write, read for read, the SAM records the
reference's CPU path wrote (tests/golden/*/sam_s*.sam.gz, produced by oracle/_ref/ref_harness with the reference's own
hostKernel + OCCOutputSAMAPI): position, strand, CIGAR, MAPQ, X0/X1/XM and the XA:Z list."""
import gzip
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "soap3-dp_amd", "soap3-dp")


def sam_records(text):
    hdr, rec = [], {}
    for l in text.split("\n"):
        if not l:
            continue
        if l.startswith("@"):
            if not l.startswith("@PG"):
                hdr.append(l)
        else:
            rec[l.split("\t", 1)[0]] = l
    return hdr, rec


# (fixture, extra options): -h 2 all best (default), -h 1 all valid, -h 3 unique best, -h 4 random best (+ -p: NM/MD tags)
VARIANTS = [("s2", ["-s", "2"]), ("s4", ["-s", "4"]), ("h1_s2", ["-s", "2", "-h", "1"]), ("h1_s0", ["-s", "0", "-h", "1"]),
            ("h3_s3", ["-s", "3", "-h", "3"]), ("h4_s3", ["-s", "3", "-h", "4", "-p"])]


@pytest.mark.parametrize("name,maxlen", [("g1", 100), ("g2", 120)])
@pytest.mark.parametrize("variant,opts", VARIANTS)
@pytest.mark.parametrize("chunk", ["", "192"])
def test_cli_sam_matches_reference_cpu_path(tmp_path, name, maxlen, variant, opts, chunk):
    """chunk: S3DP_CHUNK_READS -- the input is streamed in chunks of that many reads (several batches in flight on their own HIP streams, a loader
    thread and a writer thread around them); '' = the default size, one chunk for these files"""
    assert os.path.exists(BIN), "soap3-dp binary not built"
    fq = tmp_path / (name + ".fq")                      # the read-group id is the query file name as typed
    shutil.copy(os.path.join(GOLDEN, name, "reads.fq"), fq)
    env = dict(os.environ)
    if chunk:
        env["S3DP_CHUNK_READS"] = chunk
    out = subprocess.run([BIN, "single", os.path.join(GOLDEN, name, "index"), name + ".fq", "-L", str(maxlen)] + opts + ["-b", "2",
                          "-o", "out"], cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    mine_h, mine = sam_records(open(tmp_path / "out.gout.1").read())
    ref_h, ref = sam_records(gzip.open(os.path.join(GOLDEN, name, "sam_%s.sam.gz" % variant), "rt").read())
    assert mine_h == ref_h
    assert set(mine) == set(ref)
    bad = [(k, ref[k], mine[k]) for k in ref if ref[k] != mine[k]]
    assert not bad, "first differing record:\n%s\n%s" % (bad[0][1], bad[0][2])
    assert os.path.exists(tmp_path / "out.done")


PROCESS_KNOBS = [
    {"S3DP_VERBOSE": "1"},                                            # the timeline and per-launch laps go to stderr / stdout lines, never into the SAM file
    {"S3DP_BLOCKING_SYNC": "1"},                                      # host threads sleep while they wait for the device
    {"S3DP_H2D_PIECE_MB": "1"}, {"S3DP_H2D_PIECE_MB": "0"},           # index upload in 1 MB pieces / in one piece
    {"S3DP_STREAM_BLOCKS_PER_CU": "1", "S3DP_CHUNK_READS": "192"},    # grid widths of the batches that share the device
    {"S3DP_STREAM_BLOCKS_PER_CU": "8", "S3DP_CHUNK_READS": "192"},
    {"S3DP_HOST_THREADS": "1", "S3DP_CHUNK_READS": "192"},
    {"S3DP_STREAM_DEPTH": "1", "S3DP_CHUNK_READS": "192"},            # batches in flight
    {"S3DP_STREAM_DEPTH": "6", "S3DP_CHUNK_READS": "192"},
    {"GPU_MAX_HW_QUEUES": "2", "S3DP_CHUNK_READS": "192"},            # the HIP runtime's variable (the library asks for 16 unless it is set)
    {"S3DP_GENERIC_PARSER": "1"},                                     # every record through the general parser
]


@pytest.mark.parametrize("knobs", PROCESS_KNOBS, ids=lambda k: ",".join("%s=%s" % (a[5:], b) for a, b in k.items()))
def test_cli_process_knobs_do_not_change_the_sam_file(tmp_path, knobs):
    """Knobs read once per process (device flags, upload piece size, stream grid widths, verbosity, host threads): same records as the reference's CPU path"""
    name, maxlen, variant, opts = "g2", 120, "s4", ["-s", "4"]
    shutil.copy(os.path.join(GOLDEN, name, "reads.fq"), tmp_path / (name + ".fq"))
    out = subprocess.run([BIN, "single", os.path.join(GOLDEN, name, "index"), name + ".fq", "-L", str(maxlen)] + opts + ["-b", "2", "-o", "out"],
                         cwd=tmp_path, capture_output=True, text=True, timeout=600, env=dict(os.environ, **knobs))
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    mine_h, mine = sam_records(open(tmp_path / "out.gout.1").read())
    ref_h, ref = sam_records(gzip.open(os.path.join(GOLDEN, name, "sam_%s.sam.gz" % variant), "rt").read())
    assert mine_h == ref_h and mine == ref
    if "S3DP_VERBOSE" in knobs:
        assert "[s3dp]" in out.stderr                                   # the timeline was printed


@pytest.mark.parametrize("variant,opts", [("s4", ["-s", "4"]), ("h1_s2p", ["-s", "2", "-h", "1", "-p"])])
def test_cli_sam_at_the_ends_of_the_text(tmp_path, variant, opts):
    """fixture g6: reads that hug or overhang the ends of the text and a sequence boundary (boundary trimming, soft clips, padded lookup keys)"""
    name = "g6"
    shutil.copy(os.path.join(GOLDEN, name, "reads.fq"), tmp_path / "g6.fq")
    out = subprocess.run([BIN, "single", os.path.join(GOLDEN, name, "index"), "g6.fq", "-L", "100"] + opts + ["-b", "2", "-o", "out"], cwd=tmp_path, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    mine_h, mine = sam_records(open(tmp_path / "out.gout.1").read())
    ref_h, ref = sam_records(gzip.open(os.path.join(GOLDEN, name, "sam_%s.sam.gz" % variant), "rt").read())
    assert mine_h == ref_h and set(mine) == set(ref)
    bad = [(k, ref[k], mine[k]) for k in ref if ref[k] != mine[k]]
    assert not bad, "%d differing records, first:\n%s\n%s" % (len(bad), bad[0][1], bad[0][2])


# long-read mode: -L 200, reads of 121..200 bases
LONG_VARIANTS = [("s3", ["-s", "3"]), ("s1", ["-s", "1", "-p"]), ("h1_s2", ["-s", "2", "-h", "1"]), ("h3_s2", ["-s", "2", "-h", "3"]),
                 ("h4_s3", ["-s", "3", "-h", "4"])]


@pytest.mark.parametrize("variant,opts", LONG_VARIANTS)
def test_cli_sam_long_read_mode(tmp_path, variant, opts):
    name = "g3"
    fq = tmp_path / (name + ".fq")
    shutil.copy(os.path.join(GOLDEN, name, "reads.fq"), fq)
    out = subprocess.run([BIN, "single", os.path.join(GOLDEN, name, "index"), name + ".fq", "-L", "200"] + opts + ["-b", "2", "-o", "out"],
                         cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    mine_h, mine = sam_records(open(tmp_path / "out.gout.1").read())
    ref_h, ref = sam_records(gzip.open(os.path.join(GOLDEN, name, "sam_%s.sam.gz" % variant), "rt").read())
    assert mine_h == ref_h
    assert set(mine) == set(ref)
    bad = [(k, ref[k], mine[k]) for k in ref if ref[k] != mine[k]]
    assert not bad, "%d differing records, first:\n%s\n%s" % (len(bad), bad[0][1], bad[0][2])


# paired-end, DP stages off: pair <index> <r1> <r2> -u 500 -v 200 -s k
PAIR_VARIANTS = [("s3", ["-s", "3"]), ("s1", ["-s", "1", "-p"]), ("s4", ["-s", "4"]), ("h1_s2", ["-s", "2", "-h", "1"]),
                 ("h3_s3", ["-s", "3", "-h", "3"]), ("h4_s2", ["-s", "2", "-h", "4"])]


def sam_pair_records(text):
    hdr, rec = [], {}
    for l in text.split("\n"):
        if not l:
            continue
        if l.startswith("@"):
            if not l.startswith("@PG"):
                hdr.append(l)
        else:
            f = l.split("\t", 2)
            rec[(f[0], int(f[1]) & 0xC0)] = l             # (name, first / second in pair)
    return hdr, rec


@pytest.mark.parametrize("variant,opts", PAIR_VARIANTS)
@pytest.mark.parametrize("chunk", ["", "640"])
def test_cli_sam_pairs(tmp_path, variant, opts, chunk):
    name = "g4"
    for k in ("1", "2"):
        shutil.copy(os.path.join(GOLDEN, name, "reads_%s.fq" % k), tmp_path / ("%s_%s.fq" % (name, k)))
    env = dict(os.environ)
    if chunk:
        env["S3DP_CHUNK_READS"] = chunk
    out = subprocess.run([BIN, "pair", os.path.join(GOLDEN, name, "index"), name + "_1.fq", name + "_2.fq", "-L", "100", "-u", "500", "-v", "200"] + opts +
                         ["-b", "2", "-o", "out"], cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    mine_h, mine = sam_pair_records(open(tmp_path / "out.gout.1").read())
    ref_h, ref = sam_pair_records(gzip.open(os.path.join(GOLDEN, name, "pe_%s.sam.gz" % variant), "rt").read())
    assert mine_h == ref_h
    assert set(mine) == set(ref)
    bad = [(k, ref[k], mine[k]) for k in ref if ref[k] != mine[k]]
    assert not bad, "%d differing records, first:\n%s\n%s" % (len(bad), bad[0][1], bad[0][2])


# BASELINE config C3's shape with the DP stages off: 2 x 150 bp, -u 500 -v 1 (fixture g7; with the DP stages on: tests/test_gpu_dp_pipeline.py)
C3_VARIANTS = [("s3", ["-s", "3"]), ("h1_s3", ["-s", "3", "-h", "1"]), ("h3_s3", ["-s", "3", "-h", "3"]), ("h4_s3", ["-s", "3", "-h", "4", "-p"])]


@pytest.mark.parametrize("variant,opts", C3_VARIANTS)
def test_cli_sam_pairs_c3_shape(tmp_path, variant, opts):
    name = "g7"
    for k in ("1", "2"):
        shutil.copy(os.path.join(GOLDEN, name, "reads_%s.fq" % k), tmp_path / ("%s_%s.fq" % (name, k)))
    out = subprocess.run([BIN, "pair", os.path.join(GOLDEN, name, "index"), name + "_1.fq", name + "_2.fq", "-L", "150", "-u", "500", "-v", "1"] + opts +
                         ["-b", "2", "-o", "out"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    mine_h, mine = sam_pair_records(open(tmp_path / "out.gout.1").read())
    ref_h, ref = sam_pair_records(gzip.open(os.path.join(GOLDEN, name, "pe_%s.sam.gz" % variant), "rt").read())
    assert mine_h == ref_h
    assert set(mine) == set(ref)
    bad = [(k, ref[k], mine[k]) for k in ref if ref[k] != mine[k]]
    assert not bad, "%d differing records, first:\n%s\n%s" % (len(bad), bad[0][1], bad[0][2])


def test_cli_default_mode_single_reads_with_dp(tmp_path):
    """`soap3-dp single <index> <reads>` without -s: BWT stage at Soap3MisMatchAllow = 2 mismatches into .gout.1, the reads it leaves unaligned
    through seeding + DP into .dpout.1.  (Property test kept from round 1; the record-for-record comparison with the reference's own engines is tests/test_gpu_dp_pipeline.py.)  The check is semantic -- every read is reported once, the
    damaged reads (indels, 5 substitutions) come back at their true place with a CIGAR of the expected shape."""
    import re
    import numpy as np
    name = "g1"
    rng = np.random.default_rng(8)
    chroms = {}
    cur = None
    for l in open(os.path.join(GOLDEN, name, "genome.fa")):
        if l.startswith(">"):
            cur = l[1:].split()[0]
            chroms[cur] = []
        else:
            chroms[cur].append(l.strip())
    chroms = {k: "".join(v) for k, v in chroms.items()}
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    L, reads, truth = 100, [], []
    names = sorted(chroms)
    while len(reads) < 400:
        ch = names[int(rng.integers(0, len(names)))]
        st = int(rng.integers(0, len(chroms[ch]) - L - 10))
        seg = chroms[ch][st:st + L + 10]
        if "N" in seg:
            continue
        kind = len(reads) % 4
        if kind == 0:
            s, shape = seg[:L], "exact"
        elif kind == 1:
            c = int(rng.integers(30, 70))
            s, shape = seg[:c] + seg[c + 3:L + 3], "D"
        elif kind == 2:
            c = int(rng.integers(30, 70))
            s, shape = seg[:c] + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=2)) + seg[c:L - 2], "I"
        else:
            b = list(seg[:L])
            for q in rng.choice(np.arange(5, L - 5), size=5, replace=False):
                b[q] = "ACGT"[("ACGT".index(b[q]) + 1 + int(rng.integers(0, 3))) % 4]
            s, shape = "".join(b), "sub"
        neg = bool(rng.integers(0, 2))
        if neg:
            s = "".join(comp[c] for c in reversed(s))
        reads.append(s)
        truth.append((ch, st + 1, neg, shape))
    with open(tmp_path / "r.fq", "w") as f:
        for i, s in enumerate(reads):
            f.write("@q%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    out = subprocess.run([BIN, "single", os.path.join(GOLDEN, name, "index"), "r.fq", "-L", "100", "-o", "out"], cwd=tmp_path, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    gout = [l.rstrip("\n").split("\t") for l in open(tmp_path / "out.gout.1") if not l.startswith("@")]
    dpout = [l.rstrip("\n").split("\t") for l in open(tmp_path / "out.dpout.1") if not l.startswith("@")]
    seen = [f[0] for f in gout] + [f[0] for f in dpout]
    assert sorted(seen) == sorted("q%d" % i for i in range(len(reads))), "every read exactly once over .gout.1 and .dpout.1"
    by_name = {f[0]: ("bwt", f) for f in gout}
    by_name.update({f[0]: ("dp", f) for f in dpout})
    ok = {"exact": 0, "D": 0, "I": 0, "sub": 0}
    tot = {"exact": 0, "D": 0, "I": 0, "sub": 0}
    for i, (ch, pos, neg, shape) in enumerate(truth):
        src, f = by_name["q%d" % i]
        tot[shape] += 1
        if shape == "exact":
            assert src == "bwt" and f[5] == "100M"
            if "X0:i:1" in f:                                   # not inside one of the planted repeats
                assert f[2] == ch and int(f[3]) == pos
            ok[shape] += 1
            continue
        assert src == "dp"                                     # 3-base deletion / 2-base insertion / 5 substitutions never pass a 2-mismatch BWT stage
        if int(f[1]) & 4:
            continue
        ops = re.sub(r"\d+", "", f[5])
        good = f[2] == ch and abs(int(f[3]) - pos) <= 3 and (int(f[1]) & 16 != 0) == neg
        if shape == "D":
            good &= ops == "MDM" and "XO:i:1" in f and "XG:i:3" in f
        elif shape == "I":
            good &= ops == "MIM" and "XO:i:1" in f and "XG:i:2" in f
        else:
            good &= ops == "M" and "XM:i:5" in f
        ok[shape] += good
    assert ok["D"] >= 0.8 * tot["D"] and ok["I"] >= 0.8 * tot["I"] and ok["sub"] >= 0.6 * tot["sub"], (ok, tot)   # repeats and spoiled seeds take their share
    assert "Number of reads aligned by DP" in out.stdout


def test_cli_default_mode_pairs_with_dp(tmp_path):
    """`soap3-dp pair <index> <r1> <r2> -u 500 -v 100` without -s: the BWT ladder pairs what it can (.gout.1); pairs with one damaged end go
    through the default DP (mate rescue next to the aligned end), pairs with two damaged ends through the deep DP (seeding both ends), both into
    .dpout.1; what is left goes to .unpair.  The record writers are pinned (tests/golden/dp2), the engines are not: the check is semantic --
    every pair is reported once as two records, rescued mates sit at their true place with a CIGAR of the expected shape and proper-pair flags."""
    import re
    import numpy as np
    name = "g1"
    rng = np.random.default_rng(21)
    chroms, cur = {}, None
    for l in open(os.path.join(GOLDEN, name, "genome.fa")):
        if l.startswith(">"):
            cur = l[1:].split()[0]
            chroms[cur] = []
        else:
            chroms[cur].append(l.strip())
    chroms = {k: "".join(v) for k, v in chroms.items()}
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = lambda s: "".join(comp[c] for c in reversed(s))
    L = 100

    def damage(seg, kind):
        if kind == "exact":
            return seg[:L]
        if kind == "D":
            c = int(rng.integers(30, 70))
            return seg[:c] + seg[c + 3:L + 3]
        if kind == "I":
            c = int(rng.integers(30, 70))
            return seg[:c] + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=2)) + seg[c:L - 2]
        b = list(seg[:L])
        for q in rng.choice(np.arange(5, L - 5), size=6, replace=False):
            b[q] = "ACGT"[("ACGT".index(b[q]) + 1 + int(rng.integers(0, 3))) % 4]
        return "".join(b)
    names = sorted(chroms)
    r1s, r2s, truth = [], [], []
    kinds = [("exact", "exact"), ("exact", "D"), ("I", "exact"), ("exact", "sub"), ("D", "I"), ("sub", "D"), ("exact", "far"), ("exact", "junk")]
    while len(r1s) < 480:
        ch = names[int(rng.integers(0, len(names)))]
        ins = int(rng.integers(250, 450))
        st = int(rng.integers(0, len(chroms[ch]) - ins - 20))
        kl, kr = kinds[len(r1s) % len(kinds)]
        if kr == "far":                                       # the mate comes from 5 kb further on: both ends align, no proper pairing exists
            ins += 5000
            if st + ins + 10 > len(chroms[ch]):
                continue
        left, right = chroms[ch][st:st + L + 10], chroms[ch][st + ins - L:st + ins + 10]
        if "N" in left or "N" in right or len(right) < L + 10:
            continue
        if kr == "junk":
            right = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=L + 10))
        a, b = damage(left, kl), rc(damage(right, "exact" if kr in ("far", "junk") else kr))
        swap = bool(rng.integers(0, 2))                       # which file holds the '+' end
        r1s.append(b if swap else a); r2s.append(a if swap else b)
        truth.append((ch, st + 1, st + ins - L + 1, swap, kl, kr))
    for fn, rr, tag in (("r1.fq", r1s, 1), ("r2.fq", r2s, 2)):
        with open(tmp_path / fn, "w") as f:
            for i, s in enumerate(rr):
                f.write("@p%d/%d\n%s\n+\n%s\n" % (i, tag, s, "I" * len(s)))
    out = subprocess.run([BIN, "pair", os.path.join(GOLDEN, name, "index"), "r1.fq", "r2.fq", "-u", "500", "-v", "100", "-L", "100", "-o", "out"], cwd=tmp_path,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    recs = {}
    for src in ("gout.1", "dpout.1", "unpair"):
        for l in open(tmp_path / ("out." + src)):
            if l.startswith("@"):
                continue
            f = l.rstrip("\n").split("\t")
            recs.setdefault(f[0], []).append((src, f))
    assert sorted(recs) == sorted("p%d" % i for i in range(len(r1s)))
    assert all(len(v) == 2 and v[0][0] == v[1][0] for v in recs.values()), "every pair: two records in one file"
    ok, tot = {}, {}
    for i, (ch, lp, rp, swap, kl, kr) in enumerate(truth):
        src, _ = recs["p%d" % i][0]
        key = "%s/%s" % (kl, kr)
        tot[key] = tot.get(key, 0) + 1
        fl = {int(f[1]) & 0xC0: f for _, f in recs["p%d" % i]}
        fa, fb = (fl[0x80], fl[0x40]) if swap else (fl[0x40], fl[0x80])      # record of the '+' (left) end, of the '-' (right) end
        if key == "exact/exact":
            assert src == "gout.1"
        else:
            assert src != "gout.1"
        if kr in ("far", "junk"):
            # no stage can pair these: each end on its own in .unpair -- the first keeps its BWT hit, the second goes through the single-end DP
            # (a junk mate can scrape past the DP cutoff inside the insert window, a far mate can sit in a planted repeat: counted, not asserted)
            good = src == "unpair" and not int(fa[1]) & 2 and not int(fa[1]) & 4 and fa[2] == ch and int(fa[3]) == lp and fa[5] == "100M"
            if kr == "far":
                good = good and not int(fb[1]) & 4 and fb[2] == ch and int(fb[3]) == rp and int(fb[1]) & 16
            else:
                good = good and int(fb[1]) & 4 and int(fa[1]) & 8
            ok[key] = ok.get(key, 0) + bool(good)
            continue
        if src == "unpair":
            continue
        good = int(fa[1]) & 2 and int(fb[1]) & 2 and fa[2] == ch and fb[2] == ch and abs(int(fa[3]) - lp) <= 3 and abs(int(fb[3]) - rp) <= 3
        good = good and not int(fa[1]) & 16 and int(fb[1]) & 16
        for f, k in ((fa, kl), (fb, kr)):
            ops = re.sub(r"\d+", "", f[5])
            good = good and ops == {"exact": "M", "sub": "M", "D": "MDM", "I": "MIM"}[k]
        ok[key] = ok.get(key, 0) + bool(good)
    print(ok, tot, [l for l in out.stdout.split("\n") if "airs" in l])
    for key in tot:
        # the planted repeat family of this genome covers a good part of it: a far mate inside it loses all its seeds to the hit limit of the
        # single-end DP (10 hits per seed) and stays unaligned, as it would in the reference
        assert ok.get(key, 0) >= (0.6 if key == "exact/far" else 0.9) * tot[key], (ok, tot)
    assert "Number of pairs aligned by DP" in out.stdout and "aligned by deep DP" in out.stdout


def test_cli_pairs_new_default_dp_for_ends_with_too_many_hits(tmp_path):
    """An end that sits in a repeat with more copies than the default DP takes (getMaxHitNumForDefaultDP) while its mate failed the BWT stage:
    the pair goes through the new default DP -- the aligned end's hits as they are, the mate seeded, both aligned by the DP -- and comes out in
    .dpout.1 as a proper pair next to the right copy of the repeat."""
    import re
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "soap3-dp_amd", "py"))
    import s3dp
    from s3dp import builder
    rng = np.random.default_rng(4)
    n, L, copies = 200000, 100, 90
    codes = rng.integers(0, 4, size=n, dtype=np.uint8)
    rep = rng.integers(0, 4, size=L, dtype=np.uint8)
    starts = [1000 + 2100 * k for k in range(copies)]
    for st in starts:
        codes[st:st + L] = rep
    hi = builder.build_host_index(codes, device="cpu")
    hi.write_files(str(tmp_path / "g.index"), with_lkt=True)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    txt = lambda a: "".join("ACGT"[int(c)] for c in a)
    r1, r2, truth = [], [], []
    for k in range(0, copies, 4):                             # the repeat as the '+' end, its mate 300 further on with a 3-base deletion
        st = starts[k]
        m = st + 300
        c = int(rng.integers(30, 70))
        mate = np.concatenate([codes[m:m + c], codes[m + c + 3:m + L + 3]])
        r1.append(txt(rep)); r2.append(txt(comp[mate[::-1]]))
        truth.append((st + 1, m + 1))
    for fn, rr, tag in (("r1.fq", r1, 1), ("r2.fq", r2, 2)):
        with open(tmp_path / fn, "w") as f:
            for i, s in enumerate(rr):
                f.write("@n%d/%d\n%s\n+\n%s\n" % (i, tag, s, "I" * len(s)))
    out = subprocess.run([BIN, "pair", "g.index", "r1.fq", "r2.fq", "-u", "500", "-v", "100", "-L", "100", "-o", "out"], cwd=tmp_path, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    recs = {}
    for l in open(tmp_path / "out.dpout.1"):
        if not l.startswith("@"):
            f = l.rstrip("\n").split("\t")
            recs.setdefault(f[0], {})[int(f[1]) & 0xC0] = f
    m = re.search(r"aligned by new default DP: (\d+)", out.stdout)
    assert m and int(m.group(1)) >= 0.9 * len(truth), out.stdout[-1500:]
    good = 0
    for i, (p1, p2) in enumerate(truth):
        pr = recs.get("n%d" % i)
        if not pr or 0x40 not in pr or 0x80 not in pr:
            continue
        a, b = pr[0x40], pr[0x80]
        good += bool(int(a[1]) & 2 and int(a[3]) == p1 and a[5] == "100M" and abs(int(b[3]) - p2) <= 3 and re.sub(r"\d+", "", b[5]) == "MDM" and int(b[1]) & 16)
    assert good >= 0.9 * len(truth), (good, len(truth))


@pytest.mark.parametrize("name", ["g4", "g5"])
def test_cli_pair_dp_routing_matches_the_reference(tmp_path, name):
    """Pairs the BWT ladder leaves without a proper pairing, DP stages on: which end goes to the default DP (D), to the new default DP (N), is
    only kept for the unpaired output (O), which pairs go straight to the deep DP (U) -- and the hits in the order the DP stages will read them.
    Golden: the reference's own hostKernel (needProceedDP branch, CPUfunctions.cpp:2153-2262, 2444-2543) through oracle/ref_harness.cpp.  g5 is
    built for this: repeat families above and below the default DP's hit limit, mates damaged / random / far away."""
    env = dict(os.environ, S3DP_DUMP_ROUTING=str(tmp_path / "routing.txt"))
    out = subprocess.run([BIN, "pair", os.path.join(GOLDEN, name, "index"), os.path.join(GOLDEN, name, "reads_1.fq"), os.path.join(GOLDEN, name, "reads_2.fq"),
                          "-u", "500", "-v", "100", "-L", "100", "-o", "out"], cwd=tmp_path, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]

    def parse(text):
        lists, un = {}, set()
        for l in text.split("\n"):
            f = l.split()
            if not f:
                continue
            if f[0] == "U":
                un.add(int(f[1]))
            else:
                lists.setdefault((f[0], int(f[1])), []).append(tuple(int(x) for x in f[2:]))
        return lists, un
    mine, mine_un = parse(open(tmp_path / "routing.txt").read())
    ref, ref_un = parse(gzip.open(os.path.join(GOLDEN, name, "pe_routing.txt.gz"), "rt").read())
    assert mine_un == ref_un
    assert set(mine) == set(ref), (sorted(set(mine) ^ set(ref))[:10])
    bad = [k for k in ref if ref[k] != mine[k]]
    assert not bad, (bad[:5], ref[bad[0]][:6], mine[bad[0]][:6])
    if name == "g5":
        assert {k[0] for k in ref} == {"D", "N", "O"}


def test_cli_pairs_longer_than_150_bases_skip_the_bwt_stage(tmp_path):
    """2 x 200 bp pairs with the DP stages on: the reference skips its BWT stage above 150 bases (SOAP3-DP.cu:658-662) and hands every pair to the
    deep DP; so does the program.  Pairs with an indel or a few substitutions per end must come back as proper pairs at their true place."""
    import re
    import numpy as np
    name = "g1"
    rng = np.random.default_rng(33)
    chroms, cur = {}, None
    for l in open(os.path.join(GOLDEN, name, "genome.fa")):
        if l.startswith(">"):
            cur = l[1:].split()[0]
            chroms[cur] = []
        else:
            chroms[cur].append(l.strip())
    chroms = {k: "".join(v) for k, v in chroms.items()}
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = lambda s: "".join(comp[c] for c in reversed(s))
    L = 200

    def damage(seg, kind):
        if kind == 0:
            return seg[:L]
        if kind == 1:
            c = int(rng.integers(60, 140))
            return seg[:c] + seg[c + 4:L + 4]
        b = list(seg[:L])
        for q in rng.choice(np.arange(5, L - 5), size=4, replace=False):
            b[q] = "ACGT"[("ACGT".index(b[q]) + 1 + int(rng.integers(0, 3))) % 4]
        return "".join(b)
    names = sorted(chroms)
    r1s, r2s, truth = [], [], []
    while len(r1s) < 90:
        ch = names[int(rng.integers(0, len(names)))]
        ins = int(rng.integers(450, 560))
        st = int(rng.integers(0, len(chroms[ch]) - ins - 20))
        left, right = chroms[ch][st:st + L + 10], chroms[ch][st + ins - L:st + ins + 10]
        if "N" in left or "N" in right or len(right) < L + 10:
            continue
        k = len(r1s) % 3
        a, b = damage(left, k), rc(damage(right, (k + 1) % 3))
        swap = bool(rng.integers(0, 2))
        r1s.append(b if swap else a); r2s.append(a if swap else b)
        truth.append((ch, st + 1, st + ins - L + 1, swap))
    for fn, rr, tag in (("r1.fq", r1s, 1), ("r2.fq", r2s, 2)):
        with open(tmp_path / fn, "w") as f:
            for i, s in enumerate(rr):
                f.write("@w%d/%d\n%s\n+\n%s\n" % (i, tag, s, "I" * len(s)))
    out = subprocess.run([BIN, "pair", os.path.join(GOLDEN, name, "index"), "r1.fq", "r2.fq", "-u", "600", "-v", "200", "-L", "200", "-o", "out"], cwd=tmp_path,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    assert "All reads are directly processed by DP" in out.stdout
    assert sum(1 for l in open(tmp_path / "out.gout.1") if not l.startswith("@")) == 0
    recs = {}
    for src in ("dpout.1", "unpair"):
        for l in open(tmp_path / ("out." + src)):
            if not l.startswith("@"):
                f = l.rstrip("\n").split("\t")
                recs.setdefault(f[0], {})[int(f[1]) & 0xC0] = f
    assert sorted(recs) == sorted("w%d" % i for i in range(len(r1s)))
    good = 0
    for i, (ch, lp, rp, swap) in enumerate(truth):
        pr = recs["w%d" % i]
        fa, fb = (pr[0x80], pr[0x40]) if swap else (pr[0x40], pr[0x80])
        good += bool(int(fa[1]) & 2 and fa[2] == ch and abs(int(fa[3]) - lp) <= 4 and abs(int(fb[3]) - rp) <= 4 and int(fb[1]) & 16)
    assert good >= 0.85 * len(truth), (good, len(truth))


REF_BAM2SAM = os.path.join(ROOT, "oracle", "_ref", "ref_bam2sam")


def _decoded(path):
    out = subprocess.run([REF_BAM2SAM, path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1000:]
    return out.stdout


@pytest.mark.skipif(not os.path.exists(REF_BAM2SAM), reason="the reference's samtools is not built (oracle/ref.mk)")
@pytest.mark.parametrize("chunk", ["", "192"])
def test_cli_bam_single_reads(tmp_path, chunk):
    """`-b 3` (the reference's isOutputBinary: samopen "wb", SOAP3-DP.cu:413): the file the command line writes, read back by the REFERENCE's samtools,
    is line for line the SAM file of the same run with -b 2 (streamed single-end path: records compressed in pieces by all host threads)."""
    name = "g2"
    shutil.copyfile(os.path.join(GOLDEN, name, "reads.fq"), tmp_path / "r.fq")
    env = dict(os.environ)
    if chunk:
        env["S3DP_CHUNK_READS"] = chunk
    texts = {}
    for mode in ("sam", "bam"):
        out = subprocess.run([BIN, "single", os.path.join(GOLDEN, name, "index"), "r.fq", "-L", "120", "-s", "4", "-b", "3" if mode == "bam" else "2", "-o", mode],
                             cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        texts[mode] = open(tmp_path / (mode + ".gout.1")).read() if mode == "sam" else _decoded(str(tmp_path / "bam.gout.1"))
    assert texts["bam"].splitlines() == texts["sam"].splitlines()
    assert os.path.getsize(tmp_path / "bam.gout.1") < os.path.getsize(tmp_path / "sam.gout.1")
    # S3DP_BAM_LEVEL: zlib's level 9 and its default (-1, what samtools uses) hold the same records in fewer bytes than level 1
    sizes = {}
    for level in ("9", "-1", "0"):
        out = subprocess.run([BIN, "single", os.path.join(GOLDEN, name, "index"), "r.fq", "-L", "120", "-s", "4", "-b", "3", "-o", "bam" + level],
                             cwd=tmp_path, capture_output=True, text=True, timeout=600, env=dict(env, S3DP_BAM_LEVEL=level))
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        assert _decoded(str(tmp_path / ("bam%s.gout.1" % level))).splitlines() == texts["sam"].splitlines()
        sizes[level] = os.path.getsize(tmp_path / ("bam%s.gout.1" % level))
    assert sizes["9"] <= os.path.getsize(tmp_path / "bam.gout.1") < sizes["0"] and sizes["-1"] <= os.path.getsize(tmp_path / "bam.gout.1")


@pytest.mark.skipif(not os.path.exists(REF_BAM2SAM), reason="the reference's samtools is not built (oracle/ref.mk)")
def test_cli_bam_pairs_with_dp(tmp_path):
    """-b 3 on the paired-end default mode: all three files (.gout.1, .dpout.1, .unpair), DP stages on"""
    name = "g4"
    texts = {}
    for mode in ("sam", "bam"):
        out = subprocess.run([BIN, "pair", os.path.join(GOLDEN, name, "index"), os.path.join(GOLDEN, name, "reads_1.fq"), os.path.join(GOLDEN, name, "reads_2.fq"),
                              "-u", "500", "-v", "100", "-L", "100", "-o", mode] + (["-b", "3"] if mode == "bam" else []), cwd=tmp_path, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        for ext in ("gout.1", "dpout.1", "unpair"):
            p = str(tmp_path / (mode + "." + ext))
            texts[(mode, ext)] = open(p).read() if mode == "sam" else _decoded(p)
    for ext in ("gout.1", "dpout.1", "unpair"):
        assert texts[("bam", ext)].splitlines() == texts[("sam", ext)].splitlines(), ext


def _fastq_to_unaligned_bam(fq, bam, tmp_path, flags=(4,)):
    """the reads of a FASTQ file as unaligned BAM records (the product's BAM writer through tests/tools/sam2bam)"""
    host = os.path.join(ROOT, "soap3-dp_amd", "host")
    exe = os.path.join(ROOT, "tests", "tools", "sam2bam")
    src = [os.path.join(ROOT, "tests", "tools", "sam2bam.cpp"), os.path.join(host, "s3_bam.cpp")]
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + host, "-o", exe] + src + ["-lz"])
    sam = str(bam) + ".sam"
    with open(sam, "w") as g:
        g.write("@HD\tVN:1.3\tSO:unsorted\n@SQ\tSN:x\tLN:1000\n")
        files = [open(f).read().split("\n") for f in fq]
        n = (len(files[0]) - 1) // 4
        for i in range(n):
            for k, lines in enumerate(files):
                nm = lines[4 * i][1:].split()[0]
                if len(nm) > 2 and nm[-2] == "/":                      # BAM names carry no mate suffix (the text loader strips it: QueryParser's FASTQ path)
                    nm = nm[:-2]
                g.write("%s\t%d\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (nm, flags[k % len(flags)], lines[4 * i + 1], lines[4 * i + 3]))
    subprocess.check_call([exe, sam, str(bam)])


def test_cli_bam_input_single_and_pairs(tmp_path):
    """-bam (the reference's isReadBAM): reads from one BAM file -- same SAM records as from the FASTQ files the BAM was made of"""
    name = "g2"
    shutil.copyfile(os.path.join(GOLDEN, name, "reads.fq"), tmp_path / "r.fq")
    _fastq_to_unaligned_bam([tmp_path / "r.fq"], tmp_path / "r.bam", tmp_path)
    recs = {}
    for mode, args in (("fq", ["r.fq"]), ("bam", ["r.bam", "-bam"])):
        out = subprocess.run([BIN, "single", os.path.join(GOLDEN, name, "index")] + args + ["-L", "120", "-s", "4", "-b", "2", "-D", "rg", "-o", mode], cwd=tmp_path,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        recs[mode] = sam_records(open(tmp_path / (mode + ".gout.1")).read())[1]
    assert recs["bam"] == recs["fq"] and len(recs["fq"]) > 1000
    name = "g4"
    f1, f2 = os.path.join(GOLDEN, name, "reads_1.fq"), os.path.join(GOLDEN, name, "reads_2.fq")
    _fastq_to_unaligned_bam([f1, f2], tmp_path / "p.bam", tmp_path, flags=(77, 141))
    texts = {}
    for mode, args in (("pfq", [f1, f2]), ("pbam", ["p.bam", "-bam"])):
        out = subprocess.run([BIN, "pair", os.path.join(GOLDEN, name, "index")] + args + ["-u", "500", "-v", "100", "-L", "100", "-s", "3", "-D", "rg", "-o", mode], cwd=tmp_path,
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        texts[mode] = [l for l in open(tmp_path / (mode + ".gout.1")).read().split("\n") if l and not l.startswith("@PG")]
    assert texts["pbam"] == texts["pfq"] and len(texts["pfq"]) > 1000


def test_cli_bam_input_written_by_the_reference_library(tmp_path):
    """-bam on BAM files written by the REFERENCE's samtools library (tests/golden/bam/, made by tests/golden/make_bam_fixtures.py: the reads of g2 with every
    third record reverse-complemented under flag 16, the pairs of g4 with flags 77 / 141): the SAM records of the FASTQ run, i.e. the reference's golden
    records"""
    bam = os.path.join(GOLDEN, "bam")
    out = subprocess.run([BIN, "single", os.path.join(GOLDEN, "g2", "index"), os.path.join(bam, "se.bam"), "-bam", "-L", "120", "-s", "4", "-b", "2", "-o", "se"],
                         cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    mine = sam_records(open(tmp_path / "se.gout.1").read())[1]
    ref = sam_records(gzip.open(os.path.join(GOLDEN, "g2", "sam_s4.sam.gz"), "rt").read())[1]
    import re
    strip = lambda rec: {k: re.sub(r"\tRG:Z:[^\t]*", "", v) for k, v in rec.items()}
    assert strip(mine) == strip(ref) and len(ref) == 1500                    # (the read group is the query file's name: it differs, nothing else does)
    f1, f2 = os.path.join(GOLDEN, "g4", "reads_1.fq"), os.path.join(GOLDEN, "g4", "reads_2.fq")
    texts = {}
    for mode, args in (("pfq", [f1, f2]), ("pbam", [os.path.join(bam, "pe.bam"), "-bam"])):
        out = subprocess.run([BIN, "pair", os.path.join(GOLDEN, "g4", "index")] + args + ["-u", "500", "-v", "100", "-L", "100", "-s", "3", "-D", "rg", "-o", mode], cwd=tmp_path,
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        texts[mode] = [l for l in open(tmp_path / (mode + ".gout.1")).read().split("\n") if l and not l.startswith("@PG")]
    assert texts["pbam"] == texts["pfq"] and len(texts["pfq"]) > 1000


def test_cli_multi_modes(tmp_path):
    """single-multi / pair-multi (loadMultiInputFile, IniParam.cpp:28): one job per line of a tab-separated list, the index loaded once -- every job's
    files equal those of the same job run on its own"""
    g2, g4 = os.path.join(GOLDEN, "g2"), os.path.join(GOLDEN, "g4")
    shutil.copyfile(os.path.join(g2, "reads.fq"), tmp_path / "a.fq")
    with open(tmp_path / "list.txt", "w") as f:
        f.write("a.fq\tm1\trgA\tsampleA\n")
        f.write("a.fq\tm2\n")

    def run(args):
        out = subprocess.run([BIN] + args, cwd=tmp_path, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        return out.stdout
    log = run(["single-multi", os.path.join(g2, "index"), "list.txt", "-L", "120", "-s", "4", "-b", "2"])
    assert log.count("Finished copying index into device") == 1 and log.count("---- job") == 2
    run(["single", os.path.join(g2, "index"), "a.fq", "-L", "120", "-s", "4", "-b", "2", "-D", "rgA", "-A", "sampleA", "-o", "s1"])
    run(["single", os.path.join(g2, "index"), "a.fq", "-L", "120", "-s", "4", "-b", "2", "-o", "s2"])
    strip = lambda t: [l for l in t.split("\n") if not l.startswith("@PG")]
    assert strip(open(tmp_path / "m1.gout.1").read()) == strip(open(tmp_path / "s1.gout.1").read())
    assert strip(open(tmp_path / "m2.gout.1").read()) == strip(open(tmp_path / "s2.gout.1").read())
    assert "SM:sampleA" in open(tmp_path / "m1.gout.1").read()
    f1, f2 = os.path.join(g4, "reads_1.fq"), os.path.join(g4, "reads_2.fq")
    with open(tmp_path / "plist.txt", "w") as f:
        f.write("%s\t%s\t100\t500\tpm1\trgP\tsP\n" % (f1, f2))
    run(["pair-multi", os.path.join(g4, "index"), "plist.txt", "-L", "100", "-s", "3"])
    run(["pair", os.path.join(g4, "index"), f1, f2, "-L", "100", "-s", "3", "-u", "500", "-v", "100", "-D", "rgP", "-A", "sP", "-o", "ps1"])
    assert strip(open(tmp_path / "pm1.gout.1").read()) == strip(open(tmp_path / "ps1.gout.1").read())
