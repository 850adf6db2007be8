"""-m gpu: the drop-in command line against the CPU oracle on inputs a hundred times the size of the golden fixtures.

The fixtures of tests/golden hold up to 2,000 reads each (they are the reference's own output, record for record).  Here a 3 Mbp genome with planted repeat
families and N runs, 150,000 single reads of 40-100 bases and 40,000 pairs are simulated (tools/simulate.py), the index files are built on the GPU
(python -m s3dp.fasta: byte-identical to the reference tools' on every fixture, tests/test_fasta_front_end.py), and the SAM files the product writes with the
HIP path must equal, record for record, the files the SAME host code writes when the CPU oracle stands in the device's place (tests/tools/oracle_sam.cpp; the
oracle is pinned against the reference on the fixtures, tests/test_oracle_vs_reference.py, tests/test_host_sam_cpu.py).  The input is streamed in chunks, so
chunk boundaries, slot reuse and the heavy-read path of repeat families are all inside the comparison."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from test_gpu_cli_sam import BIN, sam_pair_records, sam_records
from test_host_sam_cpu import tool as oracle_tool

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workload(tmp_path_factory):
    w = tmp_path_factory.mktemp("scale")
    sim = os.path.join(ROOT, "tools", "simulate.py")
    subprocess.check_call([sys.executable, sim, "--genome-len", "3000000", "--chroms", "4", "--n-runs", "6", "--repeats", "24", "--genome-seed", "101", "--fasta", str(w / "g.fa"),
                           "--reads", "150000", "--read-len", "100", "--var-len", "40", "100", "--read-seed", "7", "--fastq", str(w / "se.fq")])
    subprocess.check_call([sys.executable, sim, "--genome-len", "3000000", "--chroms", "4", "--n-runs", "6", "--repeats", "24", "--genome-seed", "101", "--fasta", str(w / "g2.fa"),
                           "--pairs", "40000", "--read-len", "100", "--read-seed", "9", "--fastq", str(w / "pe_1.fq"), "--fastq2", str(w / "pe_2.fq")])
    assert open(w / "g.fa").read() == open(w / "g2.fa").read()
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "soap3-dp_amd", "py"))
    subprocess.check_call([sys.executable, "-m", "s3dp.fasta", str(w / "g.fa")], env=env, cwd=w)
    return w


# report types as in the fixtures' variants: all best (default), all valid, unique best, random best with MD / NM tags
@pytest.mark.parametrize("tag,opts", [("s4", ["-s", "4"]), ("h1_s2", ["-s", "2", "-h", "1"]), ("h3_s3", ["-s", "3", "-h", "3"]), ("h4_s3p", ["-s", "3", "-h", "4", "-p"])])
def test_single_reads_equal_the_oracle_at_scale(workload, tag, opts):
    w = workload
    env = dict(os.environ, S3DP_CHUNK_READS="40000")
    out = subprocess.run([BIN, "single", "g.fa.index", "se.fq", "-L", "100"] + opts + ["-b", "2", "-o", "gpu_se_" + tag], cwd=w, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    ora = subprocess.run([oracle_tool(), "single", "g.fa.index", "se.fq", "-L", "100"] + opts + ["-o", "cpu_se_" + tag], cwd=w, capture_output=True, text=True, timeout=1800)
    assert ora.returncode == 0, ora.stderr[-2000:]
    mine_h, mine = sam_records(open(w / ("gpu_se_%s.gout.1" % tag)).read())
    ref_h, ref = sam_records(open(w / ("cpu_se_%s.gout.1" % tag)).read())
    assert mine_h == ref_h and len(ref) == 150000 and set(mine) == set(ref)
    bad = [k for k in ref if ref[k] != mine[k]]
    assert not bad, "%d of %d records differ, first:\n%s\n%s" % (len(bad), len(ref), ref[bad[0]], mine[bad[0]])
    if tag == "s4":
        assert sum(1 for v in ref.values() if "\tXA:Z:" in v) > 3000      # the repeat families are in it


@pytest.mark.parametrize("tag,opts", [("s3", ["-s", "3"]), ("h1_s2", ["-s", "2", "-h", "1"]), ("h3_s3", ["-s", "3", "-h", "3"])])
def test_pairs_equal_the_oracle_at_scale(workload, tag, opts):
    w = workload
    env = dict(os.environ, S3DP_CHUNK_READS="30000")
    out = subprocess.run([BIN, "pair", "g.fa.index", "pe_1.fq", "pe_2.fq", "-L", "100", "-u", "500", "-v", "200"] + opts + ["-b", "2", "-o", "gpu_pe_" + tag], cwd=w,
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    ora = subprocess.run([oracle_tool(), "pair", "g.fa.index", "pe_1.fq", "pe_2.fq", "-L", "100", "-u", "500", "-v", "200"] + opts + ["-o", "cpu_pe_" + tag], cwd=w,
                         capture_output=True, text=True, timeout=1800)
    assert ora.returncode == 0, ora.stderr[-2000:]
    mine = sam_pair_records(open(w / ("gpu_pe_%s.gout.1" % tag)).read())
    ref = sam_pair_records(open(w / ("cpu_pe_%s.gout.1" % tag)).read())
    assert mine[0] == ref[0] and set(mine[1]) == set(ref[1]) and len(ref[1]) >= 40000
    bad = [k for k in ref[1] if ref[1][k] != mine[1][k]]
    assert not bad, "%d of %d records differ, first:\n%s\n%s" % (len(bad), len(ref[1]), ref[1][bad[0]], mine[1][bad[0]])


REF_HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


@pytest.mark.skipif(not os.path.exists(REF_HARNESS), reason="oracle/_ref/ref_harness is not built (make -f oracle/ref.mk where /root/reference exists)")
@pytest.mark.parametrize("mode,extra", [("pair", []), ("pair", ["-h", "1"]), ("pair", ["-h", "3"]), ("pair", ["-h", "4", "-p"]), ("single", []), ("single", ["-h", "1", "-p"]),
                                        ("pair", ["families"]), ("pair", ["crowded"])],
                         ids=["pair", "pair_h1", "pair_h3", "pair_h4p", "single", "single_h1p", "pair_every_route", "pair_crowded"])
def test_dp_stages_equal_the_reference_engines_at_scale(tmp_path, mode, extra):
    """The DP stages ON, against the REFERENCE's own pipeline run at test time: oracle/_ref/ref_harness is the reference's host code (BWT ladder by its CPU twin,
    routing, the DP engines' packers / seeding / merges / decode / OutputBuffer::ready / record writers compiled from /root/reference by oracle/ref.mk; the two DP
    kernels behind them are the restatement that tests/test_gpu_dp_reference.py holds bit-identical to the reference's device code).  30,000 pairs 2 x 150 bp
    (config C3's options; a tenth of the mates with an indel, a twentieth of the pairs damaged in both mates) -- twenty times fixture g7 -- and 40,000 single
    reads: the three SAM files (two for single reads) must hold the same records."""
    from test_gpu_dp_pipeline import by_read, records
    w = tmp_path
    sim = os.path.join(ROOT, "tools", "simulate.py")
    ini = os.path.join(ROOT, "soap3-dp_amd", "host", "soap3-dp.ini")
    families = extra == ["families"]
    crowded = extra == ["crowded"]
    if crowded:
        # a family of 9,000 copies one per cent apart: an end inside it reaches the BWT stage's hit limit (8,000 per end in the paired ladder), its pair the
        # too-many-hits routes -- what round 6's parity leg at the benchmark's size (bench.py dp_oracle_check: pairs of the 10 M-pair run through the reference's
        # own host pipeline on the 3.1 Gbp index) found three pairs in two thousand to differ in
        extra = []
        subprocess.check_call([sys.executable, sim, "--genome-len", "6000000", "--chroms", "3", "--n-runs", "4", "--repeats", "16", "--genome-seed", "77", "--fasta", str(w / "g.fa"),
                               "--family-copies", "9000", "--family-len", "320", "--family-div", "0.01",
                               "--pairs", "12000", "--read-len", "150", "--read-seed", "5", "--fastq", str(w / "r_1.fq"), "--fastq2", str(w / "r_2.fq"),
                               "--indel-frac", "0.10", "--damaged-frac", "0.05"])
        files, opts = ["r_1.fq", "r_2.fq"], ["-L", "150", "-u", "500", "-v", "1", "-b", "2"]
    elif families:
        # EVERY route of the DP stages populated (round 5): a 3 Mbp genome with a young family of 600 copies (0.4 % apart) -- an end inside it comes with
        # hundreds of BWT hits, above the default DP's hit limit: 700 pairs go through the NEW DEFAULT DP (given hits of one end, long merge runs), 9,000
        # through the default DP, 7,700 through the deep DP, 7,400 ends through the single-end DP; every record of every route against the reference engines
        extra = []
        subprocess.check_call([sys.executable, sim, "--genome-len", "3000000", "--chroms", "3", "--n-runs", "4", "--repeats", "16", "--genome-seed", "55", "--fasta", str(w / "g.fa"),
                               "--family-copies", "600", "--family-len", "700", "--family-div", "0.004",
                               "--pairs", "30000", "--read-len", "150", "--read-seed", "5", "--fastq", str(w / "r_1.fq"), "--fastq2", str(w / "r_2.fq"),
                               "--indel-frac", "0.10", "--damaged-frac", "0.05"])
        files, opts = ["r_1.fq", "r_2.fq"], ["-L", "150", "-u", "500", "-v", "1", "-b", "2"]
    elif mode == "pair":
        subprocess.check_call([sys.executable, sim, "--genome-len", "2000000", "--chroms", "3", "--n-runs", "4", "--repeats", "16", "--genome-seed", "55", "--fasta", str(w / "g.fa"),
                               "--pairs", "30000", "--read-len", "150", "--read-seed", "5", "--fastq", str(w / "r_1.fq"), "--fastq2", str(w / "r_2.fq"),
                               "--indel-frac", "0.10", "--damaged-frac", "0.05"])
        files, opts = ["r_1.fq", "r_2.fq"], ["-L", "150", "-u", "500", "-v", "1", "-b", "2"] + extra
    else:
        subprocess.check_call([sys.executable, sim, "--genome-len", "2000000", "--chroms", "3", "--n-runs", "4", "--repeats", "16", "--genome-seed", "55", "--fasta", str(w / "g.fa"),
                               "--reads", "40000", "--read-len", "100", "--read-seed", "6", "--fastq", str(w / "r_1.fq"), "--mix", "30", "20", "15", "15", "10", "10"])
        files, opts = ["r_1.fq"], ["-L", "100", "-b", "2"] + extra
    # the index files come from the reference's own tools here (the harness wants the CPU 2BWT's .fmv files too, which the product's builder has no use for)
    refdir = os.path.dirname(REF_HARNESS)
    import shutil
    shutil.copy(os.path.join(refdir, "soap3-dp-builder.ini"), w / "soap3-dp-builder.ini")
    subprocess.check_call([os.path.join(refdir, "soap3-dp-builder"), "g.fa"], cwd=w, stdout=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(refdir, "BGS-Build"), "g.fa.index"], cwd=w, stdout=subprocess.DEVNULL)
    ref = subprocess.run([REF_HARNESS, "sam", ini, mode, "g.fa.index"] + files + opts + ["-o", "ref"], cwd=w, capture_output=True, text=True, timeout=1800)
    assert ref.returncode == 0, ref.stdout[-2000:] + ref.stderr[-2000:]
    out = subprocess.run([BIN, mode, "g.fa.index"] + files + opts + ["-o", "mine"], cwd=w, capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, S3DP_CHUNK_READS="16384"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    problems, seen = [], 0
    for kind in ("gout.1", "dpout.1", "unpair"):
        rp, mp = w / ("ref." + kind), w / ("mine." + kind)
        if not os.path.exists(rp):
            continue
        ref_h, refc = records(open(rp).read())
        mine_h, minec = records(open(mp).read()) if os.path.exists(mp) else ([], {})
        seen += sum(refc.values())
        if mine_h != ref_h:
            problems.append("%s: header differs" % kind)
        if minec != refc:
            rb, mb = by_read(refc), by_read(minec)
            diff = [k for k in set(rb) | set(mb) if sorted(rb.get(k, [])) != sorted(mb.get(k, []))]
            problems.append("%s: %d of %d reads differ, e.g.\nREF  %s\nMINE %s" % (kind, len(diff), len(rb), rb.get(diff[0]), mb.get(diff[0])))
    assert not problems, "; ".join(problems)
    assert seen >= (24000 if crowded else 60000 if mode == "pair" else 40000)                    # every read has its record in one of the files
    if families:
        import re
        got = {k: int(v) for k, v in re.findall(r"\[Main\] (Number of pairs aligned by new default DP|Number of pairs aligned by DP|Number of pairs aligned by deep DP|"
                                                r"Pairs left without a proper pairing): (\d+)", out.stdout)}
        want = {k: int(v) for k, v in re.findall(r"\[harness\] (new default DP|default DP|deep DP): \d+ pairs in, (\d+) aligned", ref.stdout + ref.stderr)}
        assert got["Number of pairs aligned by new default DP"] == want["new default DP"] >= 300, (got, want)
        assert got["Number of pairs aligned by DP"] == want["default DP"] >= 4000 and got["Number of pairs aligned by deep DP"] == want["deep DP"] >= 1000, (got, want)
        assert got["Pairs left without a proper pairing"] >= 1000, got


@pytest.mark.parametrize("tag,opts", [("s3", ["-s", "3"]), ("h1_s2p", ["-s", "2", "-h", "1", "-p"])])
def test_long_reads_equal_the_oracle_at_scale(workload, tmp_path, tag, opts):
    """long-read mode (reads above 120 bases: the first 100 are searched, the rest verified on the device): 30,000 reads of 130-220 bases.  (Reads above 120 bases only:
    the reference mixes up seed lengths when a run holds both kinds)"""
    w = workload
    sim = os.path.join(ROOT, "tools", "simulate.py")
    subprocess.check_call([sys.executable, sim, "--genome-len", "3000000", "--chroms", "4", "--n-runs", "6", "--repeats", "24", "--genome-seed", "101", "--fasta", str(tmp_path / "g.fa"),
                           "--reads", "30000", "--read-len", "220", "--var-len", "130", "220", "--read-seed", "11", "--fastq", str(w / "long.fq")])
    env = dict(os.environ, S3DP_CHUNK_READS="8192")
    out = subprocess.run([BIN, "single", "g.fa.index", "long.fq", "-L", "220"] + opts + ["-b", "2", "-o", "gpu_long_" + tag], cwd=w, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    ora = subprocess.run([oracle_tool(), "single", "g.fa.index", "long.fq", "-L", "220"] + opts + ["-o", "cpu_long_" + tag], cwd=w, capture_output=True, text=True, timeout=1800)
    assert ora.returncode == 0, ora.stderr[-2000:]
    mine_h, mine = sam_records(open(w / ("gpu_long_%s.gout.1" % tag)).read())
    ref_h, ref = sam_records(open(w / ("cpu_long_%s.gout.1" % tag)).read())
    assert mine_h == ref_h and len(ref) == 30000 and set(mine) == set(ref)
    bad = [k for k in ref if ref[k] != mine[k]]
    assert not bad, "%d of %d records differ, first:\n%s\n%s" % (len(bad), len(ref), ref[bad[0]], mine[bad[0]])


def test_two_members_on_one_device_at_scale(workload):
    """`-c 0,0`: two members of a group on the one GPU, chunks dealt to whichever is free, a set of output files each -- the union of the two files is the
    oracle's file"""
    w = workload
    env = dict(os.environ, S3DP_CHUNK_READS="10000")
    out = subprocess.run([BIN, "single", "g.fa.index", "se.fq", "-L", "100", "-s", "4", "-b", "2", "-c", "0,0", "-o", "gpu_two"], cwd=w, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    if not os.path.exists(w / "cpu_se_s4.gout.1"):
        ora = subprocess.run([oracle_tool(), "single", "g.fa.index", "se.fq", "-L", "100", "-s", "4", "-o", "cpu_se_s4"], cwd=w, capture_output=True, text=True, timeout=1800)
        assert ora.returncode == 0, ora.stderr[-2000:]
    ref = sam_records(open(w / "cpu_se_s4.gout.1").read())[1]
    mine = {}
    for k in (1, 2):
        part = sam_records(open(w / ("gpu_two.gout.%d" % k)).read())[1]
        assert not (set(part) & set(mine))
        mine.update(part)
    assert len(mine) == len(ref) == 150000
    bad = [k for k in ref if ref[k] != mine.get(k)]
    assert not bad, "%d records differ, first:\n%s\n%s" % (len(bad), ref[bad[0]], mine.get(bad[0]))


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "soap3-dp-builder")), reason="the reference builder is not built (oracle/ref.mk)")
def test_gpu_built_index_equals_the_reference_tools_at_scale(workload, tmp_path):
    """the index files `python -m s3dp.fasta` builds on the GPU for the 3 Mbp genome of this module (four sequences, N runs, repeat families: the suffix sort
    needs several rounds of prefix doubling) against the files the REFERENCE's soap3-dp-builder + BGS-Build write for the same FASTA file, run here: byte for
    byte, the lookup tables included (the fixtures of tests/test_fasta_front_end.py are a tenth of this size)."""
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "soap3-dp_amd", "py"))
    from s3dp import fasta
    refdir = os.path.join(ROOT, "oracle", "_ref")
    shutil.copy(os.path.join(refdir, "soap3-dp-builder.ini"), tmp_path / "soap3-dp-builder.ini")
    shutil.copy(workload / "g.fa", tmp_path / "g.fa")
    subprocess.check_call([os.path.join(refdir, "soap3-dp-builder"), "g.fa"], cwd=tmp_path, stdout=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(refdir, "BGS-Build"), "g.fa.index"], cwd=tmp_path, stdout=subprocess.DEVNULL)
    seed = int(open(tmp_path / "g.fa.index.ann").readline().split()[2])          # (the reference stamps a seed into the first line; it decides nothing else)
    out = str(tmp_path / "mine")
    fasta.build_index_files(str(workload / "g.fa"), out, device="cuda", seed=seed, with_lkt=True)
    for ext in (".ann", ".amb", ".tra", ".pac", ".bwt", ".rev.bwt", ".fmv.gpu", ".rev.fmv.gpu", ".sa", ".lkt", ".rev.lkt"):
        a, b = open(out + ext, "rb").read(), open(str(tmp_path / "g.fa.index") + ext, "rb").read()
        assert a == b, "%s differs from the reference tools' file (%d / %d bytes)" % (ext, len(a), len(b))


@pytest.mark.skipif(not os.path.exists(REF_HARNESS), reason="oracle/_ref/ref_harness is not built (make -f oracle/ref.mk where /root/reference exists)")
@pytest.mark.parametrize("mode", ["pair", "single"])
def test_reads_with_dp_hits_on_both_strands(tmp_path, mode):
    """Round 6 (found by bench.py's DP parity leg on the benchmark genome, whose repeat families have inverted copies): a read the BWT stage cannot place
    whose single-end DP finds EQUALLY good hits on both strands.  The reference lists a read's DP hits in position order whatever their strands (the strand
    pass of its candidate sort is lost: s3dp_seed.hip, singleMergeKernel) -- the primary record is the hit at the smaller position; the product used to
    list the '+' hits first.  Eight such reads (an inverted copy of their 150 bases planted 300 kbp away, eight substitutions each) among 2,000 ordinary
    pairs, against the reference's own pipeline."""
    from test_gpu_dp_pipeline import by_read, records
    import shutil
    import numpy as np
    w = tmp_path
    sim = os.path.join(ROOT, "tools", "simulate.py")
    ini = os.path.join(ROOT, "soap3-dp_amd", "host", "soap3-dp.ini")
    subprocess.check_call([sys.executable, sim, "--genome-len", "2000000", "--chroms", "3", "--n-runs", "4", "--repeats", "16", "--genome-seed", "55", "--fasta", str(w / "g0.fa"),
                           "--pairs", "2000", "--read-len", "150", "--read-seed", "5", "--fastq", str(w / "a_1.fq"), "--fastq2", str(w / "a_2.fq"),
                           "--indel-frac", "0.10", "--damaged-frac", "0.05"])
    txt = open(w / "g0.fa").read().split("\n")
    hdr = [i for i, l in enumerate(txt) if l.startswith(">")]
    width = len(txt[1])
    seq = list("".join(txt[hdr[0] + 1:hdr[1]]))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    rng = np.random.default_rng(9)
    L, special = 150, []
    for k in range(8):
        lo, hi = 20000 + k * 9000, 300000 + k * 11000
        seg = seq[hi:hi + L]
        assert "N" not in seg and "N" not in seq[lo:lo + L]
        seq[lo:lo + L] = [comp[c] for c in reversed(seg)]
        r = list(seg)
        for p in rng.choice(np.arange(25, 125), 8, replace=False):
            r[p] = rng.choice([c for c in "ACGT" if c != r[p]])
        if k % 2:
            r = [comp[c] for c in reversed(r)]
        special.append(("".join(r), "".join(rng.choice(list("ACGT"), L))))
    s = "".join(seq)
    open(w / "g.fa", "w").write("\n".join(txt[:hdr[0] + 1] + [s[i:i + width] for i in range(0, len(s), width)] + txt[hdr[1]:]))
    l1, l2 = open(w / "a_1.fq").read().strip("\n").split("\n"), open(w / "a_2.fq").read().strip("\n").split("\n")
    at = {0: [0], 1: [1], 500: [2], 501: [3], 1200: [4], 1999: [5], 2000: [6, 7]}
    o1, o2 = [], []
    for j in range(2001):
        for k in at.get(j, []):
            o1 += ["@s%d/1" % k, special[k][0], "+", "I" * L]
            o2 += ["@s%d/2" % k, special[k][1], "+", "I" * L]
        if j < 2000:
            o1 += l1[4 * j:4 * j + 4]
            o2 += l2[4 * j:4 * j + 4]
    open(w / "r_1.fq", "w").write("\n".join(o1) + "\n")
    open(w / "r_2.fq", "w").write("\n".join(o2) + "\n")
    refdir = os.path.dirname(REF_HARNESS)
    shutil.copy(os.path.join(refdir, "soap3-dp-builder.ini"), w / "soap3-dp-builder.ini")
    subprocess.check_call([os.path.join(refdir, "soap3-dp-builder"), "g.fa"], cwd=w, stdout=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(refdir, "BGS-Build"), "g.fa.index"], cwd=w, stdout=subprocess.DEVNULL)
    files = ["r_1.fq", "r_2.fq"] if mode == "pair" else ["r_1.fq"]
    opts = ["-L", "150", "-u", "500", "-v", "1", "-b", "2"] if mode == "pair" else ["-L", "150", "-b", "2"]
    ref = subprocess.run([REF_HARNESS, "sam", ini, mode, "g.fa.index"] + files + opts + ["-o", "ref"], cwd=w, capture_output=True, text=True, timeout=900)
    assert ref.returncode == 0, ref.stdout[-2000:] + ref.stderr[-2000:]
    out = subprocess.run([BIN, mode, "g.fa.index"] + files + opts + ["-o", "mine"], cwd=w, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    kind = "unpair" if mode == "pair" else "dpout.1"
    ref_s = [l for l in open(w / ("ref." + kind)).read().split("\n") if l.startswith("s")]
    # the case is what it claims to be: reads with two equally good DP hits, the smaller position first whatever its strand
    tied = [l for l in ref_s if "\tX0:i:2\t" in l and "XA:Z:" in l]
    assert len(tied) >= 6 and any(int(l.split("\t")[1]) & 16 for l in tied) and any(not int(l.split("\t")[1]) & 16 for l in tied)
    assert all(int(l.split("\t")[3]) < int(l.split("XA:Z:")[1].split(",")[1][1:]) for l in tied)
    problems = []
    for k in ("gout.1", "dpout.1", "unpair"):
        rp, mp = w / ("ref." + k), w / ("mine." + k)
        if not os.path.exists(rp):
            continue
        refc = records(open(rp).read())[1]
        minec = records(open(mp).read())[1] if os.path.exists(mp) else {}
        if minec != refc:
            rb, mb = by_read(refc), by_read(minec)
            diff = [q for q in set(rb) | set(mb) if sorted(rb.get(q, [])) != sorted(mb.get(q, []))]
            problems.append("%s: %d of %d reads differ, e.g.\nREF  %s\nMINE %s" % (k, len(diff), len(rb), rb.get(diff[0]), mb.get(diff[0])))
    assert not problems, "; ".join(problems)
