"""-m gpu: the whole DP pipeline of the drop-in `soap3-dp` against the reference's OWN DP engines.

The golden files tests/golden/*/dp_<tag>.* were written by oracle/_ref/ref_harness running the reference's host code for the DP stages --
HalfEndAlgnBatch::pack / PairEndAlgnBatch::packLeft,packRight / SingleEndAlgnBatch::pack (window geometry, clips, anchors), the seeding
engines with singleMerge / pairEndMerge, the decode threads (CIGAR, edit distance, insert size), OutputBuffer::ready and the record
writers -- compiled from /root/reference by oracle/ref.mk, behind the reference's BWT ladder (tests/golden/make_fixtures.sh, section
"DP stages ON").  Checked here:
  * the three SAM files (.gout.1, .dpout.1, .unpair), record for record;
  * the multiset of DP problems the engines submitted (S3DP_DUMP_DP): read / window sequences, lengths, clip allowances, anchors, cutoffs
    as the packers built them, and score / start / tie count / op string as the decode stage received them.  (Compared as sets, and the
    product may submit fewer: the reference's packRight (DV-DPfunctions.cu:3420-3472) leaves the slot of a candidate whose left end failed as
    it is, so that unchanged left problem runs a second time with the right ends; the product does not submit it again.)
Record ORDER inside a file is not compared, and cannot be: the reference writes .gout per host thread and phase, .dpout per DP stage (default DP, new
default DP, deep DP rounds) through its output threads, the product per chunk of the input and stage -- the same records come in a different interleaving
by construction (with S3DP_CHUNK_READS every chunk writes its own stages behind each other).  What order could break is pinned elsewhere: the records of one
READ (primary first, XA lists in hit order) are inside the compared lines, and the order of a read's hits is compared list for list in test_gpu_parity.py.
g7 is BASELINE config C3's shape (2 x 150 bp, -u 500 -v 1); dp_long (2 x 200 bp) skips the BWT stage; dp_se* are single reads."""
import gzip
import os
import re
import shutil
import subprocess
from collections import Counter

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "soap3-dp_amd", "soap3-dp")
REPORT = os.path.join(ROOT, "gpurun_out", "dp_pipeline")

RG = re.compile(r"\tRG:Z:[^\t]*")


def records(text):
    """header lines (without @PG / @RG, whose id is the file name as typed) and the multiset of records with the read-group tag removed"""
    hdr, rec = [], []
    for l in text.split("\n"):
        if not l:
            continue
        if l.startswith("@"):
            if not l.startswith("@PG") and not l.startswith("@RG"):
                hdr.append(l)
        else:
            rec.append(RG.sub("", l))
    return hdr, Counter(rec)


def report(case, name, lines):
    try:
        os.makedirs(REPORT, exist_ok=True)
        with open(os.path.join(REPORT, "%s.%s.txt" % (case, name)), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass


def by_read(counter):
    d = {}
    for l, c in counter.items():
        f = l.split("\t", 2)
        d.setdefault((f[0], int(f[1]) & 0xC0), []).extend([l] * c)
    return d


CASES = [
    # case, golden dir, tag, mode, read files (golden names), options
    ("g7_h2", "g7", "h2", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "150", "-u", "500", "-v", "1", "-b", "2"]),
    ("g7_h2p", "g7", "h2p", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "150", "-u", "500", "-v", "1", "-b", "2", "-p"]),
    # the small-device decision of SemiGlobalAligner::decideConfiguration (scheme 2 on the 650-column windows of the default DP; S3DP_DP_SCHEME=2)
    ("g7_h2_scheme2", "g7", "h2s2", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "150", "-u", "500", "-v", "1", "-b", "2"]),
    # the same run with one DP problem per wavefront (S3DP_DP_PAIRED=0): the paired fill kernel and the one-problem kernel write the same tables
    ("g7_h2_unpaired_dp", "g7", "h2", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "150", "-u", "500", "-v", "1", "-b", "2"]),
    ("g7_h1", "g7", "h1", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "150", "-u", "500", "-v", "1", "-b", "2", "-h", "1"]),
    ("g7_h3", "g7", "h3", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "150", "-u", "500", "-v", "1", "-b", "2", "-h", "3"]),
    ("g7_h4", "g7", "h4", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "150", "-u", "500", "-v", "1", "-b", "2", "-h", "4"]),
    ("g7_long", "g7", "long", "pair", ("long_1.fq", "long_2.fq"), ["-L", "200", "-u", "600", "-v", "100", "-b", "2"]),
    ("g7_se", "g7", "se", "single", ("reads_1.fq",), ["-L", "150", "-b", "2"]),
    ("g7_se_h1", "g7", "se_h1", "single", ("reads_2.fq",), ["-L", "150", "-b", "2", "-h", "1", "-p"]),
    ("g4_h2", "g4", "h2", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "100", "-u", "500", "-v", "100", "-b", "2"]),
    ("g4_h1", "g4", "h1", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "100", "-u", "500", "-v", "100", "-b", "2", "-h", "1"]),
    ("g5_h2", "g5", "h2", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "100", "-u", "500", "-v", "100", "-b", "2"]),
    ("g1_se", "g1", "se", "single", ("reads.fq",), ["-L", "100", "-b", "2"]),
    # the same input in chunks of 1024 / 512 reads (the command line streams its input: S3DP_CHUNK_READS): nothing may depend on the chunking
    ("g7_h2_chunked", "g7", "h2", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "150", "-u", "500", "-v", "1", "-b", "2"]),
    ("g7_se_chunked", "g7", "se", "single", ("reads_1.fq",), ["-L", "150", "-b", "2"]),
    # ... and with two DP stage threads taking the chunks in turns (S3DP_DP_THREADS=2: two sets of aligners, one writer per output file at a time)
    ("g7_h2_chunked_two_dp_threads", "g7", "h2", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "150", "-u", "500", "-v", "1", "-b", "2"]),
    ("g7_se_chunked_two_dp_threads", "g7", "se", "single", ("reads_1.fq",), ["-L", "150", "-b", "2"]),
    # pairEndMerge: every left run walked by the whole wave of its head lane (S3DP_PAIRMERGE_WAVE=0) / every run by its head lane alone (round 5: runs
    # above 24 positions go to the wave by default); g5 holds the pairs whose ends sit in repeat families above the hit limit (new default DP)
    ("g7_h2_merge_by_wave", "g7", "h2", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "150", "-u", "500", "-v", "1", "-b", "2"]),
    ("g5_h2_merge_by_wave", "g5", "h2", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "100", "-u", "500", "-v", "100", "-b", "2"]),
    ("g4_h2_merge_by_wave", "g4", "h2", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "100", "-u", "500", "-v", "100", "-b", "2"]),
    ("g5_h2_merge_by_lane", "g5", "h2", "pair", ("reads_1.fq", "reads_2.fq"), ["-L", "100", "-u", "500", "-v", "100", "-b", "2"]),
]


@pytest.mark.parametrize("case,gdir,tag,mode,reads,opts", CASES, ids=[c[0] for c in CASES])
def test_dp_pipeline_matches_the_reference_engines(tmp_path, case, gdir, tag, mode, reads, opts):
    assert os.path.exists(BIN), "soap3-dp binary not built"
    names = []
    for k, r in enumerate(reads):
        n = "%s_%d.fq" % (gdir, k + 1)
        shutil.copy(os.path.join(GOLDEN, gdir, r), tmp_path / n)
        names.append(n)
    env = dict(os.environ, S3DP_DUMP_DP=str(tmp_path / "problems.txt"))
    if case.endswith("_unpaired_dp"):
        env["S3DP_DP_PAIRED"] = "0"
    if case.endswith("_scheme2"):
        env["S3DP_DP_SCHEME"] = "2"
    if case.endswith("_two_dp_threads"):
        env["S3DP_DP_THREADS"] = "2"
        env["S3DP_CHUNK_READS"] = "512"
    if case.endswith("_merge_by_wave"):
        env["S3DP_PAIRMERGE_WAVE"] = "0"
    if case.endswith("_merge_by_lane"):
        env["S3DP_PAIRMERGE_WAVE"] = str(0xFFFFFFFF)
    if case.endswith("_chunked"):
        env["S3DP_CHUNK_READS"] = "1024" if mode == "pair" else "512"
    out = subprocess.run([BIN, mode, os.path.join(GOLDEN, gdir, "index")] + names + opts + ["-o", "out"], cwd=tmp_path, capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    problems = []
    for kind in ("gout", "dpout", "unpair"):
        g = os.path.join(GOLDEN, gdir, "dp_%s.%s.sam.gz" % (tag, kind))
        mine_path = tmp_path / ("out.%s" % {"gout": "gout.1", "dpout": "dpout.1", "unpair": "unpair"}[kind])
        if not os.path.exists(g):
            continue
        ref_h, ref = records(gzip.open(g, "rt").read())
        mine_h, mine = records(open(mine_path).read()) if os.path.exists(mine_path) else ([], Counter())
        if sum(ref.values()) == 0 and not os.path.exists(mine_path):
            continue
        if mine_h != ref_h:
            problems.append("%s: header differs" % kind)
        if mine != ref:
            rb, mb = by_read(ref), by_read(mine)
            lines = []
            for k in sorted(set(rb) | set(mb)):
                if sorted(rb.get(k, [])) != sorted(mb.get(k, [])):
                    for l in rb.get(k, []):
                        lines.append("REF  " + l)
                    for l in mb.get(k, []):
                        lines.append("MINE " + l)
                    lines.append("")
            nd = sum(1 for k in set(rb) | set(mb) if sorted(rb.get(k, [])) != sorted(mb.get(k, [])))
            report(case, kind, lines)
            problems.append("%s: %d of %d reads differ (ref %d records, mine %d)" % (kind, nd, len(rb), sum(ref.values()), sum(mine.values())))
    g = os.path.join(GOLDEN, gdir, "dp_%s.problems.txt.gz" % tag)
    if os.path.exists(g):
        ref_p = Counter(l for l in gzip.open(g, "rt").read().split("\n") if l.startswith("P"))
        mine_p = Counter(l for l in open(tmp_path / "problems.txt").read().split("\n") if l.startswith("P")) if os.path.exists(tmp_path / "problems.txt") else Counter()
        if set(ref_p) != set(mine_p) or sum(mine_p.values()) > sum(ref_p.values()):
            only_ref = sorted(set(ref_p) - set(mine_p))
            only_mine = sorted(set(mine_p) - set(ref_p))
            report(case, "problems", ["only in the reference: %d" % len(only_ref)] + only_ref[:400] + ["", "only in the product: %d" % len(only_mine)] + only_mine[:400])
            problems.append("DP problems: %d only in the reference, %d only in the product (of %d / %d)" % (len(only_ref), len(only_mine), sum(ref_p.values()),
                                                                                                         sum(mine_p.values())))
    assert not problems, "; ".join(problems) + "  (details: gpurun_out/dp_pipeline/%s.*.txt)" % case
