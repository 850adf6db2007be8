"""The product's streaming read loader (soap3-dp_amd/host: ReadChunker) must give the same reads whatever the chunk size: plain and gzip, FASTQ and
FASTA (multi-line), single files and pairs of files with different record sizes, a quality line that begins with '@', files of unequal length."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

TOOL = "/tmp/s3dp_chunk_reads"


@pytest.fixture(scope="module")
def tool():
    src = [os.path.join(ROOT, "tests", "tools", "chunk_reads.cpp"), os.path.join(ROOT, "soap3-dp_amd", "host", "s3_host.cpp"), os.path.join(ROOT, "soap3-dp_amd", "host", "s3_bam.cpp"),
           os.path.join(ROOT, "soap3-dp_amd", "host", "s3_host_dp.cpp"), os.path.join(ROOT, "soap3-dp_amd", "host", "s3_host_dp_pair.cpp")]
    lib = os.path.join(ROOT, "soap3-dp_amd")
    subprocess.check_call(["g++", "-O2", "-mssse3", "-std=c++17", "-pthread", "-o", TOOL] + src + ["-L" + lib, "-ls3dp", "-lz", "-lm", "-Wl,-rpath," + lib])
    return TOOL


def dump(tool, chunk, L, files, tmp_path, tag, threads="8", generic=False):
    out = tmp_path / ("dump_%s.txt" % tag)
    env = dict(os.environ, S3DP_HOST_THREADS=threads)
    if generic:
        env["S3DP_GENERIC_PARSER"] = "1"
    r = subprocess.run([tool, str(chunk), str(L), str(out)] + [str(f) for f in files], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr
    return open(out).read().split("\n"), int(r.stderr.split()[0])


def write_fastq(path, n, rng, lo, hi, at_quals=False, gz=False):
    op = gzip.open if gz else open
    with op(path, "wt") as f:
        for i in range(n):
            L = int(rng.integers(lo, hi + 1))
            seq = "".join("ACGTN"[int(x)] for x in rng.integers(0, 5, size=L))
            q = "".join(chr(33 + int(x)) for x in rng.integers(0, 41, size=L))
            if at_quals and i % 3 == 0:
                q = "@" + q[1:]                     # a quality line that looks like a header
            f.write("@read%d/1 comment %d\n%s\n+\n%s\n" % (i, i, seq, q))


def test_chunk_size_does_not_change_the_reads(tool, tmp_path):
    rng = np.random.default_rng(5)
    fq = tmp_path / "a.fq"
    write_fastq(fq, 3000, rng, 30, 100, at_quals=True)
    whole, c0 = dump(tool, 1 << 30, 100, [fq], tmp_path, "whole")
    assert c0 == 1 and len([l for l in whole if l]) == 3000
    for chunk, thr in ((64, "8"), (1000, "3"), (2999, "1"), (3000, "8"), (4096, "16")):
        got, c = dump(tool, chunk, 100, [fq], tmp_path, "c%d" % chunk, thr)
        assert got == whole, "chunk size %d changes the reads" % chunk
        assert c == (3000 + chunk - 1) // chunk
    gzf = tmp_path / "a.fq.gz"
    with open(fq, "rb") as f, gzip.open(gzf, "wb") as g:
        g.write(f.read())
    got, _ = dump(tool, 700, 100, [gzf], tmp_path, "gz")
    assert got == whole


def test_line_by_line_parser_equals_the_general_one(tool, tmp_path):
    """Four-line FASTQ records are taken line by line (memchr, SIMD packing); everything else goes through the byte-by-byte parser that follows the
    reference's reader.  Both must make the same reads of every file: S3DP_GENERIC_PARSER=1 switches the short cut off."""
    rng = np.random.default_rng(23)
    recs = []
    for i in range(4000):
        L = int(rng.integers(1, 131))
        seq = "".join("ACGTNacgtnUuRY"[int(x)] for x in rng.integers(0, 14, size=L))
        q = "".join(chr(33 + int(x)) for x in rng.integers(0, 60, size=L))
        name = "r%d" % i
        kind = i % 16
        if kind == 1: name += "/1"
        elif kind == 2: name += "/2 trailing words"
        elif kind == 3: name += "\ttabbed"
        elif kind == 4: name = name + "x" * 80                       # longer than MaxLenReadName
        elif kind == 5: name = name + "y" * 61 + "/1"                # "/1" falls behind the name limit
        plus = "+" if kind != 6 else "+" + name
        if kind == 7: q = "@" + q[1:]
        elif kind == 8: q = q + "IIII"                               # quality line longer than the sequence
        elif kind == 9 and L > 3: q = q[:L // 2]                     # ... and shorter
        elif kind == 10 and L > 3: q = "@" + q[1:L // 2]
        if kind == 11 and L > 20:
            seq = seq[:L // 2] + "\n" + seq[L // 2:]                  # wrapped sequence: the general parser's business
        nl = "\r\n" if kind == 13 else "\n"
        recs.append("@%s%s%s%s%s%s%s%s" % (name, nl, seq, nl, plus, nl, q, nl))
    text = "".join(recs)
    for tag, body in (("nl", text), ("no_final_newline", text[:-1]), ("blank_lines", text.replace("@r100\n", "\n\n@r100\n"))):
        fq = tmp_path / (tag + ".fq")
        with open(fq, "w", newline="") as f:
            f.write(body)
        for L in (100, 130):
            gen, _ = dump(tool, 1 << 30, L, [fq], tmp_path, tag + "_gen%d" % L, generic=True)
            assert len([l for l in gen if l]) >= 3900
            for chunk, thr in ((1 << 30, "8"), (257, "5"), (1000, "1")):
                got, _ = dump(tool, chunk, L, [fq], tmp_path, tag + "_fast%d_%d" % (L, chunk), thr)
                assert got == gen, "the line-by-line parser differs (%s, L = %d, chunks of %d)" % (tag, L, chunk)
    # a character the general parser stops at inside a sequence line (it ends the sequence there): such a record is not the line-by-line parser's
    odd = tmp_path / "odd.fq"
    with open(odd, "w") as f:
        f.write("@a\nACGTACGTACGTACGTACGT+ACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIII\n@b\nACG>T\n+\nIIIII\n@c\nACGT\n+\nIIII\n")
    gen, _ = dump(tool, 1 << 30, 100, [odd], tmp_path, "odd_gen", generic=True)
    got, _ = dump(tool, 1 << 30, 100, [odd], tmp_path, "odd_fast")
    assert got == gen and gen[0].split(" ")[:2] == ["a", "20"] and gen[1].split(" ")[:2] == ["b", "3"]
    # pairs through both
    f1, f2 = tmp_path / "q_1.fq", tmp_path / "q_2.fq"
    write_fastq(f1, 900, rng, 50, 100, at_quals=True)
    write_fastq(f2, 900, rng, 90, 100)
    gen, _ = dump(tool, 500, 100, [f1, f2], tmp_path, "q_gen", generic=True)
    got, _ = dump(tool, 500, 100, [f1, f2], tmp_path, "q_fast")
    assert got == gen


def test_fasta_and_pairs(tool, tmp_path):
    rng = np.random.default_rng(9)
    fa = tmp_path / "a.fa"
    with open(fa, "w") as f:
        for i in range(500):
            seq = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(40, 150))))
            f.write(">s%d\n" % i + "\n".join(seq[k:k + 60] for k in range(0, len(seq), 60)) + "\n")
    whole, _ = dump(tool, 1 << 30, 160, [fa], tmp_path, "fa_whole")
    got, c = dump(tool, 96, 160, [fa], tmp_path, "fa_96")
    assert got == whole and c == 6
    # pairs: file 2 has longer records and fewer of them -- the loader stops at the shorter file, chunk by chunk the same reads
    f1, f2 = tmp_path / "p_1.fq", tmp_path / "p_2.fq.gz"
    write_fastq(f1, 1200, rng, 50, 60)
    write_fastq(f2, 1000, rng, 90, 100, gz=True)
    whole, _ = dump(tool, 1 << 30, 100, [f1, f2], tmp_path, "pe_whole")
    assert len([l for l in whole if l]) == 2000
    for chunk in (64, 500, 2000):
        got, c = dump(tool, chunk, 100, [f1, f2], tmp_path, "pe_%d" % chunk)
        assert got == whole


def test_golden_reads_through_the_chunker(tool, tmp_path):
    whole, _ = dump(tool, 1 << 30, 100, [os.path.join(GOLDEN, "g4", "reads_1.fq"), os.path.join(GOLDEN, "g4", "reads_2.fq")], tmp_path, "g4_whole")
    got, c = dump(tool, 512, 100, [os.path.join(GOLDEN, "g4", "reads_1.fq"), os.path.join(GOLDEN, "g4", "reads_2.fq")], tmp_path, "g4_512")
    assert got == whole and c == 5


def _sam2bam():
    host = os.path.join(ROOT, "soap3-dp_amd", "host")
    exe = os.path.join(ROOT, "tests", "tools", "sam2bam")
    src = [os.path.join(ROOT, "tests", "tools", "sam2bam.cpp"), os.path.join(host, "s3_bam.cpp")]
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + host, "-o", exe] + src + ["-lz"])
    return exe


_COMP = str.maketrans("ACGTN", "TGCAN")


def test_bam_input_gives_the_reads_of_the_fastq(tool, tmp_path):
    """-bam (loadBAMReads, QueryParser.cpp:27): the reads of a BAM file are the reads of the FASTQ it was made from, whatever the chunk size; a record
    that carries the reverse-strand flag holds the reverse complement and is turned back.  (The BAM is written by the product's own BAM writer, which
    tests/test_host_bam_cpu.py pins against the reference's samtools.)"""
    rng = np.random.default_rng(11)
    fq = tmp_path / "a.fq"
    write_fastq(fq, 2500, rng, 30, 100)
    sam = tmp_path / "a.sam"
    with open(fq) as f, open(sam, "w") as g:
        g.write("@HD\tVN:1.3\tSO:unsorted\n@SQ\tSN:chr1\tLN:1000\n")
        lines = f.read().split("\n")
        for i in range(0, len(lines) - 3, 4):
            name, seq, qual = lines[i][1:].split()[0], lines[i + 1], lines[i + 3]
            if (i // 4) % 3 == 1:                                   # an "aligned to the reverse strand" record: sequence and qualities as on the reference strand
                g.write("%s\t16\tchr1\t5\t30\t%dM\t*\t0\t0\t%s\t%s\n" % (name, len(seq), seq.translate(_COMP)[::-1], qual[::-1]))
            else:
                g.write("%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (name, seq, qual))
    bam = tmp_path / "a.bam"
    subprocess.check_call([_sam2bam(), str(sam), str(bam)])
    whole, _ = dump(tool, 1 << 30, 100, [fq], tmp_path, "fq")
    for chunk in (1 << 30, 1000, 64):
        got, c = dump(tool, chunk, 100, ["bam:" + str(bam)], tmp_path, "bam%d" % chunk)
        assert c == (2500 + chunk - 1) // chunk
        assert len(got) == len(whole)
        for a, b in zip(whole, got):
            # (the FASTQ name keeps "/1"; N is read as G on both paths: charMap)
            assert a.split(" ")[1:] == b.split(" ")[1:], (a[:80], b[:80])
            assert a.split(" ")[0].split("/")[0] == b.split(" ")[0].split("/")[0]
    # pairs: the mates behind each other in one file; an odd record at the end is dropped
    got, _ = dump(tool, 1000, 100, ["bampair:" + str(bam)], tmp_path, "bampair")
    assert [l.split(" ")[1:] for l in got if l] == [l.split(" ")[1:] for l in whole if l][:2500]


def test_bam_files_written_by_the_reference_library(tool, tmp_path):
    """tests/golden/bam/*.bam were written by the REFERENCE's samtools-0.1.18 (oracle/ref_sam2bam.c over its library, tests/golden/make_bam_fixtures.py) from the
    reads of g2 (every third record flagged as aligned to the reverse strand: sequence reverse-complemented, qualities reversed) and the pairs of g4: the
    loader gives the reads of the FASTQ files, whatever the chunk size.  (The test above feeds it BAM files of the product's own writer.)"""
    bam = os.path.join(GOLDEN, "bam")
    whole, _ = dump(tool, 1 << 30, 120, [os.path.join(GOLDEN, "g2", "reads.fq")], tmp_path, "g2fq")
    n = len([l for l in whole if l])
    assert n == 1500
    for chunk in (1 << 30, 512, 64):
        got, c = dump(tool, chunk, 120, ["bam:" + os.path.join(bam, "se.bam")], tmp_path, "g2bam%d" % chunk)
        assert c == (n + chunk - 1) // chunk
        assert [l.split(" ")[1:] for l in got] == [l.split(" ")[1:] for l in whole]
        assert [l.split(" ")[0].split("/")[0] for l in got if l] == [l.split(" ")[0].split("/")[0] for l in whole if l]
    wholep, _ = dump(tool, 1 << 30, 100, [os.path.join(GOLDEN, "g4", "reads_1.fq"), os.path.join(GOLDEN, "g4", "reads_2.fq")], tmp_path, "g4fq")
    for chunk in (1 << 30, 500):
        got, _ = dump(tool, chunk, 100, ["bampair:" + os.path.join(bam, "pe.bam")], tmp_path, "g4bam%d" % chunk)
        assert [l.split(" ")[1:] for l in got] == [l.split(" ")[1:] for l in wholep]
        assert [l.split(" ")[0].split("/")[0] for l in got if l] == [l.split(" ")[0].split("/")[0] for l in wholep if l]
