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
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", TOOL] + src + ["-L" + lib, "-ls3dp", "-lz", "-lm", "-Wl,-rpath," + lib])
    return TOOL


def dump(tool, chunk, L, files, tmp_path, tag, threads="8"):
    out = tmp_path / ("dump_%s.txt" % tag)
    env = dict(os.environ, S3DP_HOST_THREADS=threads)
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
