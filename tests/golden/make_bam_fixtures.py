#!/usr/bin/env python3
"""BAM files for the `-bam` input tests, written by the REFERENCE's samtools library (oracle/_ref/ref_sam2bam, built by oracle/ref.mk from
/root/reference/samtools-0.1.18): tests/golden/bam/se.bam = the reads of g2/reads.fq as unaligned records, every third one as a record "aligned to
the reverse strand" (flag 16: sequence reverse-complemented, qualities reversed -- the loader has to turn it back, loadBAMReads QueryParser.cpp:27-171);
pe.bam = the pairs of g4/reads_1.fq + reads_2.fq, mates behind each other (flags 77 / 141).  Runs in the authoring container only."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLDEN = os.path.join(ROOT, "tests", "golden")
TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_sam2bam")
COMP = str.maketrans("ACGTN", "TGCAN")


def fastq(path):
    lines = open(path).read().split("\n")
    for i in range(0, len(lines) - 3, 4):
        yield lines[i][1:].split()[0], lines[i + 1], lines[i + 3]


def main():
    out = os.path.join(GOLDEN, "bam")
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        sam = os.path.join(tmp, "se.sam")
        with open(sam, "w") as f:
            f.write("@HD\tVN:1.3\tSO:unsorted\n@SQ\tSN:chr1\tLN:60000\n")
            for k, (name, seq, qual) in enumerate(fastq(os.path.join(GOLDEN, "g2", "reads.fq"))):
                if k % 3 == 1:
                    f.write("%s\t16\tchr1\t%d\t30\t%dM\t*\t0\t0\t%s\t%s\n" % (name, 5 + k, len(seq), seq.translate(COMP)[::-1], qual[::-1]))
                else:
                    f.write("%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (name, seq, qual))
        subprocess.check_call([TOOL, sam, os.path.join(out, "se.bam")])
        sam = os.path.join(tmp, "pe.sam")
        with open(sam, "w") as f:
            f.write("@HD\tVN:1.3\tSO:unsorted\n@SQ\tSN:chr1\tLN:400000\n")
            for (n1, s1, q1), (n2, s2, q2) in zip(fastq(os.path.join(GOLDEN, "g4", "reads_1.fq")), fastq(os.path.join(GOLDEN, "g4", "reads_2.fq"))):
                name = n1[:-2] if n1.endswith("/1") else n1
                f.write("%s\t77\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (name, s1, q1))
                f.write("%s\t141\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (name, s2, q2))
        subprocess.check_call([TOOL, sam, os.path.join(out, "pe.bam")])
    for f in sorted(os.listdir(out)):
        print(f, os.path.getsize(os.path.join(out, f)))


if __name__ == "__main__":
    sys.exit(main())
