#!/usr/bin/env python3
"""Fixtures for the FASTA front end of the index builder (s3dp/fasta.py): FASTA files built to hit every branch of the reference's parser
(HSPParseFASTAToPacked, 2bwt-lib/HSP.c:677-1086), run through the REAL reference builder (oracle/_ref/soap3-dp-builder + BGS-Build, built by
oracle/ref.mk); kept are the FASTA files and what the reference wrote for them: .ann .amb .tra .pac (small text / packed files) and, for the first
case, the whole index (.bwt .fmv.gpu .rev.* .sa) for the GPU build test.  Run from the repository root in the build container:
    python3 tests/golden/make_fasta_fixtures.py
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(ROOT, "tests", "golden", "fasta")


def rnd(rng, n):
    return "".join("ACGT"[c] for c in rng.integers(0, 4, size=n))


def wrap(s, w=60, eol="\n"):
    return eol.join(s[i:i + w] for i in range(0, len(s), w)) + eol


def cases():
    rng = np.random.default_rng(77)
    out = {}
    # f1: runs of 9 / 10 / 11 N, one inside a line, one across line breaks; IUPAC codes; lower case; three sequences; a gi| name; a comment after the name
    s1 = rnd(rng, 700) + "N" * 9 + rnd(rng, 333) + "N" * 10 + rnd(rng, 401) + "n" * 11 + rnd(rng, 290).lower() + "RYKM" + rnd(rng, 120)
    s2 = rnd(rng, 515) + "N" * 37 + rnd(rng, 800) + "NNNNNNNNNNNN" + rnd(rng, 64)
    s3 = rnd(rng, 2000)
    out["f1"] = ">gi|12345|ref|chrA some comment\n" + wrap(s1) + ">chrB\tcomment after a tab\n" + wrap(s2, 70) + ">chrC\n" + wrap(s3, 50)
    # f2: runs at the very start and the very end of sequences (a removed run AT a sequence boundary), a sequence that is ambiguous only in a short run,
    # a '>' in the middle of a line, tabs inside the sequence, no newline at the end of the file
    t1 = "N" * 25 + rnd(rng, 900) + "N" * 15
    t2 = "N" * 12 + rnd(rng, 600) + "NNN" + rnd(rng, 50)
    t3 = rnd(rng, 300) + "\t" + rnd(rng, 300)
    t4 = rnd(rng, 450)
    out["f2"] = ">s1\n" + wrap(t1) + ">s2 x\n" + wrap(t2, 80) + ">s3\n" + wrap(t3, 61) + "ACGTACGTAC>s4 mid-line start\n" + wrap(t4).rstrip("\n")
    # f3: carriage returns (every CR is an ambiguous character: lines of 60 make runs of one), a name longer than 256 characters, blanks in the sequence
    u1 = rnd(rng, 1200)
    u2 = rnd(rng, 500) + " " * 10 + rnd(rng, 500) + " " + rnd(rng, 77)
    out["f3"] = ">" + "L" * 300 + " tail\r\n" + wrap(u1, 60, "\r\n") + ">u2\n" + wrap(u2, 100)
    # f4: one long sequence with many removed runs (several grid cells would need > 262144 characters: kept small, one cell) and total % 4 == 0
    parts = []
    for k in range(12):
        parts.append(rnd(rng, 397 + k))
        parts.append("N" * (10 + 3 * k))
    body = "".join(parts) + rnd(rng, 300)
    pad = (-sum(1 for c in body if c in "ACGT")) % 4
    out["f4"] = ">only\n" + wrap(body + rnd(rng, pad), 73)
    # f5 (built with -U, MaskLowerCase): soft-masked stretches, a masked base in front of a long run of N (the run swallows it), masked bases in front of nine
    # N and behind them, lower-case n runs, mixed-case stretches, IUPAC codes -- a masked letter opens a run, only real ambiguity codes extend one (HSP.c:849-870)
    def low(n):
        return rnd(rng, n).lower()
    v1 = (rnd(rng, 300) + low(25) + rnd(rng, 200) + "N" * 15 + rnd(rng, 100) + "a" + "N" * 12 + rnd(rng, 50) + "NNNN" + rnd(rng, 40) + low(3) + "N" * 9 + "g" +
          rnd(rng, 60))
    v2 = rnd(rng, 120) + "n" * 11 + rnd(rng, 80) + "".join("acgtn"[c] for c in rng.integers(0, 5, size=40)) + rnd(rng, 100) + "RYK" + rnd(rng, 30) + "t" + "n" * 9 + rnd(rng, 21)
    out["f5"] = ">m1 soft-masked\n" + wrap(v1) + ">m2\n" + wrap(v2, 47)
    return out


MASKED = {"f5"}                                   # built with the reference builder's -U


def main():
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])
    for name, text in cases().items():
        if only and name not in only:
            continue
        with tempfile.TemporaryDirectory() as td:
            fa = os.path.join(td, name + ".fa")
            with open(fa, "w", newline="") as f:
                f.write(text)
            shutil.copy(os.path.join(REF, "soap3-dp-builder.ini"), td)
            subprocess.check_call([os.path.join(REF, "soap3-dp-builder"), name + ".fa"] + (["-U"] if name in MASKED else []), cwd=td, stdout=subprocess.DEVNULL)
            keep = [".ann", ".amb", ".tra", ".pac"]
            if name == "f1":
                subprocess.check_call([os.path.join(REF, "BGS-Build"), name + ".fa.index"], cwd=td, stdout=subprocess.DEVNULL)
                keep += [".bwt", ".rev.bwt", ".fmv.gpu", ".rev.fmv.gpu", ".sa"]
            shutil.copy(fa, os.path.join(OUT, name + ".fa"))
            for ext in keep:
                shutil.copy(fa + ".index" + ext, os.path.join(OUT, name + ".index" + ext))
        print("fixture", name, "written")


if __name__ == "__main__":
    sys.exit(main())
