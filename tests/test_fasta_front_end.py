"""The FASTA front end of the index builder (s3dp/fasta.py) against what the reference builder wrote for the same FASTA files: .ann .amb .tra .pac
byte for byte (fixtures: tests/golden/make_fasta_fixtures.py ran oracle/_ref/soap3-dp-builder; g1's genome.fa and index files come from
make_fixtures.sh).  The whole index of f1 (suffix array, both BWTs, GPU-2BWT occurrence tables) is compared with the CPU build here and with the
GPU build in the -m gpu test."""
import os
import re

import numpy as np
import pytest

from s3dp import fasta
from conftest import GOLDEN

CASES = [("fasta/f1.fa", "fasta/f1.index"), ("fasta/f2.fa", "fasta/f2.index"), ("fasta/f3.fa", "fasta/f3.index"), ("fasta/f4.fa", "fasta/f4.index"),
         ("g1/genome.fa", "g1/index"), ("fasta/f5.fa", "fasta/f5.index")]
MASKED = {"fasta/f5.fa"}                          # built by the reference builder with -U (MaskLowerCase)


def ref_seed(prefix):
    return int(open(prefix + ".ann").readline().split()[2])          # (the reference stamps a seed into the first line; it decides nothing else)


@pytest.mark.parametrize("fa,prefix", CASES)
def test_annotation_translate_and_packed_text_match_the_reference_builder(tmp_path, fa, prefix):
    prefix = os.path.join(GOLDEN, prefix)
    p = fasta.parse_fasta(os.path.join(GOLDEN, fa), seed=ref_seed(prefix), mask_lower_case=fa in MASKED)
    fasta.write_annotation_files(p, str(tmp_path / "x"))
    for ext in (".ann", ".amb", ".tra"):
        assert open(str(tmp_path / "x") + ext, "rb").read() == open(prefix + ext, "rb").read(), ext
    assert fasta.pac_bytes(p.codes) == open(prefix + ".pac", "rb").read()
    assert sum(s[3] for s in p.seqs) == p.n


def test_cases_cover_the_parser_branches():
    """what the fixtures were built to contain, checked on the parse results (so that a regenerated fixture cannot silently lose a branch)"""
    f1 = fasta.parse_fasta(os.path.join(GOLDEN, "fasta/f1.fa"))
    assert [s[0] for s in f1.seqs] == [12345, 0, 0] and [s[1] for s in f1.seqs] == ["gi|12345|ref|chrA", "chrB", "chrC"]
    assert len(f1.amb) == 4                                          # runs of 10, 11, 37, 12 removed; the run of 9 and "RYKM" became G
    f2 = fasta.parse_fasta(os.path.join(GOLDEN, "fasta/f2.fa"))
    assert [s[1] for s in f2.seqs] == ["s1", "s2", "s3", "s4"]        # s4 starts in the middle of a line
    assert any(a[0] == s[2] for a in f2.amb for s in f2.seqs[1:])     # a removed run exactly at a sequence start
    f3 = fasta.parse_fasta(os.path.join(GOLDEN, "fasta/f3.fa"))
    assert len(f3.seqs[0][1]) == 256 and f3.seqs[0][3] == 1200 + 20  # name cut at 256; every CR is a character (a run of one: G)
    f4 = fasta.parse_fasta(os.path.join(GOLDEN, "fasta/f4.fa"))
    assert f4.n % 4 == 0 and len(f4.amb) == 12


def test_mask_lower_case():
    """-U as the reference builder applies it (HSP.c:849-870): a masked letter opens a run of ambiguous characters, only real ambiguity codes extend one --
    a soft-masked stretch becomes 'G' base by base and stays in the text; fixture f5 is the reference builder's own output with -U"""
    p = fasta.parse_fasta(b">x\nACGTacgtacgtacgtACGT\n", mask_lower_case=True)
    assert p.n == 20 and p.amb == [] and p.codes.tolist() == [0, 1, 2, 3] + [2] * 12 + [0, 1, 2, 3]
    q = fasta.parse_fasta(b">x\nACGTaNNNNNNNNNNNNACGTcgNNNNNNNNNtACGT\n", mask_lower_case=True)
    # 'a' + 12 N: one run of 13, removed; 'c': a run of one, G; 'g' + 9 N: a run of ten, removed; 't': a run of one, G
    assert q.n == 14 and q.amb == [(4, 16), (9, 31)] and q.codes.tolist() == [0, 1, 2, 3, 0, 1, 2, 3, 2, 2, 0, 1, 2, 3]
    prefix = os.path.join(GOLDEN, "fasta/f5.index")
    f5 = fasta.parse_fasta(os.path.join(GOLDEN, "fasta/f5.fa"), mask_lower_case=True)
    assert fasta.pac_bytes(f5.codes) == open(prefix + ".pac", "rb").read()
    assert len(f5.amb) >= 4
    with pytest.raises(ValueError):
        fasta.parse_fasta(b"ACGT\n")


def test_index_of_a_fasta_file_equals_the_reference_tools(tmp_path):
    """FASTA -> index files with the CPU build: every file the reference builder + BGS-Build wrote for f1, byte for byte"""
    prefix = os.path.join(GOLDEN, "fasta/f1.index")
    out = str(tmp_path / "f1.index")
    fasta.build_index_files(os.path.join(GOLDEN, "fasta/f1.fa"), out, device="cpu", seed=ref_seed(prefix), with_lkt=False)
    for ext in (".ann", ".amb", ".tra", ".pac", ".bwt", ".rev.bwt", ".fmv.gpu", ".rev.fmv.gpu", ".sa"):
        assert open(out + ext, "rb").read() == open(prefix + ext, "rb").read(), ext


@pytest.mark.gpu
@pytest.mark.parametrize("fa,prefix", [("fasta/f1.fa", "fasta/f1.index"), ("g1/genome.fa", "g1/index"), ("g2/genome.fa", "g2/index"), ("g7/genome.fa", "g7/index")])
def test_gpu_build_from_fasta_equals_the_reference_tools(tmp_path, fa, prefix):
    """`python -m s3dp.fasta ref.fa` on the GPU (suffix sort, BWTs, occurrence tables on cuda:0): byte-identical to soap3-dp-builder + BGS-Build"""
    import torch
    assert torch.cuda.is_available()
    prefix = os.path.join(GOLDEN, prefix)
    out = str(tmp_path / "x.index")
    fasta.build_index_files(os.path.join(GOLDEN, fa), out, device="cuda", seed=ref_seed(prefix), with_lkt=False)
    for ext in (".ann", ".amb", ".tra", ".pac", ".bwt", ".rev.bwt", ".fmv.gpu", ".rev.fmv.gpu", ".sa"):
        assert open(out + ext, "rb").read() == open(prefix + ext, "rb").read(), ext
