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


def test_single_reads_equal_the_oracle_at_scale(workload):
    w = workload
    env = dict(os.environ, S3DP_CHUNK_READS="40000")
    out = subprocess.run([BIN, "single", "g.fa.index", "se.fq", "-L", "100", "-s", "4", "-b", "2", "-o", "gpu_se"], cwd=w, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    ora = subprocess.run([oracle_tool(), "single", "g.fa.index", "se.fq", "-L", "100", "-s", "4", "-o", "cpu_se"], cwd=w, capture_output=True, text=True, timeout=1800)
    assert ora.returncode == 0, ora.stderr[-2000:]
    mine_h, mine = sam_records(open(w / "gpu_se.gout.1").read())
    ref_h, ref = sam_records(open(w / "cpu_se.gout.1").read())
    assert mine_h == ref_h and len(ref) == 150000 and set(mine) == set(ref)
    bad = [k for k in ref if ref[k] != mine[k]]
    assert not bad, "%d of %d records differ, first:\n%s\n%s" % (len(bad), len(ref), ref[bad[0]], mine[bad[0]])
    multi = sum(1 for v in ref.values() if "\tXA:Z:" in v)
    assert multi > 3000                                                    # the repeat families are in it


def test_pairs_equal_the_oracle_at_scale(workload):
    w = workload
    env = dict(os.environ, S3DP_CHUNK_READS="30000")
    out = subprocess.run([BIN, "pair", "g.fa.index", "pe_1.fq", "pe_2.fq", "-L", "100", "-u", "500", "-v", "200", "-s", "3", "-b", "2", "-o", "gpu_pe"], cwd=w,
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    ora = subprocess.run([oracle_tool(), "pair", "g.fa.index", "pe_1.fq", "pe_2.fq", "-L", "100", "-u", "500", "-v", "200", "-s", "3", "-o", "cpu_pe"], cwd=w,
                         capture_output=True, text=True, timeout=1800)
    assert ora.returncode == 0, ora.stderr[-2000:]
    mine = sam_pair_records(open(w / "gpu_pe.gout.1").read())
    ref = sam_pair_records(open(w / "cpu_pe.gout.1").read())
    assert mine[0] == ref[0] and set(mine[1]) == set(ref[1]) and len(ref[1]) >= 40000
    bad = [k for k in ref[1] if ref[1][k] != mine[1][k]]
    assert not bad, "%d of %d records differ, first:\n%s\n%s" % (len(bad), len(ref[1]), ref[1][bad[0]], mine[1][bad[0]])
