"""-m gpu: several devices in the product (s3_group_*, `soap3-dp -c a,b`), exercised on ONE GPU: two group members on device 0 take the chunks of
the input as they come free, write their own output files, and their counters go through s3_group_allreduce.  The union of the members' records
must be the single-device run's (= the reference's golden files).  Also: the RCCL call path with a one-member group, and the torch.distributed
(nccl = RCCL) path of bench.py with world size 1."""
import gzip
import os
import re
import shutil
import subprocess
import sys
from collections import Counter

import numpy as np
import pytest

import s3dp
from conftest import GOLDEN, ROOT, golden_reads, host_index
from test_gpu_dp_pipeline import records

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "soap3-dp_amd", "soap3-dp")


def union(tmp_path, names):
    hdrs, rec = [], Counter()
    for n in names:
        p = tmp_path / n
        if os.path.exists(p):
            h, r = records(open(p).read())
            hdrs.append(h)
            rec += r
    return hdrs, rec


def test_cli_single_reads_on_two_members(tmp_path):
    shutil.copy(os.path.join(GOLDEN, "g1", "reads.fq"), tmp_path / "g1.fq")
    env = dict(os.environ, S3DP_CHUNK_READS="256")
    out = subprocess.run([BIN, "single", os.path.join(GOLDEN, "g1", "index"), "g1.fq", "-L", "100", "-s", "4", "-b", "2", "-c", "0,0", "-o", "out"], cwd=tmp_path,
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    ref_h, ref = records(gzip.open(os.path.join(GOLDEN, "g1", "sam_s4.sam.gz"), "rt").read())
    hdrs, mine = union(tmp_path, ["out.gout.1", "out.gout.2"])
    assert len(hdrs) == 2 and all(h == ref_h for h in hdrs)
    assert mine == ref
    assert sum(records(open(tmp_path / "out.gout.2").read())[1].values()) > 0, "the second member got no chunk"
    m = re.search(r"Overall number of reads aligned: (\d+)", out.stdout)
    aligned = sum(1 for l in ref.elements() if int(l.split("\t")[1]) & 4 == 0)
    assert m and int(m.group(1)) == aligned                                   # the counters of both members, summed by the all-reduce


def test_cli_pairs_with_dp_on_two_members(tmp_path):
    """config C3's shape (fixture g7) with the DP stages on, two members"""
    for k in ("1", "2"):
        shutil.copy(os.path.join(GOLDEN, "g7", "reads_%s.fq" % k), tmp_path / ("g7_%s.fq" % k))
    env = dict(os.environ, S3DP_CHUNK_READS="512")
    out = subprocess.run([BIN, "pair", os.path.join(GOLDEN, "g7", "index"), "g7_1.fq", "g7_2.fq", "-L", "150", "-u", "500", "-v", "1", "-b", "2", "-c", "0,0",
                          "-o", "out"], cwd=tmp_path, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    for kind, files in (("gout", ["out.gout.1", "out.gout.2"]), ("dpout", ["out.dpout.1", "out.dpout.2"]), ("unpair", ["out.unpair", "out.unpair.2"])):
        ref_h, ref = records(gzip.open(os.path.join(GOLDEN, "g7", "dp_h2.%s.sam.gz" % kind), "rt").read())
        hdrs, mine = union(tmp_path, files)
        assert all(h == ref_h for h in hdrs)
        assert mine == ref, kind


def test_group_allreduce():
    hi = host_index("g1")
    g = s3dp.Group(hi, [0])                                                    # one member: nothing to exchange, the counters are the sums
    a = g.allreduce(np.array([[3, 5, 7]], dtype=np.uint64))
    assert a.tolist() == [[3, 5, 7]]
    assert g.member_of_batch(5) == 0
    g.close()
    g = s3dp.Group(hi, [0, 0])                                                 # two members on one device: summed on the host (RCCL needs distinct devices)
    a = g.allreduce(np.array([[1, 2], [10, 20]], dtype=np.uint64))
    assert a.tolist() == [[11, 22], [11, 22]]
    assert [g.member_of_batch(i) for i in range(4)] == [0, 1, 0, 1]
    # both members hold a working index
    reads = golden_reads("g1")[:256]
    q, lens, wpq = s3dp.pack_reads(reads, 100)
    res = []
    for k in range(2):
        b = s3dp.Batch(g.index(k), len(reads), wpq)
        b.upload(q, lens)
        b.align_single_allbest(2, 1000)
        res.append(b.fetch_alignments())
        b.close()
    for x, y in zip(res[0], res[1]):
        assert (x == y).all()
    g.close()


WORKER = r'''
import json, os, sys
sys.path.insert(0, os.path.join(%(root)r, "soap3-dp_amd", "py")); sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from s3dp import dist as sd
import s3dp
from conftest import host_index, golden_reads
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
ws, rank, _ = sd.world()
reads = golden_reads("g1")[:600]
lo, hi = sd.shard_bounds(len(reads), rank, ws)
q, lens, wpq = s3dp.pack_reads(reads[lo:hi], 100)
ix = s3dp.Index(host_index("g1"))
b = s3dp.Batch(ix, hi - lo, wpq)
b.upload(q, lens)
b.align_single_allbest(2, 1000)
phase, werr, offs, pos, strand, mis = b.fetch_alignments()
tot = sd.reduce_counters({"reads": hi - lo, "readsAligned": int((phase > 0).sum()), "alignments": int(offs[-1])}, device=torch.device("cuda", 0))
print("RESULT " + json.dumps(tot))
dist.destroy_process_group()
'''


def test_distributed_path_with_world_size_one(tmp_path):
    """bench.py's multi-GPU plumbing (torch.distributed, backend nccl = RCCL) around the PRODUCT's kernels, world size 1"""
    import json
    import s3o
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29633",
                          str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    tot = json.loads([l for l in out.stdout.split("\n") if l.startswith("RESULT ")][0][7:])
    reads = golden_reads("g1")[:600]
    q, lens, wpq = s3dp.pack_reads(reads, 100)
    oix = s3o.Index(host_index=host_index("g1"))
    phase, werr, offs, pos, strand, mis, st = oix.align_single_allbest(q, lens, wpq, 2, 1000)
    assert tot["reads"] == 600 and tot["readsAligned"] == int((phase > 0).sum()) and tot["alignments"] == int(offs[-1])



def test_bench_py_starts_two_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: the script spawns its ranks, rank 0 builds the index once and writes the files,
    rank 1 maps them (HostIndex.from_files(shared=True)), both align their own shard, the counters are reduced, ONE line comes back.  On a one-GPU box the
    two ranks share device 0 over gloo (S3DP_BENCH_SHARE_GPU=1; RCCL wants a device per rank): everything of the N > 1 path but the RCCL transport."""
    import json
    full = str(tmp_path / "bench_full.json")
    env = dict(os.environ, S3DP_BENCH_SHARE_GPU="1", S3DP_BENCH_FULL=full)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--genome-bp", "60000000", "--reads", "200000", "--steps", "2", "--warmup", "1",
                          "--cpu-budget", "0", "--check", "256", "--exact-rank", "0", "--stream-batch", "65536", "--extras", "0", "--uniform", "0", "--robustness", "",
                          "--c4-pairs", "40000"],
                         capture_output=True, text=True, env=env, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])                                        # the compact line the driver parses; the full object is in the side file
    assert len(lines[0]) < 8192 and line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0 and line["full_object"] == full
    assert line["roofline"]["frac"] > 0 and line["parity"]["oracle_check_bit_exact"] and line["summary"]["c4_pairs"]["pairs"] == 80000
    res = json.loads(open(full).read())
    assert res["n_gpus"] == 2 and res["value"] == pytest.approx(line["value"], rel=1e-4)
    assert res["aligned_reads"] == 400000                              # both shards, summed by the all-reduce
    assert res["truth_check"]["ok"] and res["oracle_check"]["bit_exact"]
    assert res["streamed"]["same_results_as_resident"]
    # the paired leg of an N > 1 run (BASELINE configs[3]): every rank's shard through the command line, counts summed over the ranks
    c4 = res["extras"]["c4_pairs"]
    e2e = c4["end_to_end"]
    assert "error" not in e2e, e2e
    assert c4["gpus"] == 2 and e2e["counts"]["pairs"] == 80000 and e2e["counts"]["failed"] == 0
    assert len(e2e["per_device"]) == 2 and all(r["pairs"] == 40000 and r["aligned_reads"] > 70000 for r in e2e["per_device"])
    assert e2e["counts_sum_equals_sum_of_shards"] and e2e["counts"]["aligned_reads"] == sum(r["aligned_reads"] for r in e2e["per_device"])
    assert e2e["per_device"][0]["aligned_reads"] != e2e["per_device"][1]["aligned_reads"]          # (own shards: own seeds)
    assert e2e["M_pairs_per_s_reads_in_sam_out"] > 0 and c4["truth_check"]["ok"]
    assert any(l.startswith("[summary] C4:") for l in res["summary"]["lines"])


def test_group_on_two_devices_runs_the_rccl_allreduce():
    """With two or more devices visible: s3_group_create on devices [0, 1] and the real ncclAllReduce between them (one-GPU boxes skip this; the
    two-members-on-one-device tests above cover everything but the collective itself)."""
    if s3dp.device_count() < 2:
        pytest.skip("needs two HIP devices")
    hi = host_index("g1")
    g = s3dp.Group(hi, [0, 1])
    big = (1 << 60) + 12345                                  # would not survive a trip through float64
    a = g.allreduce(np.array([[1, big, 7], [10, 1, 70]], dtype=np.uint64))
    assert a.tolist() == [[11, big + 1, 77], [11, big + 1, 77]]
    reads = golden_reads("g1")[:256]
    q, lens, wpq = s3dp.pack_reads(reads, 100)
    res = []
    for k in range(2):
        b = s3dp.Batch(g.index(k), len(reads), wpq)
        b.upload(q, lens)
        b.align_single_allbest(2, 1000)
        res.append(b.fetch_alignments())
        b.close()
    for x, y in zip(res[0], res[1]):
        assert (x == y).all()
    g.close()


def test_cli_on_two_devices(tmp_path):
    """soap3-dp -c 0,1 with two real devices: the union of the two members' files is the reference's golden output"""
    if s3dp.device_count() < 2:
        pytest.skip("needs two HIP devices")
    shutil.copy(os.path.join(GOLDEN, "g1", "reads.fq"), tmp_path / "g1.fq")
    env = dict(os.environ, S3DP_CHUNK_READS="256")
    out = subprocess.run([BIN, "single", os.path.join(GOLDEN, "g1", "index"), "g1.fq", "-L", "100", "-s", "4", "-b", "2", "-c", "0,1", "-o", "out"], cwd=tmp_path,
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-2000:]
    ref_h, ref = records(gzip.open(os.path.join(GOLDEN, "g1", "sam_s4.sam.gz"), "rt").read())
    hdrs, mine = union(tmp_path, ["out.gout.1", "out.gout.2"])
    assert all(h == ref_h for h in hdrs) and mine == ref
    m = re.search(r"Overall number of reads aligned: (\d+)", out.stdout)
    aligned = sum(1 for l in ref.elements() if int(l.split("\t")[1]) & 4 == 0)
    assert m and int(m.group(1)) == aligned                                   # summed by ncclAllReduce between the two devices
