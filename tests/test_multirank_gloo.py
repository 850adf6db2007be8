"""N > 1 path on CPU: two processes (gloo), reads sharded by rank, index replicated, one all-reduce of the
counters.  The per-rank work is done by the CPU oracle here (no GPU in this container); the sharding and the
reduce are the product's (s3dp.dist), exactly what bench.py uses with the nccl/RCCL backend."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.path.join(%(root)r, "soap3-dp_amd", "py")); sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from s3dp import dist as sd
import s3dp, s3o
from conftest import host_index, golden_reads
dist.init_process_group(backend="gloo")
ws, rank, _ = sd.world()
reads = golden_reads("g1")[:600]
lo, hi = sd.shard_bounds(len(reads), rank, ws)
mine = reads[lo:hi]
q, lens, wpq = s3dp.pack_reads(mine, 100)
ix = s3o.Index(host_index=host_index("g1"))
phase, werr, offs, pos, strand, mis, st = ix.align_single_allbest(q, lens, wpq, 2, 1000)
c = {"reads": len(mine), "readsAligned": int((phase > 0).sum()), "alignments": int(offs[-1]), "rankQueries": st.rankQueries, "records": st.hits}
tot = sd.reduce_counters(c)
t = sd.max_over_ranks(1.0 + rank)
sums, per_rank = sd.sum_dict_over_ranks({"pairs": 10 + rank, "aligned_reads": (1 << 40) + 7 * rank, "failed": 0})    # (the paired leg's shard counts)
if rank == 0:
    print("RESULT " + json.dumps({"tot": tot, "tmax": t, "lo_hi": [lo, hi], "sums": sums, "per_rank": per_rank}))
dist.destroy_process_group()
'''


def test_two_ranks_shard_and_reduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29617", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.split("\n") if l.startswith("RESULT ")][0]
    res = json.loads(line[7:])
    # single-process reference
    import s3dp
    import s3o
    from conftest import golden_reads, host_index
    reads = golden_reads("g1")[:600]
    q, lens, wpq = s3dp.pack_reads(reads, 100)
    ix = s3o.Index(host_index=host_index("g1"))
    phase, werr, offs, pos, strand, mis, st = ix.align_single_allbest(q, lens, wpq, 2, 1000)
    assert res["tot"]["reads"] == 600
    assert res["tot"]["readsAligned"] == int((phase > 0).sum())
    assert res["tot"]["alignments"] == int(offs[-1])
    assert res["tot"]["rankQueries"] == st.rankQueries
    assert res["tmax"] == 2.0
    assert res["sums"] == {"pairs": 21, "aligned_reads": (1 << 41) + 7, "failed": 0}
    assert res["per_rank"] == [{"pairs": 10, "aligned_reads": 1 << 40, "failed": 0}, {"pairs": 11, "aligned_reads": (1 << 40) + 7, "failed": 0}]


def test_shard_bounds_cover_everything_and_keep_pairs():
    from s3dp import dist as sd
    for n in (0, 2, 10, 1001, 12582912):
        for ws in (1, 2, 3, 8):
            prev = 0
            for r in range(ws):
                lo, hi = sd.shard_bounds(n, r, ws)
                assert lo == prev and lo % 2 == 0
                prev = hi
            assert prev == n
