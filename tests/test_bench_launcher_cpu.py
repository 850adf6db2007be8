"""`python bench.py --gpus N` starts its own ranks.  No GPU here: --launch-selftest runs the plumbing of the N > 1
path -- child ranks under torch.distributed.run, gloo process group, shard bounds, counters' reduce, max-over-ranks timing, one relayed JSON line."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=600)


def test_gpus_2_spawns_two_ranks_and_relays_one_line():
    out = run(["--gpus", "2", "--launch-selftest", "1", "--reads", "1001", "--steps", "3", "--warmup", "1"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["selftest"] is True and res["scaling"] == "weak"
    assert res["reads"] == 2002 and res["alignments"] == 4004                 # both shards counted, summed by the all-reduce
    assert res["max_seconds"] >= 0.02                                          # the slower rank's time


def test_gpus_2_without_two_devices_fails_on_the_device_count():
    out = run(["--gpus", "2"])
    assert out.returncode == 2
    assert "HIP device" in out.stderr and "WORLD_SIZE" not in out.stderr


def test_rank_count_mismatch_is_refused():
    out = run(["--gpus", "2", "--launch-selftest", "1"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode == 2 and "WORLD_SIZE is 1" in out.stderr
