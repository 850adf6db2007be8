"""Multi-GPU plumbing: one process per GPU, reads sharded by rank, index replicated, and ONE collective --
the final reduce of the alignment counters (the `[Main] Overall ...` numbers of SOAP3-DP.cu:816-830).
Backend-agnostic: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
import os

import numpy as np


def world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_bounds(n_total, rank, world_size, multiple=2):
    """contiguous shard [lo, hi) of n_total reads for `rank`; shard sizes are multiples of `multiple`
    (mates of a pair are adjacent reads and must stay together: CPUfunctions.cpp:1982,2029)"""
    units = n_total // multiple
    base, extra = divmod(units, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    lo, hi = lo * multiple, hi * multiple
    if rank == world_size - 1:
        hi = n_total
    return lo, hi


COUNTER_KEYS = ("reads", "readsAligned", "alignments", "rankQueries", "records")


def reduce_counters(counters, device=None):
    """sum the per-rank counter dict over all ranks (no-op without an initialised process group)"""
    import torch
    import torch.distributed as dist
    vec = torch.tensor([int(counters.get(k, 0)) for k in COUNTER_KEYS], dtype=torch.int64, device=device)      # (64-bit integers: rank-query counts pass 2^53)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return {k: int(v) for k, v in zip(COUNTER_KEYS, vec.tolist())}


def sum_dict_over_ranks(d, device=None):
    """sum a dict of integers over all ranks, and gather every rank's own values: returns (sums, per_rank list of dicts).  The paired leg of bench.py:
    every rank's command line reports its shard's counts (the `[Main] Number of pairs ...` lines, SOAP3-DP.cu:816-830); the job's are their sums."""
    import torch
    import torch.distributed as dist
    keys = sorted(d)
    vec = torch.tensor([int(d[k]) for k in keys], dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return dict(d), [dict(d)]
    ws = dist.get_world_size()
    parts = [torch.zeros_like(vec) for _ in range(ws)]
    dist.all_gather(parts, vec)
    dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return {k: int(v) for k, v in zip(keys, vec.tolist())}, [{k: int(v) for k, v in zip(keys, p.tolist())} for p in parts]


def max_over_ranks(x, device=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
