#!/usr/bin/env python3
"""The paired-end BWT stage (s3_batch_align_pair, DP stages off) on bench.py's repeat-rich genome, resident batch: device milliseconds per pass and
the counters of the search.  Measurement aid for profiling (`rocprofv3 --kernel-trace --stats -- python3 tools/bench_pairs.py ...`); the figure the
driver sees is bench.py's `extras.c3_pairs`.  (GPU only.)"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "soap3-dp_amd", "py"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-bp", type=int, default=3_100_000_000)
    ap.add_argument("--pairs", type=int, default=2_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--repeat-frac", type=float, default=0.5)
    ap.add_argument("--index-dir", default=None)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--mismatch", type=int, default=3)
    ap.add_argument("--counters", type=int, default=1, help="0: no counted pass (profiling runs: the counted pass launches the COUNT instantiations of the kernels)")
    a = ap.parse_args()
    import torch
    import bench
    import s3dp
    from s3dp import builder
    dev = torch.device("cuda:0")
    codes, _ = bench.make_genome(a.genome_bp, 1, dev, a.repeat_frac)
    label = "repeat-rich" if a.repeat_frac > 0 else "uniform"
    prefix = os.path.join(a.index_dir, "bench_%s_%d.index" % (label, a.genome_bp)) if a.index_dir else None
    if prefix and os.path.exists(prefix + ".sa"):
        hi = s3dp.HostIndex.from_files(prefix)
    else:
        hi = builder.build_host_index(codes.cpu().numpy(), device=dev, verbose=False)
    L = a.read_len
    reads2d = bench.make_pairs(codes, a.pairs, L, 4242, dev)
    wpq = s3dp.word_per_query(L)
    q = bench.pack_on_device(reads2d, wpq)
    del reads2d, codes
    torch.cuda.empty_cache()
    ix = s3dp.Index(hi, device=0)
    lens = np.full(2 * a.pairs, L, dtype=np.uint32)
    b = s3dp.Batch(ix, 2 * a.pairs, wpq)
    b.upload(q, lens)
    cnt = {}
    if a.counters:
        b.enable_counters(True)
        b.align_pair(a.mismatch, 8000, 2, 1, 500, 1000)
        b.sync()
        cnt = b.counters()
        b.enable_counters(False)
    b.align_pair(a.mismatch, 8000, 2, 1, 500, 1000)
    b.sync()
    t0 = time.perf_counter()
    dev_ms = search_ms = 0.0
    for _ in range(a.steps):
        b.align_pair(a.mismatch, 8000, 2, 1, 500, 1000)
        b.sync()
        tot, s_ms, k = b.timing()
        dev_ms += tot
        search_ms += s_ms
    dt = (time.perf_counter() - t0) / a.steps
    pairs = b.fetch_pairs()[0]
    print(json.dumps({"pairs": a.pairs, "read_len": L, "ms_per_pass": dt * 1e3, "device_ms_per_pass": dev_ms / a.steps, "search_kernel_ms_per_pass": search_ms / a.steps,
                      "properly_paired": int((pairs["phase"] > 0).sum()), "counters": {k: int(v) for k, v in cnt.items()}}), flush=True)
    b.close()
    ix.close()


if __name__ == "__main__":
    main()
