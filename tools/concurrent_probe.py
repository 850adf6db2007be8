#!/usr/bin/env python3
"""How much of a launch's tail does a second batch on the same device hide?  The bench's 10 M resident reads as ONE batch, and as K batches of 10/K M reads
aligned by K host threads at once (half-width search grids: S3DP_BLOCKS_PER_CU=2 for K >= 2).  Prints reads/s for both.  (GPU only; measurement aid.)"""
import argparse
import os
import sys
import threading
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "soap3-dp_amd", "py"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-bp", type=int, default=3_100_000_000)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--index-dir", default=None)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    import torch
    import bench
    import s3dp
    from s3dp import builder
    dev = torch.device("cuda:0")
    codes, _ = bench.make_genome(a.genome_bp, 1, dev, 0.5)
    reads2d, _, _, _ = bench.make_reads(codes, a.reads, 100, 2026, dev)
    wpq = s3dp.word_per_query(100)
    q = bench.pack_on_device(reads2d, wpq)
    prefix = os.path.join(a.index_dir, "bench_repeat-rich_%d.index" % a.genome_bp) if a.index_dir else None
    if prefix and os.path.exists(prefix + ".sa"):
        hi = s3dp.HostIndex.from_files(prefix)
    else:
        hi = builder.build_host_index(codes.cpu().numpy(), device=dev, verbose=False)
    del codes
    torch.cuda.empty_cache()
    ix = s3dp.Index(hi, device=0)
    per = a.reads // a.k // 32 * 32
    batches = []
    for j in range(a.k):
        b = s3dp.Batch(ix, per, wpq)
        qq = np.ascontiguousarray(q.reshape(-1, wpq * 32)[j * per // 32:(j + 1) * per // 32]).ravel()      # (the packed layout interleaves groups of 32 reads)
        b.upload(qq, np.full(per, 100, dtype=np.uint32))
        batches.append(b)

    def one(b):
        b.align_single_allbest(4, 1000)
        b.sync()

    def step():
        th = [threading.Thread(target=one, args=(b,)) for b in batches]
        for t in th:
            t.start()
        for t in th:
            t.join()

    step()
    step()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    dt = (time.perf_counter() - t0) / a.steps
    print("k=%d batches of %d reads at once: %.1f ms per step, %.2f M reads/s (S3DP_BLOCKS_PER_CU=%s)" % (a.k, per, dt * 1e3, per * a.k / dt / 1e6,
          os.environ.get("S3DP_BLOCKS_PER_CU", "default")), flush=True)


if __name__ == "__main__":
    main()
