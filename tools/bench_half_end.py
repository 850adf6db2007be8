#!/usr/bin/env python3
"""s3_dp_half_end on ~30 k half-aligned pairs of a small random genome: wall clock of the engine call against the time of its kernels (bench.py's
extras.default_dp_half_end without the 3.1 Gbp genome).  S3DP_VERBOSE=1 prints where the call's time goes.  (GPU only; measurement aid.)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "soap3-dp_amd", "py"))


def main():
    import torch
    import bench
    import s3dp
    from s3dp import builder
    dev = torch.device("cuda:0")
    codes, _ = bench.make_genome(200_000_000, 1, dev, 0.0)
    hi = builder.build_host_index(codes.cpu().numpy(), device=dev, verbose=False)
    ix = s3dp.Index(hi, device=0)
    L, n_dp = 150, 32768
    wpq = s3dp.word_per_query(L)
    reads2d = bench.make_pairs(codes, n_dp, L, 777, dev)
    q = bench.pack_on_device(reads2d, wpq)
    lens = np.full(2 * n_dp, L, dtype=np.uint32)
    bb = s3dp.Batch(ix, 2 * n_dp, wpq)
    bb.upload(q, lens)
    bb.align_pair(3, 8000, 2, 1, 500, 1000)
    pr = bb.fetch_pairs()[0]
    occ = np.zeros(n_dp, dtype=s3dp.OCC_DTYPE)
    k = 0
    for p in range(n_dp):
        if pr["phase"][p]:
            occ[k] = (2 * p, pr["bestA1"][p], pr["bestS1"][p], pr["bestM1"][p], (0, 0))
            k += 1
    dp = s3dp.DpAligner(n_dp, 152, 500 - 1 + 152 + 1, scheme=1)
    out = []
    for rep in range(4):
        t0 = time.perf_counter()
        resd, cig = dp.half_end(bb, lens, occ[:k], 1, 500, (45, 45), 49, 49)
        wall = (time.perf_counter() - t0) * 1e3
        out.append({"problems": k, "kernel_ms": dp.timing(), "engine_call_ms": getattr(dp, "last_call_ms", None), "wall_ms_incl_python_binding": wall})
    print(json.dumps(out[-1]))
    print(json.dumps(out), file=sys.stderr)


if __name__ == "__main__":
    main()
