#!/usr/bin/env python3
"""bench.py's C3 leg alone (2 x 150 bp pairs, DP stages on, through the command line) on the benchmark genome, with the command line's own report
printed: measurement aid for the host side of the paired-end path.  (GPU only.)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "soap3-dp_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-bp", type=int, default=3_100_000_000)
    ap.add_argument("--pairs", type=int, default=1_000_000)
    ap.add_argument("--repeat-frac", type=float, default=0.5)
    ap.add_argument("--index-dir", default=None)
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--no-cli", type=int, default=0, help="1: write the reads and the index files and stop (the command line is then run by the caller, e.g. under a profiler as the FIRST run on the files)")
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
    ix = s3dp.Index(hi, device=0)
    # (the command line runs after this process's own copy of the index is closed: two copies of a 219 GB index do not fit one device)
    res, cli = bench.c3_leg(s3dp, ix, hi, codes, dev, a.pairs, workdir=a.workdir, echo=True, defer_cli=True)
    ix.close()
    del codes
    torch.cuda.empty_cache()
    if a.no_cli:
        print(json.dumps({"prepared": a.workdir}))
        return
    res = cli()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
