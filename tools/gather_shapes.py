#!/usr/bin/env python3
"""Random 64-byte-line gather rate by launch shape: a few long-running workgroups (the persistent form of the search kernels) against many short
ones (one to eight lines per lane, the grid covers the work).  Run on the GPU box:  python tools/gather_shapes.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "soap3-dp_amd", "py"))
import s3dp

total = 1 << 29
print("table_GB  blocks  iters ilp loads/line dependent      ms  Glines/s")
for table in (6e9, 50e9):
    for iters, ilp in ((128, 4), (128, 8), (32, 4), (8, 4), (2, 4), (1, 4), (1, 8), (1, 2), (1, 1)):
        blocks = max(256, total // (256 * iters * ilp))
        for dep in (0, 1):
            if dep and iters == 1:
                continue
            ms, lines = s3dp.probe_gather(int(table), blocks=blocks, iters=iters, ilp=ilp, loads_per_line=1, dependent=dep)
            print("%7.1f %8d %5d %3d %5d %9d %9.3f %8.2f" % (table / 1e9, blocks, iters, ilp, 1, dep, ms, lines / ms / 1e6))

print("\ngranule  table_GB   ms  G accesses/s  GB/s fetched if whole granules move")
for table in (6e9, 50e9):
    for lpl, name, size in ((10, "32 B", 32), (1, "64 B", 64), (20, "128 B", 128), (21, "256 B", 256)):
        ms, lines = s3dp.probe_gather(int(table), blocks=4096, iters=128, ilp=4, loads_per_line=lpl)
        print("%7s %8.1f %8.3f %8.2f %10.0f" % (name, table / 1e9, ms, lines / ms / 1e6, lines / ms / 1e6 * size))
