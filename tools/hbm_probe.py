#!/usr/bin/env python3
"""Random 64-byte-line gather rate of the GPU (practical ceiling of the BWT rank-query stream).
Writes a small table to stdout; run on the GPU box:  python tools/hbm_probe.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "soap3-dp_amd", "py"))
import s3dp

print("table_GB blocks ilp loads/line dependent   ms   Glines/s  GB/s(64B lines)  GB/s(bytes loaded)")
for table in (1.5e9, 6e9):
    for blocks in (256 * 8, 256 * 16):
        for ilp in (1, 2, 4, 8):
            for lpl in (1, 2):
                ms, lines = s3dp.probe_gather(int(table), blocks=blocks, iters=128, ilp=ilp, loads_per_line=lpl)
                rate = lines / ms / 1e6
                print("%7.1f %6d %3d %5d %9d %7.3f %8.2f %10.1f %12.1f" % (table / 1e9, blocks, ilp, lpl, 0, ms, rate, rate * 64, rate * 16 * lpl))
