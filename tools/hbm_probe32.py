#!/usr/bin/env python3
"""Does a random 32-byte access cost less than a random 64-byte line?  (rank lines could be 32 B: counts + 64 characters)"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "soap3-dp_amd", "py"))
import s3dp

print("table_GB ilp mode            ms   Gaccess/s")
for table in (6e9,):
    for ilp in (2, 4, 8):
        for lpl, name in ((1, "64B line, 16B load"), (2, "64B line, 2x16B   "), (11, "32B half, 16B load"), (12, "32B half, 2x16B   ")):
            ms, lines = s3dp.probe_gather(int(table), blocks=4096, iters=128, ilp=ilp, loads_per_line=lpl)
            print("%7.1f %3d %s %8.3f %8.2f" % (table / 1e9, ilp, name, ms, lines / ms / 1e6))
