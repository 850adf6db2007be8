#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per dispatch for kernels whose name contains a pattern (default: firstKernel)."""
import collections
import csv
import sys

path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "firstKernel")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
order = []
for r in csv.DictReader(open(path)):
    if pat not in r["Kernel_Name"]:
        continue
    d = r["Dispatch_Id"]
    if d not in order:
        order.append(d)
    agg[d][r["Counter_Name"]] += float(r["Counter_Value"])
for d in order:
    print("dispatch", d, " ".join("%s=%.4g" % kv for kv in sorted(agg[d].items())))
