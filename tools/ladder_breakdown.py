#!/usr/bin/env python3
"""Per-kernel device time of ONE ladder of the bench, out of a rocprofv3 --kernel-trace CSV: the window between the last two launches of the first
pass's widest kernel call pattern (the ladder's first firstKernel<..., false, false>) -- python3 tools/ladder_breakdown.py <kernel_trace.csv>"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "firstKernel<true, false, false>" in r["Kernel_Name"]]
if len(first) < 8:
    sys.exit("fewer than two ladders in the trace")
# a ladder = 4 first passes; the last complete one
a, b = first[-8], first[-4]
t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
acc, calls = collections.Counter(), collections.Counter()
busy = 0
for r in rows[a:b]:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-70:]
    acc[name] += d; calls[name] += 1; busy += d
print("ladder window %.2f ms, kernels %.2f ms" % ((t1 - t0) / 1e6, busy / 1e6))
for name, d in acc.most_common(40):
    print("%8.3f ms  %4d  %s" % (d / 1e6, calls[name], name))
print("the search kernels' launches in order (ms):")
for r in rows[a:b]:
    n = r["Kernel_Name"]
    if "firstKernel" in n or "unitKernel" in n or "ceKernel" in n:
        print("  %8.3f  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, n.split("(")[0][-44:]))
