#!/usr/bin/env python3
"""HIP API calls that took long, from a rocprofv3 --hip-trace csv next to a --kernel-trace csv: calls above a threshold inside the window that starts with the
first kernel of a given name, in time order (offset, duration, call, thread).
    tools/trace_api_stalls.py <hip_api_trace.csv> <kernel_trace.csv> <part of a kernel name> [threshold ms]"""
import csv
import sys

api, ker, name = sys.argv[1], sys.argv[2], sys.argv[3]
thr = float(sys.argv[4]) if len(sys.argv) > 4 else 5.0
kernels = list(csv.DictReader(open(ker)))
lo = min(int(r["Start_Timestamp"]) for r in kernels if name in r["Kernel_Name"])
by_corr = {r.get("Correlation_Id"): r["Kernel_Name"].split("(")[0][-70:] for r in kernels}
rows = []
for r in csv.DictReader(open(api)):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > lo and (e - s) / 1e6 >= thr:
        rows.append((s, e, r["Function"], r.get("Thread_Id", "?"), by_corr.get(r.get("Correlation_Id"), "")))
rows.sort()
print("HIP API calls of %.1f ms and more after the first %s (offset, duration, call, thread):" % (thr, name))
for s, e, f, t, k in rows[:80]:
    print("  at %8.1f ms  %8.2f ms  %-28s %s  %s" % ((s - lo) / 1e6, (e - s) / 1e6, f, t, k))
