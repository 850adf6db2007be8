#!/usr/bin/env python3
"""Device occupancy from a rocprofv3 --kernel-trace csv: how long the device ran at least one kernel, how many kernels ran at once (time-weighted), the
idle gaps (none running) with the kernels before and after the longest ones, and per kernel name its summed and exclusive share.
    tools/trace_gaps.py <kernel_trace.csv> [t0_fraction t1_fraction]
    tools/trace_gaps.py <kernel_trace.csv> from <part of a kernel name>      (the window starts at the first kernel of that name)"""
import csv
import sys
import collections

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
rows.sort()
lo, hi = rows[0][0], max(r[1] for r in rows)
if len(sys.argv) > 3 and sys.argv[2] == "from":
    lo = min(r[0] for r in rows if sys.argv[3] in r[2])
    rows = [r for r in rows if r[1] > lo]
elif len(sys.argv) > 3:
    span = hi - lo
    lo, hi = lo + int(float(sys.argv[2]) * span), lo + int(float(sys.argv[3]) * span)
    rows = [r for r in rows if r[1] > lo and r[0] < hi]
ev = []
for s, e, n in rows:
    ev.append((max(s, lo), 1, n)); ev.append((min(e, hi), -1, n))
ev.sort()
active = 0
last = lo
busy = 0
conc = collections.Counter()
gaps = []
lastEnd = None
for t, d, n in ev:
    if t > last:
        conc[active] += t - last
        if active > 0:
            busy += t - last
        elif lastEnd is not None:
            gaps.append((t - last, last, lastEnd))
    last = t
    if d < 0 and active == 1:
        lastEnd = n
    active += d
total = hi - lo
print("window %.3f s, device busy %.3f s (%.1f %%), kernels %d" % (total / 1e9, busy / 1e9, 100.0 * busy / total, len(rows)))
print("concurrency (share of the window): " + ", ".join("%d: %.1f %%" % (k, 100.0 * v / total) for k, v in sorted(conc.items())))
gaps.sort(reverse=True)
print("idle gaps: %d, summed %.3f s; > 100 us: %d summing %.3f s; > 1 ms: %d summing %.3f s" % (
    len(gaps), sum(g[0] for g in gaps) / 1e9, sum(1 for g in gaps if g[0] > 1e5), sum(g[0] for g in gaps if g[0] > 1e5) / 1e9,
    sum(1 for g in gaps if g[0] > 1e6), sum(g[0] for g in gaps if g[0] > 1e6) / 1e9))
print("the gaps above 1 ms, in time order (offset in the window, length, the kernel that ended before):")
for g in sorted((g for g in gaps if g[0] > 1e6), key=lambda g: g[1]):
    print("  at %8.1f ms  %7.2f ms  %s" % ((g[1] - lo) / 1e6, g[0] / 1e6, g[2]))
after = collections.Counter()
for g in gaps:
    after[g[2]] += g[0]
print("idle time by the kernel that ended before the gap:")
for n, v in after.most_common(12):
    print("  %8.1f ms  %s" % (v / 1e6, n))
per = collections.Counter()
for s, e, n in rows:
    per[n] += min(e, hi) - max(s, lo)
print("summed kernel time by name:")
for n, v in per.most_common(14):
    print("  %8.1f ms  %s" % (v / 1e6, n))
