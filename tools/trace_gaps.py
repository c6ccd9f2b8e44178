#!/usr/bin/env python
"""Kernel durations AND the gaps between consecutive kernels of the sweep loop, from a rocprofv3 --kernel-trace CSV
(start / end timestamps per dispatch): where a graph-replayed sweep's wall time goes.

  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py ...
  python tools/trace_gaps.py DIR/**/t_kernel_trace.csv [first_dispatch_to_skip]
"""
import csv, glob, sys
from collections import defaultdict

f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    return r["Kernel_Name"].split("(")[0].replace("void svils::", "").replace("svils::", "")[:28]
dur = defaultdict(list)
gap = defaultdict(list)
prev = None
for r in rows:
    n = nm(r)
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append((e - s) / 1e3)
    if prev is not None:
        g = (s - prev[1]) / 1e3
        if g < 50:   # same burst
            gap[(prev[0], n)].append(g)
    prev = (n, e)
import statistics as st
print("%-30s %8s %10s %10s" % ("kernel", "n", "median_us", "mean_us"))
for n, v in sorted(dur.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 20:
        print("%-30s %8d %10.2f %10.2f" % (n, len(v), st.median(v), sum(v) / len(v)))
print()
print("%-60s %8s %10s" % ("gap: end of A -> start of B", "n", "median_us"))
for (a, b), v in sorted(gap.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 20:
        print("%-60s %8d %10.2f" % (a + " -> " + b, len(v), st.median(v)))
