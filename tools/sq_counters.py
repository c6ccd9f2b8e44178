#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes of SQ counters (a few counters per pass, --kernel-trace only, csv output; the
passes are directories under one parent) into one table: averages per launch and kernel.

  python tools/sq_counters.py <parent dir with one sub-directory per pass> "<header text>"
"""
import csv, glob, os, sys
from collections import defaultdict, OrderedDict

parent = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
counters = []
for f in sorted(glob.glob(os.path.join(parent, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void svils::", "").replace("svils::", "")
        c = r["Counter_Name"]
        if c not in counters:
            counters.append(c)
        a = acc[name][c]
        a[0] += float(r["Counter_Value"]); a[1] += 1
if len(sys.argv) > 2:
    print("# " + sys.argv[2])
print("# SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are summed over waves or SIMDs in quad-cycles (MI355X_MICROARCH.md)")
print("%-34s" % "kernel" + "".join("%22s" % c for c in counters))
for k, d in acc.items():
    if k.startswith("__amd"):
        continue
    print("%-34s" % k[:34] + "".join("%22.0f" % (d[c][0] / max(d[c][1], 1)) if c in d else "%22s" % "-" for c in counters))
