#!/usr/bin/env python
"""Where the wave cycles of each kernel go: one rocprofv3 --pmc pass of the SQ counters (MI355X_MICROARCH.md, "rocprofv3 PMC
slots": WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, in quad-cycles) summarised per kernel.

  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES \\
            --kernel-trace --output-format csv -d DIR -o p -- python tools/kernel_times.py WORKLOAD 6
  python tools/pmc_sq.py TABLE.txt LABEL DIR [LABEL DIR ...]
"""
import csv, glob, os, sys
from collections import defaultdict

NAMES = ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES")


def main():
    table = sys.argv[1]
    rest = sys.argv[2:]
    lines = []
    for i in range(0, len(rest), 2):
        label, d = rest[i:i + 2]
        fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not fs:
            lines.append("# %s: no counter file (the pass failed)" % label)
            continue
        acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(fs[0])):
            k = r["Kernel_Name"].split("(")[0].replace("void svils::", "").replace("svils::", "")
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        lines.append("# %s   (per launch; SQ counters in quad-cycles summed over all waves / SEs)" % label)
        lines.append("%-40s %6s %14s %9s %9s %9s %9s %12s" % ("kernel", "n", "WAVE_CYCLES", "parked", "issue-st", "issuing", "of it VALU", "INSTS_VALU"))
        for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [0, 1])[0]):
            if not k.startswith("k_"):
                continue
            n = max(c["SQ_WAVE_CYCLES"][1], 1)
            v = {m: c[m][0] / max(c[m][1], 1) for m in NAMES}
            wc = max(v["SQ_WAVE_CYCLES"], 1.0)
            lines.append("%-40s %6d %14.0f %9.3f %9.3f %9.3f %9.3f %12.0f" % (k[:40], n, v["SQ_WAVE_CYCLES"], v["SQ_WAIT_ANY"] / wc, v["SQ_WAIT_INST_ANY"] / wc,
                                                                           v["SQ_ACTIVE_INST_ANY"] / wc, v["SQ_ACTIVE_INST_VALU"] / max(v["SQ_ACTIVE_INST_ANY"], 1.0), v["SQ_INSTS_VALU"]))
        lines.append("")
    open(table, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
