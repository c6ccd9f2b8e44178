#!/usr/bin/env python
"""Roofline records of the kernels a SHARDED run executes on one rank: rocprofv3 kernel stats (average launch duration)
joined with the --pmc FETCH_SIZE / WRITE_SIZE passes of the same command (tools/shard_rank.py), per kernel:
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB (the gfx950 correction of MI355X_MICROARCH.md, as tools/pmc_traffic.py),
achieved = bytes / duration, frac = achieved / 8 TB/s.

  python tools/pmc_sharded.py OUT.json TABLE.txt LABEL STATS.csv PMCF_DIR PMCW_DIR [LABEL STATS.csv PMCF_DIR PMCW_DIR ...]
"""
import csv, glob, json, os, sys
from collections import defaultdict

PEAK = 8.0e12


def short(name):
    return name.split("(")[0].replace("void svils::", "").replace("svils::", "")


def load_pmc(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = short(r["Kernel_Name"])
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    return acc


def load_stats(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[short(r["Name"])] = (float(r["AverageNs"]), int(r["Calls"]))
    return out


def main():
    out_json, table = sys.argv[1:3]
    rest = sys.argv[3:]
    rec, lines = {}, []
    for i in range(0, len(rest), 4):
        label, stats, fd, wd = rest[i:i + 4]
        st, fe, wr = load_stats(stats), load_pmc(fd, "FETCH_SIZE"), load_pmc(wd, "WRITE_SIZE")
        lines.append("# %s" % label)
        lines.append("%-44s %8s %12s %16s %10s %7s" % ("kernel", "calls", "avg us", "HBM bytes/launch", "TB/s", "frac"))
        ks = {}
        # the sweep's own kernels: launched at least once per sweep (set-up launches of one or two calls are left out); sweeps run =
        # launches of the kernel the run spent most time in (the phi pass: one launch per sweep)
        nmax = max(((ns * c, c) for k, (ns, c) in st.items() if k.startswith("k_")), default=(0, 1))[1]
        tot_b = tot_t = 0.0
        for k, (ns, calls) in sorted(st.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            if not k.startswith("k_") or k not in fe or calls < nmax:
                continue
            f = fe[k][0] / max(fe[k][1], 1)
            w = wr.get(k, [0.0, 1])[0] / max(wr.get(k, [0.0, 1])[1], 1)
            b = (2 * f + w) * 1024
            ach = b / (ns * 1e-9)
            per_sweep = calls / float(nmax)
            lines.append("%-44s %8d %12.1f %16.0f %10.3f %7.3f" % (k[:44], calls, ns / 1e3, b, ach / 1e12, ach / PEAK))
            ks[k] = {"avg_launch_us": ns / 1e3, "calls": calls, "hbm_bytes_per_launch": b, "achieved_TBps": ach / 1e12, "frac": ach / PEAK,
                     "launches_per_sweep": per_sweep}
            tot_b += b * per_sweep
            tot_t += ns * 1e-9 * per_sweep
        lines.append("%-44s %8s %12.1f %16.0f %10.3f %7.3f" % ("whole rank-sweep (kernels above)", "", tot_t * 1e6, tot_b, tot_b / tot_t / 1e12, tot_b / tot_t / PEAK))
        lines.append("")
        rec[label] = {"kernels": ks, "sweep_hbm_bytes": tot_b, "sweep_kernel_time_ms": tot_t * 1e3, "sweep_frac": tot_b / tot_t / PEAK,
                      "commit": os.environ.get("EVIDENCE_COMMIT"),
                      "counters": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) KB; "
                                  "durations: rocprofv3 --kernel-trace --stats of the same command (tools/shard_rank.py)"}
    open(table, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    json.dump(rec, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
