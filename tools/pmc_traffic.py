#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, csv output)
into HBM bytes per kernel launch, and update profiles/traffic.json for the phi kernel.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_X -o p -- python tools/kernel_times.py WORKLOAD 15
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_X -o p -- python tools/kernel_times.py WORKLOAD 15
  python tools/pmc_traffic.py WORKLOAD gpurun_out/pmc_fetch_X gpurun_out/pmc_write_X [profiles/<file this table is kept in>]

Counters are in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv, glob, json, os, sys
from collections import defaultdict


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void svils::", "").replace("svils::", "")
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return acc


def main():
    wl, fd, wd = sys.argv[1:4]
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    print("%-14s %-34s %14s %14s %18s %6s" % ("workload", "kernel", "FETCH_SIZE_KB", "WRITE_SIZE_KB", "hbm_bytes/launch", "n"))
    phi = None
    for k in fe:
        n = max(fe[k][1], 1)
        f, w = fe[k][0] / n, wr.get(k, [0, 1])[0] / max(wr.get(k, [0, 1])[1], 1)
        b = (2 * f + w) * 1024
        print("%-14s %-34s %14.0f %14.0f %18.0f %6d" % (wl, k[:34], f, w, b, n))
        if k.startswith("k_phi"):
            phi = b
    if phi is not None:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(root, "profiles", "traffic.json")
        t = json.load(open(path)) if os.path.exists(path) else {}
        import subprocess
        try:
            commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
            dirty = bool(subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "svinet_amd/csrc"], text=True).strip())
            commit += "+uncommitted kernel changes" if dirty else ""
        except Exception:
            commit = None
        t[wl] = {"phi_hbm_bytes_per_launch": phi,
                 "source": sys.argv[4] if len(sys.argv) > 4 else None,   # the profiles/ file holding this table
                 "commit": commit,
                 "counters": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) KB"}
        json.dump(t, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
