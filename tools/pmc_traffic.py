#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, csv output) into HBM bytes
per kernel launch and per whole sweep, and write the record bench.py reads (profiles/traffic.json).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR/pmcf_W -o p -- python tools/kernel_times.py WORKLOAD 6
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d DIR/pmcw_W -o p -- python tools/kernel_times.py WORKLOAD 6
  python tools/pmc_traffic.py OUT.json TABLE.txt WORKLOAD DIR/pmcf_W DIR/pmcw_W [WORKLOAD DIR DIR ...]
(tools/evidence.sh pmc does all of it on the GPU box; copy OUT.json to profiles/traffic.json and TABLE.txt beside it.)

Counters are in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM
section), so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  Every record carries the hashes of the kernel's source
files as they were in the tree that was measured (bench.KERNEL_SOURCES): bench.py refuses the record once they differ.
"""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


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


def workload_k(wl):
    if wl.startswith("astroph-k"):
        return int(wl[len("astroph-k"):])
    if wl.startswith("lfr-k"):
        return int(wl[len("lfr-k"):])
    return int(wl.split(":")[2])


def main():
    from bench import kernel_source_hashes
    out_json, table = sys.argv[1:3]
    rest = sys.argv[3:]
    rec = {}
    lines = []
    commit = os.environ.get("EVIDENCE_COMMIT")   # the GPU box has no .git: the caller names the commit it snapshotted
    for i in range(0, len(rest), 3):
        wl, fd, wd = rest[i:i + 3]
        fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
        lines.append("%-24s %-34s %14s %14s %18s %6s" % ("workload", "kernel", "FETCH_SIZE_KB", "WRITE_SIZE_KB", "hbm_bytes/launch", "n"))
        phi, nphi, per_kernel = None, 0, {}
        for k in fe:
            n = max(fe[k][1], 1)
            f, w = fe[k][0] / n, wr.get(k, [0, 1])[0] / max(wr.get(k, [0, 1])[1], 1)
            b = (2 * f + w) * 1024
            lines.append("%-24s %-34s %14.0f %14.0f %18.0f %6d" % (wl, k[:34], f, w, b, n))
            per_kernel[k] = (b, n)
            if k.startswith("k_phi") and (phi is None or b > phi):   # (handles that store no Elogpi launch a second, normally empty, phi kernel)
                phi, nphi = b, n
        if phi is None:
            continue
        # one whole sweep: every kernel the loop launches (launched at least every other sweep), bytes x launches / sweeps
        in_loop = {k: v for k, v in per_kernel.items() if k.startswith("k_") and v[1] * 2 >= nphi}
        sweep = sum(b * n for b, n in in_loop.values()) / nphi
        lines.append("%-24s %-34s %48.0f %6d" % (wl, ("whole sweep: " + "+".join(sorted(x.split("<")[0] for x in in_loop)))[:34], sweep, nphi))
        rec[wl] = {"phi_hbm_bytes_per_launch": phi, "sweep_hbm_bytes": sweep,
                   "sweep_kernels": {k: {"hbm_bytes_per_launch": v[0], "launches": v[1]} for k, v in sorted(in_loop.items())},
                   # where tools/evidence.sh's table is kept once copied: profiles/<tag>_hbm_traffic_pmc.txt (TRAFFIC_SOURCE overrides)
                   "source": os.environ.get("TRAFFIC_SOURCE") or "profiles/%s_%s" % (os.path.basename(os.path.dirname(os.path.abspath(table))), os.path.basename(table)),
                   "commit": commit, "source_hashes": kernel_source_hashes(workload_k(wl)),
                   "counters": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) KB"}
    open(table, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    json.dump(rec, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
