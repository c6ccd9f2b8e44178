#!/bin/bash
# L2 (TCC) hit rate per kernel of the sweep: rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum (one pass, counters only) over
# tools/kernel_times.py <workload> 6.   tools/l2_hit_rate.sh <out.txt> <workload> [<workload> ...]
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
: > $OUT
for wl in "$@"; do
  w=$(echo $wl | tr ':' '_'); D=$(dirname $OUT)/tcc_$w
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $D -o p -- python $R/tools/kernel_times.py $wl 6 > $D.log 2>&1)
  python - "$wl" $D >> $OUT <<'PY'
import csv, glob, sys
from collections import defaultdict
wl, d = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void svils::", "").replace("svils::", "")
        v = float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_HIT_sum": acc[k][0] += v; acc[k][2] += 1
        elif r["Counter_Name"] == "TCC_MISS_sum": acc[k][1] += v
print("%-26s %-34s %14s %14s %8s %6s" % ("workload", "kernel", "TCC_HIT/launch", "TCC_MISS/launch", "hit rate", "n"))
for k, (h, m, n) in sorted(acc.items(), key=lambda kv: -kv[1][0] - kv[1][1]):
    if n and k.startswith("k_"):
        print("%-26s %-34s %14.0f %14.0f %8.3f %6d" % (wl, k[:34], h / n, m / n, h / max(h + m, 1), n))
PY
  rm -rf $D $D.log
done
cat $OUT
