#!/bin/bash
# A/B of one environment knob of libsvils on one box, alternating:  tools/ab_env.sh OUT VAR "workload args" ["workload args" ...]
OUT=$1; VAR=$2; shift 2
for rep in 1 2 3; do
 for v in 0 1; do
  for wl in "$@"; do
    r=$(env $VAR=$v python bench.py --workload $wl --reps 20 --no-hbm-bound --no-config5 --no-cpu-baseline --no-cli 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f us  (p10 %.3f p90 %.3f)' % (d['ms_per_step']*1e3, d['repeat']['p10_ms_per_step']*1e3, d['repeat']['p90_ms_per_step']*1e3))")
    echo "rep$rep $VAR=$v $wl: $r"
  done
 done
done | tee $OUT
