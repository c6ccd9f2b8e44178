#!/bin/bash
# edge-updates/s and phi-kernel figures over K on ca-AstroPh (tools/k_scan.sh > profiles/<name>.txt)
cd "$(dirname "$0")/.."
printf "%-6s %12s %14s %12s %10s\n" K ms_per_sweep edge_updates_s phi_us alg_TBs
for k in 4 8 16 20 24 28 32 33 40 48 56 64 65 100 128 200 256 400 512 513 640 768 769 1024; do
  python bench.py --no-cpu-baseline --no-hbm-bound --workload astroph-k$k --steps 60 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-6d %12.4f %14.4g %12.1f %10.2f' % ($k, d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['roofline']['achieved_survey_model'] / 1000))"
done
