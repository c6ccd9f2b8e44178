#!/bin/bash
# A/B of libsvils builds on the graph-replayed sweep: tools/ab_libs.sh OUT lib1 lib2 ...   (names under svinet_amd/lib/)
out=$1; shift
mkdir -p "$(dirname "$out")"; : > "$out"
for rep in 1 2; do
for lib in "$@"; do
  for wl in astroph-k20 lfr-k28; do
    SVILS_LIB=$PWD/svinet_amd/lib/$lib timeout 300 python bench.py --workload $wl --steps 400 --warmup 5 --no-hbm-bound --no-cpu-baseline 2>/dev/null |
      python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $wl rep$rep  events-run %.2f us  graph-replay %.2f us  phi %.2f us' % (d['ms_per_step']*1e3, d['graph_replay']['ms_per_step']*1e3, d['roofline']['avg_launch_us']))" >> "$out"
  done
done
done
cat "$out"
