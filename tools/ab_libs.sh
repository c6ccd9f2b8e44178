#!/bin/bash
# A/B of libsvils builds (tools/build_variant.sh) on one box, alternating:  tools/ab_libs.sh OUT lib1 lib2 ...  (names under svinet_amd/lib/)
# per lib x workload x repetition: the bench line's median sweep time over 30 re-seeded windows, and the per-kernel hipEvent times
out=$1; shift
WLS=${WLS:-"astroph-k20 lfr-k28"}
mkdir -p "$(dirname "$out")"; : > "$out"
for rep in 1 2 3; do
for lib in "$@"; do
  for wl in $WLS; do
    line=$(SVILS_LIB=$PWD/svinet_amd/lib/$lib timeout 300 python bench.py --workload $wl --steps 100 --warmup 5 --reps 30 --no-hbm-bound --no-config5 --no-cpu-baseline --no-cli 2>/dev/null | tail -1 |
      python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('sweep median %.2f us (min %.2f)  phi %.2f us' % (d['ms_per_step']*1e3, d['repeat']['min_ms_per_step']*1e3, d['roofline']['avg_launch_us']))")
    kt=$(SVILS_LIB=$PWD/svinet_amd/lib/$lib timeout 300 python tools/kernel_times.py $wl 100 2>/dev/null | tail -1 | cut -d' ' -f3-)
    echo "$lib $wl rep$rep  $line  | eager per-kernel: $kt" >> "$out"
  done
done
done
cat "$out"
