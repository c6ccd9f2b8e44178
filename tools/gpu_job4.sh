#!/bin/bash
# round-2 measurement job: default bench line, its rocprof kernel summary, PMC traffic passes
TAG=${1:-r02b}
mkdir -p gpurun_out/$TAG
python bench.py > gpurun_out/$TAG/bench_astroph_k20.json 2> gpurun_out/$TAG/bench.err
tail -c 3000 gpurun_out/$TAG/bench_astroph_k20.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof_bench -o p -- python $R/bench.py --no-cpu-baseline --no-hbm-bound > $R/gpurun_out/$TAG/prof_bench.log 2>&1
for wl in astroph-k20 synthetic:200000:512:24; do
  w=$(echo $wl | tr ':' '_')
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/pmcf_$w -o p -- python $R/tools/kernel_times.py $wl 15 > $R/gpurun_out/$TAG/pmcf_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/pmcw_$w -o p -- python $R/tools/kernel_times.py $wl 15 > $R/gpurun_out/$TAG/pmcw_$w.log 2>&1
done
cd $R
python tools/rocprof_summary.py $(ls gpurun_out/$TAG/prof_bench/*results.db | head -1) "python bench.py --no-cpu-baseline --no-hbm-bound" | tee gpurun_out/$TAG/bench_kernel_stats.txt | head -12
ls gpurun_out/$TAG
