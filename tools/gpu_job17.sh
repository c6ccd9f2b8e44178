#!/bin/bash
# round 3, final evidence on the current code: gpu_job15.sh (smoke, bench lines, rocprof kernel stats, PMC traffic, world-of-one
# N>1 path) + the whole GPU suite + the per-rank cost-model inputs with per-kernel durations:  gpu_job17.sh [tag]
TAG=${1:-r03z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
bash $R/tools/gpu_job15.sh $TAG
cd $R
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
cd /tmp; export TMPDIR=/tmp
for wl in astroph-k200 mmsb:1000000:512:24; do
  t=$(echo $wl | tr ':' '_')
  python $R/tools/shard_cost.py $wl 2,4,8 2>/dev/null | tee -a $O/shard_cost_model_inputs.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$t -o p -- python $R/tools/shard_cost.py $wl 8 > /dev/null 2>&1
  f=$(find $O/prof_$t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/shard_rank0of8_kernel_stats_$t.csv
  rm -rf $O/prof_$t
done
du -sh $O
