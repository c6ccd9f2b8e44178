#!/bin/bash
# round 3, evidence on the code of the last session (one-byte cflag, finalise prefetch, tail / scatter blocks, pipelined K > 56 s3):
# gpu_job17.sh (smoke, bench lines, rocprof kernel stats, PMC traffic, world-of-one N>1 path, whole GPU suite, cost-model inputs)
# + the large-graph small-K sweep and the K = 100 / 200 per-kernel times:   gpu_job18.sh [tag]
TAG=${1:-r03zd}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
bash $R/tools/gpu_job17.sh $TAG
cd $R
python tools/large_small_k.py 1000000 20 2>&1 | grep -v "^W2\|^E2\|amdgpu" | tail -2 | tee $O/large_small_k.txt
for wl in astroph-k100 astroph-k200 astroph-k512; do python tools/kernel_times.py $wl 100 2>&1 | tail -1 | tee -a $O/kernel_times_large_k.txt; done
