#!/bin/bash
# rocprofv3 kernel-trace summary of tools/kernel_times.py for a workload:  gpu_job2.sh [workload] [tag]
WL=${1:-astroph-k20}; TAG=${2:-r02a}
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_$WL
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_$WL -o p -- python $GRAFT_REPO_ROOT/tools/kernel_times.py $WL 200 > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_$WL.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(ls gpurun_out/$TAG/prof_$WL/*/*results.db gpurun_out/$TAG/prof_$WL/*results.db 2>/dev/null | head -1) "kernel_times $WL 200" | tee gpurun_out/$TAG/${WL}_kernel_stats.txt
