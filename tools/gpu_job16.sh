#!/bin/bash
# SQ counter passes (three counters each, kernel-trace only) on the round-3 kernels: ca-AstroPh K=20 and config 5:  gpu_job16.sh [tag]
TAG=${1:-r03v}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for wl in astroph-k20 mmsb:1000000:512:24; do
  w=$(echo $wl | tr ':' '_')
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/sq_$w/pass$i -o p -- python $R/tools/kernel_times.py $wl 10 > $R/gpurun_out/$TAG/sq_${w}_pass$i.log 2>&1
  done
  cd $R; python tools/sq_counters.py gpurun_out/$TAG/sq_$w "rocprofv3 --pmc (5 passes of 3 SQ counters, --kernel-trace only), averages per launch over the dispatches of tools/kernel_times.py $wl 10" > gpurun_out/$TAG/sq_counters_$w.txt; cut -c1-250 gpurun_out/$TAG/sq_counters_$w.txt; cd /tmp
done
