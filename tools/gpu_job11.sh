#!/bin/bash
# PMC traffic passes (FETCH_SIZE and WRITE_SIZE in separate runs, kernel-trace only) on the current code:  gpu_job11.sh [tag]
TAG=${1:-r02o}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in astroph-k20 synthetic:200000:512:24; do
  w=$(echo $wl | tr ':' '_')
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf_$w -o p -- python $R/tools/kernel_times.py $wl 15 > $O/pmcf_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw_$w -o p -- python $R/tools/kernel_times.py $wl 15 > $O/pmcw_$w.log 2>&1
  find $O/pmcf_$w $O/pmcw_$w -type f ! -name "*counter_collection.csv" -delete   # keep only what tools/pmc_traffic.py reads
done
du -sh $O
# then, in the repository (git knows the commit):  python tools/pmc_traffic.py WORKLOAD gpurun_out/TAG/pmcf_W gpurun_out/TAG/pmcw_W profiles/TAG_hbm_traffic_pmc.txt
