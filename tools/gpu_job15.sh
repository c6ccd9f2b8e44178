#!/bin/bash
# round 3 evidence on the current code: smoke, driver-style bench (--steps 20 --warmup 5) and the default bench, rocprofv3 kernel
# stats of the same command, PMC traffic of three workloads, the N>1 code path with a world of one:  gpu_job15.sh [tag]
TAG=${1:-r03m}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_astroph_k20_steps20.json 2> $O/bench.err; tail -c 300 $O/bench_astroph_k20_steps20.json; echo
python bench.py --no-hbm-bound --no-config5 > $O/bench_astroph_k20.json 2>> $O/bench.err
python bench.py --workload lfr-k28 --no-hbm-bound --no-config5 --no-cpu-baseline > $O/bench_lfr_k28.json 2>> $O/bench.err
python bench.py --workload astroph-k200 --no-hbm-bound --no-config5 --no-cpu-baseline > $O/bench_astroph_k200.json 2>> $O/bench.err
python bench.py --force-sharded --steps 100 --extra-list config4_astroph_k200,hbm_bound_n200k_k512,ksharded_config4_astroph_k200,ksharded_hbm_bound_n200k_k512 > $O/bench_force_sharded_world1.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $R/bench.py --no-cpu-baseline --no-hbm-bound --no-config5 > $O/prof_bench.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/astroph_k20_kernel_stats.csv && head -8 $f | cut -c1-200
rm -rf $O/prof
for wl in astroph-k20 synthetic:200000:512:24 mmsb:1000000:512:24; do
  w=$(echo $wl | tr ':' '_')
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf_$w -o p -- python $R/tools/kernel_times.py $wl 6 > $O/pmcf_$w.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw_$w -o p -- python $R/tools/kernel_times.py $wl 6 > $O/pmcw_$w.log 2>&1
  find $O/pmcf_$w $O/pmcw_$w -type f ! -name "*counter_collection.csv" -delete
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof5 -o k -- python $R/tools/kernel_times.py mmsb:1000000:512:24 6 > $O/prof5.log 2>&1
f=$(find $O/prof5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/mmsb_n1m_k512_kernel_stats.csv
rm -rf $O/prof5
du -sh $O
