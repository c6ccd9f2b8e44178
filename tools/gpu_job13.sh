#!/bin/bash
# round 3: native multi-rank tests, the new bench line (repeated windows, config5 record), PMC traffic of three workloads
TAG=${1:-r03b}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_native_ranks.py -q -m gpu --timeout 900 > $O/pytest_native.log 2>&1; echo "pytest rc=$?" >> $O/pytest_native.log
tail -30 $O/pytest_native.log
python bench.py --steps 20 --warmup 5 > $O/bench_astroph_k20_steps20.json 2> $O/bench.err; tail -c 600 $O/bench_astroph_k20_steps20.json; echo
python bench.py --no-hbm-bound --no-config5 --no-cpu-baseline > $O/bench_astroph_k20.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
for wl in astroph-k20 synthetic:200000:512:24 mmsb:1000000:512:24; do
  w=$(echo $wl | tr ':' '_')
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf_$w -o p -- python $R/tools/kernel_times.py $wl 6 > $O/pmcf_$w.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw_$w -o p -- python $R/tools/kernel_times.py $wl 6 > $O/pmcw_$w.log 2>&1
  find $O/pmcf_$w $O/pmcw_$w -type f ! -name "*counter_collection.csv" -delete
done
du -sh $O
