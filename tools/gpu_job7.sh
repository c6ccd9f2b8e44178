#!/bin/bash
# round-2 evidence run: full gpu suite, default bench line, its rocprof kernel summary, PMC traffic passes,
# K scan, phi-vs-work, the other configs.   gpu_job7.sh [tag]
TAG=${1:-r02f}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python bench.py > $O/bench_astroph_k20.json 2> $O/bench.err; tail -c 600 $O/bench_astroph_k20.json; echo
for wl in lfr-k28 astroph-k200; do python bench.py --workload $wl --no-hbm-bound --no-cpu-baseline > $O/bench_$wl.json 2>> $O/bench.err; done
timeout 900 python bench.py --workload mmsb:1000000:512:24 --steps 20 --warmup 3 --no-hbm-bound --no-cpu-baseline > $O/bench_mmsb_n1m_k512.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_bench -o p -- python $R/bench.py --no-cpu-baseline --no-hbm-bound > $O/prof_bench.log 2>&1
for wl in astroph-k20 synthetic:200000:512:24; do
  w=$(echo $wl | tr ':' '_')
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf_$w -o p -- python $R/tools/kernel_times.py $wl 15 > $O/pmcf_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw_$w -o p -- python $R/tools/kernel_times.py $wl 15 > $O/pmcw_$w.log 2>&1
done
cd $R
python tools/rocprof_summary.py $(ls $O/prof_bench/*results.db | head -1) "python bench.py --no-cpu-baseline --no-hbm-bound" > $O/bench_kernel_stats.txt; head -8 $O/bench_kernel_stats.txt
bash tools/k_scan.sh > $O/k_scan.txt 2>/dev/null; cat $O/k_scan.txt
timeout 600 python tools/phi_vs_work.py astroph-k20 1300 100 2>&1 | grep -v amdgpu.ids > $O/phi_vs_work_astroph_k20.txt; cat $O/phi_vs_work_astroph_k20.txt
