#!/bin/bash
# round 3, evidence on the very last code: smoke, the driver's command and the default bench line (with config5 / hbm_bound / cpu baselines),
# LFR and K=200 lines, rocprofv3 kernel stats of the default command, the whole GPU suite:   gpu_job19.sh [tag]
TAG=${1:-r03zk}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_astroph_k20_steps20.json 2> $O/bench.err; tail -c 200 $O/bench_astroph_k20_steps20.json; echo
python bench.py --no-hbm-bound --no-config5 > $O/bench_astroph_k20.json 2>> $O/bench.err
python bench.py --workload lfr-k28 --no-hbm-bound --no-config5 --no-cpu-baseline > $O/bench_lfr_k28.json 2>> $O/bench.err
python bench.py --workload astroph-k200 --no-hbm-bound --no-config5 --no-cpu-baseline > $O/bench_astroph_k200.json 2>> $O/bench.err
python tools/large_small_k.py 1000000 20 2>&1 | grep -v "^W2\|^E2\|amdgpu" | tail -2 | tee $O/large_small_k.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $R/bench.py --no-cpu-baseline --no-hbm-bound --no-config5 > $O/prof_bench.log 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/astroph_k20_kernel_stats.csv && head -5 $f | cut -c1-160
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o k -- python $R/bench.py --workload lfr-k28 --no-cpu-baseline --no-hbm-bound --no-config5 > $O/prof_bench2.log 2>&1)
f=$(find $O/prof2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/lfr_k28_kernel_stats.csv && head -5 $f | cut -c1-160
rm -rf $O/prof $O/prof2
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
