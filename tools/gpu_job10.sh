#!/bin/bash
# end-of-round evidence on the final code: smoke, full gpu suite, default bench, kernel stats of the same command,
# config-4 shape, force-sharded world-1 bench:  gpu_job10.sh [tag]
TAG=${1:-r02o}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
timeout 1500 python -m pytest tests -q -m gpu -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=" $O/pytest.log
python bench.py > $O/bench_astroph_k20.json 2> $O/bench.err; tail -c 300 $O/bench_astroph_k20.json; echo
python bench.py --workload astroph-k200 --no-hbm-bound --no-cpu-baseline > $O/bench_astroph-k200.json 2>> $O/bench.err
python bench.py --force-sharded --steps 100 > $O/bench_force_sharded_world1.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python $R/bench.py --no-cpu-baseline --no-hbm-bound > $O/prof_bench.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/astroph_k20_kernel_stats.csv && head -8 $f | cut -c1-200
rm -rf $O/prof
