#!/bin/bash
# round 3, first call: the native multi-rank tests (tests/fakerccl transport) + a baseline default bench:  gpu_job12.sh [tag]
TAG=${1:-r03a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_native_ranks.py -q -m gpu --timeout 900 > $O/pytest_native.log 2>&1; echo "pytest rc=$?" >> $O/pytest_native.log
tail -40 $O/pytest_native.log
python bench.py > $O/bench_astroph_k20.json 2> $O/bench.err; tail -c 400 $O/bench_astroph_k20.json; echo
