#!/bin/bash
# final check of a round: build() is up to date, smoke(), full gpu suite, default bench, config-4 bench, K scan
TAG=${1:-r02g}
O=gpurun_out/$TAG; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
timeout 1500 python -m pytest tests -q -m gpu -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python bench.py > $O/bench_astroph_k20.json 2> $O/bench.err; tail -c 400 $O/bench_astroph_k20.json; echo
python bench.py --workload astroph-k200 --no-hbm-bound --no-cpu-baseline > $O/bench_astroph-k200.json 2>> $O/bench.err
bash tools/k_scan.sh > $O/k_scan.txt 2>/dev/null; cat $O/k_scan.txt
