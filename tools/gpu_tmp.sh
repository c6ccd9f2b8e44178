cd $GRAFT_REPO_ROOT
O=gpurun_out/r03zb; mkdir -p $O
for rep in 1 2 3; do
for v in 1 0; do
SVILS_GRAPH_POW2=$v python bench.py --steps 20 --warmup 5 --no-hbm-bound --no-config5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('pow2(64)=$v steps20 ', d['ms_per_step'], d['repeat']['min_ms_per_step'])" | tee -a $O/pow2.txt
SVILS_GRAPH_POW2=$v python bench.py --no-hbm-bound --no-config5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('pow2(64)=$v steps100', d['ms_per_step'], d['repeat']['min_ms_per_step'])" | tee -a $O/pow2.txt
SVILS_GRAPH_POW2=$v python bench.py --workload lfr-k28 --no-hbm-bound --no-config5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('pow2(64)=$v lfr-k28 ', d['ms_per_step'], d['repeat']['min_ms_per_step'])" | tee -a $O/pow2.txt
done
done
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
