cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
timeout 2700 python -m pytest tests -q -m gpu -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
for wl in astroph-k20 lfr-k28; do
python bench.py --no-hbm-bound --no-config5 --no-cpu-baseline --reps 30 --workload $wl 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$wl', d['ms_per_step'], d['repeat']['min_ms_per_step'], d['roofline']['avg_launch_us'])" | tee -a $O/bench_small.txt
done
