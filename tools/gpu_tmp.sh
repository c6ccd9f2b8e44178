cd $GRAFT_REPO_ROOT
O=gpurun_out/r03y; mkdir -p $O
for rep in 1 2; do
for wl in astroph-k20 astroph-k200 lfr-k28; do
python bench.py --no-hbm-bound --no-config5 --no-cpu-baseline --reps 30 --workload $wl 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$wl', d['ms_per_step'], d['repeat']['min_ms_per_step'], d['roofline']['avg_launch_us'])" | tee -a $O/bench2.txt
done
done
python bench.py --steps 2000 --warmup 200 > $O/bench_driver.json 2>$O/bench_driver.err; tail -c 600 $O/bench_driver.json
