cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_astroph_k20_steps20.json 2> $O/bench2.err
python bench.py --no-hbm-bound --no-config5 > $O/bench_astroph_k20.json 2>> $O/bench2.err
python - <<'PY'
import json
for f in ('bench_astroph_k20_steps20','bench_astroph_k20'):
    d=json.loads(open('gpurun_out/r03z/%s.json'%f).read().strip().split('\n')[-1])
    print(f,d['ms_per_step'],d['cpu_baseline']['value'],d['cpu_baseline_allcores'])
PY
