# bare `python bench.py --gpus N --test-one-gpu` for N = 4, 8: all ranks on GPU 0 over the tests' transport -- the N > 1 code path of bench.py
# (self-spawned ranks, gloo control plane, svils_comm_init, sharded drivers, one side record) at the world sizes the driver uses.  Never a measurement.
mkdir -p gpurun_out/r04u
python -c "import __graft_entry__ as g; g.build_test_transport()" >/dev/null 2>&1
for N in 4 8; do
  t0=$(date +%s)
  SVILS_RCCL_LIBRARY=$PWD/tests/fakerccl/libfakerccl.so FAKERCCL_ASYNC=1 timeout 700 python bench.py --gpus $N --steps 10 --warmup 2 --test-one-gpu --extra-list ksharded_config4_astroph_k200 --no-cpu-baseline > gpurun_out/r04u/bench_gpus${N}_one_gpu.json 2> gpurun_out/r04u/err$N.txt
  echo "N=$N rc=$? wall $(( $(date +%s) - t0 )) s"
  tail -2 gpurun_out/r04u/err$N.txt
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04u/bench_gpus${N}_one_gpu.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("n_gpus","value","ms_per_step","scaling","speedup_vs_n1_same_box","error")}, d.get("rccl",{}).get("nranks"), list((d.get("sharded_extra") or {}).items())[:1])
PY
done
