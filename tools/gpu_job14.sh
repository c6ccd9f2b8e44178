#!/bin/bash
# round 3: new k_finalize_lpl (8 nodes per wavefront) -- parity subset, per-kernel times, K=28/32 phi occupancy variants
TAG=${1:-r03c}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_stochastic.py tests/test_gpu_properties.py tests/test_gpu_sharded.py -q -m gpu -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
for wl in astroph-k20 lfr-k28 astroph-k28 astroph-k32 astroph-k33 astroph-k8 astroph-k48; do
  python tools/kernel_times.py $wl 100 2>/dev/null | tee -a $O/kernel_times.txt
done
for lib in libsvils_mid42.so libsvils_mid84.so; do
  for wl in lfr-k28 astroph-k28 astroph-k32; do
    SVILS_LIB=$R/svinet_amd/lib/$lib python tools/kernel_times.py $wl 100 2>/dev/null | tee -a $O/kernel_times.txt
  done
done
python bench.py --no-hbm-bound --no-config5 --no-cpu-baseline > $O/bench_astroph_k20.json 2>> $O/bench.err; tail -c 300 $O/bench_astroph_k20.json
python bench.py --no-hbm-bound --no-config5 --no-cpu-baseline --workload lfr-k28 > $O/bench_lfr_k28.json 2>> $O/bench.err
