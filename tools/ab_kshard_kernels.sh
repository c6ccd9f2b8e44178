#!/bin/bash
# A/B of two builds on the K-sharded rank-0-of-8 runs (config 5, ca-AstroPh K=200): rocprofv3 kernel stats of tools/shard_rank.py under
# libsvils_prev.so (tools/build_variant.sh prev on the old sources) and libsvils.so, alternating; prints k_fin1_ksh / k_stop_ksh averages,
# then the K-sharded GPU tests.  Run on the GPU box: gpurun -- 'bash tools/ab_kshard_kernels.sh'  -> gpurun_out/r07y/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r07y
for lib in libsvils_prev.so libsvils.so libsvils_prev.so libsvils.so; do
  for spec in "mmsb:1000000:512:24 kshard 8 0" "astroph-k200 kshard 8 0"; do
    set -- $spec
    (cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/st_x; SVILS_LIB=$GRAFT_REPO_ROOT/svinet_amd/lib/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_x -o p -- python $GRAFT_REPO_ROOT/tools/shard_rank.py $spec 6 > /tmp/sr.log 2>&1)
    f=$(find /tmp/st_x -name '*kernel_stats.csv' | head -1)
    echo "$lib $1: $(tail -1 /tmp/sr.log) | k_fin1_ksh avg us: $(grep k_fin1_ksh $f | head -1 | awk -F, '{printf "%.1f", $(NF-4)/1000}') | k_stop_ksh avg us: $(grep k_stop_ksh $f | head -1 | awk -F, '{printf "%.1f", $(NF-4)/1000}')"
  done
done | tee gpurun_out/r07y/ab_fin1_stop_ksh.txt
timeout 1500 python -m pytest tests/test_gpu_ksharded.py tests/test_gpu_config5.py tests/test_gpu_native_ranks.py -q -m gpu -x --timeout 600 -k "kshard or ksh or K_shard or shards" 2>&1 | tail -3 | tee gpurun_out/r07y/pytest_kshard.txt
