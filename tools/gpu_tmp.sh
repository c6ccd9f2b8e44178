R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03z; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_ksharded.py tests/test_gpu_native_ranks.py tests/test_gpu_cli.py -q -m gpu --timeout 900 -k "ksh or kshard or kstep" > $O/pytest_ksh.log 2>&1; tail -5 $O/pytest_ksh.log
python tools/shard_cost.py mmsb:1000000:512:24 8 2>/dev/null | tee $O/cost2_mmsb.txt
python tools/shard_cost.py astroph-k200 4,8 2>/dev/null | tee $O/cost2_astroph.txt
