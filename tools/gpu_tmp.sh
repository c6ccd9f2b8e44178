cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_ksharded.py tests/test_gpu_native_ranks.py -q -m gpu --timeout 900 -k "minibatch or step_ksharded or rejects or virtual or active_set or native_sweep_ksharded" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -60 $O/pytest.log
