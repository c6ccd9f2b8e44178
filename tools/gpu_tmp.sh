cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_native_ranks.py tests/test_gpu_sharded.py -q -m gpu --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
