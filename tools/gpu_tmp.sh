cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_cli.py tests/test_gpu_native_ranks.py -q -m gpu --timeout 900 -k "kshard or minibatch" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
