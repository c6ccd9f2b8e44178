cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fuzz.py -q -m gpu --timeout 900 -k "too_large" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
