cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s; mkdir -p $O
timeout 2700 python -m pytest tests -q -m gpu -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
bash tools/k_scan.sh 2>/dev/null | tee $O/k_scan_astroph.txt
