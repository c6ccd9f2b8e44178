cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout 2700 python -m pytest tests -q -m gpu -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
python tools/phi_vs_work.py astroph-k20 1300 100 2>/dev/null | tee $O/phi_vs_work_astroph_k20.txt
