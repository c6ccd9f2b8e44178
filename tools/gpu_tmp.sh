cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_stochastic.py tests/test_gpu_properties.py tests/test_gpu_sharded.py tests/test_gpu_resume.py tests/test_gpu_config5.py tests/test_gpu_ksharded.py -q -m gpu -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
for wl in astroph-k200 synthetic:200000:512:24 mmsb:1000000:512:24; do
  python tools/kernel_times.py $wl 10 2>/dev/null | tee -a $O/derive_m.txt
  SVILS_DERIVE_M=0 python tools/kernel_times.py $wl 10 2>/dev/null | sed 's/^default/derive_m=0/' | tee -a $O/derive_m.txt
done
SVILS_EPI_MAX_MB=100000 python tools/kernel_times.py mmsb:1000000:512:24 10 2>/dev/null | sed 's/^default/derive_m=1,epi/' | tee -a $O/derive_m.txt
SVILS_DERIVE_M=0 SVILS_EPI_MAX_MB=100000 python tools/kernel_times.py mmsb:1000000:512:24 10 2>/dev/null | sed 's/^default/derive_m=0,epi/' | tee -a $O/derive_m.txt
