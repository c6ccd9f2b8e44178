#!/bin/bash
# full gpu suite + per-kernel times + in-kernel stamps:  gpu_job5.sh [tag]
TAG=${1:-r02d}
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests -q -m gpu -x --timeout 600 > gpurun_out/$TAG/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
tail -5 gpurun_out/$TAG/pytest.log
for wl in astroph-k20 lfr-k28; do timeout 300 python tools/kernel_times.py $wl 200 2>&1 | grep -v amdgpu.ids >> gpurun_out/$TAG/kernel_times.txt; done
cat gpurun_out/$TAG/kernel_times.txt
if [ -f svinet_amd/lib/libsvils_stamps.so ]; then
  SVILS_LIB=svinet_amd/lib/libsvils_stamps.so timeout 300 python tools/stamps.py astroph-k20 60 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$TAG/stamps_astroph_k20.txt
fi
