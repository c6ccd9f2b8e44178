#!/bin/bash
# quick A/B: parity subset + kernel times + stamps
TAG=${1:-r02e}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x --timeout 600 2>&1 | tail -3
for wl in astroph-k20 lfr-k28 astroph-k32 astroph-k8; do timeout 300 python tools/kernel_times.py $wl 200 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/$TAG/kernel_times.txt; done
SVILS_LIB=svinet_amd/lib/libsvils_stamps.so timeout 300 python tools/stamps.py astroph-k20 6 2>&1 | grep -v amdgpu.ids | head -10 | tee gpurun_out/$TAG/stamps_astroph_k20.txt
