#!/bin/bash
mkdir -p gpurun_out/r02a
timeout 900 python -m pytest tests -q -m gpu -x --timeout 600 2>&1 | tail -5
for nw in 16 8 4; do
echo "== SVILS_LPL_NW=$nw"
SVILS_LPL_NW=$nw python tools/phi_vs_work.py astroph-k20 8 4 2>&1 | grep -v amdgpu.ids | tail -2
SVILS_LPL_NW=$nw python tools/phi_vs_work.py astroph-k20 200 100 2>&1 | grep -v amdgpu.ids | tail -2
done
bash tools/gpu_job2.sh
