#!/bin/bash
# round-2 GPU job: full gpu test-suite, per-kernel timings, the default bench line
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -q -m gpu -x --timeout 600 > gpurun_out/r02a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log
tail -15 gpurun_out/r02a/pytest.log
for wl in astroph-k20 lfr-k28; do timeout 300 python tools/kernel_times.py $wl 100 >> gpurun_out/r02a/kernel_times.txt 2>&1; done
cat gpurun_out/r02a/kernel_times.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02a/bench_astroph_k20.json 2> gpurun_out/r02a/bench.err
cat gpurun_out/r02a/bench_astroph_k20.json | cut -c1-600
