#!/bin/bash
mkdir -p gpurun_out/c21
timeout 900 python -m pytest tests/test_convwrw_gpu.py tests/test_conv64_gpu.py tests/test_headline_gpu.py -x -q > gpurun_out/c21/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c21/pytest.log
TSG_CONV_C64=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/c21/bench_0.log 2>&1; tail -1 gpurun_out/c21/bench_0.log | cut -c1-200
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/c21/bench_1.log 2>&1; tail -1 gpurun_out/c21/bench_1.log | cut -c1-200
