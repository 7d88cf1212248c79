#!/bin/bash
mkdir -p gpurun_out/c28
timeout 900 python -m pytest tests/test_bnconv_gpu.py tests/test_conv64_gpu.py tests/test_convwrw_gpu.py tests/test_headline_gpu.py tests/test_families_gpu.py -x -q > gpurun_out/c28/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/c28/pytest.log
TSG_BN_ON_LOAD=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/c28/bench_0.log 2>&1; tail -1 gpurun_out/c28/bench_0.log | cut -c1-200
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/c28/bench_1.log 2>&1; tail -1 gpurun_out/c28/bench_1.log | cut -c1-200
