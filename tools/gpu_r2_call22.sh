#!/bin/bash
mkdir -p gpurun_out/c22
timeout 900 python -m pytest tests/test_bn_gpu.py tests/test_bn_multirank_gpu.py tests/test_pool_gpu.py tests/test_stemconv_gpu.py tests/test_convwrw_gpu.py -x -q > gpurun_out/c22/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c22/pytest.log
bash tools/prof_bench.sh > gpurun_out/prof_bench.out 2>&1; tail -2 gpurun_out/prof_bench.out | cut -c1-150
