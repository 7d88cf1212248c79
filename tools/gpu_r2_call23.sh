#!/bin/bash
mkdir -p gpurun_out/c23
timeout 900 python -m pytest tests/test_shadow_gpu.py tests/test_convwrw_gpu.py tests/test_optim_gpu.py tests/test_graph_gpu.py -x -q > gpurun_out/c23/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c23/pytest.log
TSG_WEIGHT_SHADOW=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/c23/bench_0.log 2>&1; tail -1 gpurun_out/c23/bench_0.log | cut -c1-200
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/c23/bench_1.log 2>&1; tail -1 gpurun_out/c23/bench_1.log | cut -c1-200
