#!/bin/bash
mkdir -p gpurun_out/c15
timeout 600 python -m pytest tests/test_fused_head_gpu.py tests/test_ohem_gpu.py -x -q > gpurun_out/c15/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/c15/pytest.log
timeout 300 python tools/bench_head.py > gpurun_out/c15/bench_head.log 2>&1; cat gpurun_out/c15/bench_head.log
