#!/bin/bash
mkdir -p gpurun_out/c17
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/c17/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/c17/pytest.log
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/c17/bench.log 2>&1; tail -1 gpurun_out/c17/bench.log | cut -c1-1500
