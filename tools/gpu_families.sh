#!/bin/bash
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_families_gpu.py tests/test_psa_gpu.py -x -q -s > gpurun_out/families.log 2>&1; echo "families+psa rc=$?"
grep -E "rel-L2|passed|failed|Error|assert" gpurun_out/families.log | head -20
timeout 200 python tools/bench_psa.py > gpurun_out/bench_psa.log 2>&1; echo "bench_psa rc=$?"; tail -12 gpurun_out/bench_psa.log
