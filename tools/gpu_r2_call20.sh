#!/bin/bash
mkdir -p gpurun_out/c20
timeout 900 python -m pytest tests/test_stemfuse_gpu.py -x -q > gpurun_out/c20/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c20/pytest.log
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/c20/bench_1.log 2>&1; tail -1 gpurun_out/c20/bench_1.log | cut -c1-200
tail -1 gpurun_out/c20/bench_1.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
for k, v in d['kernels_last_warmup_step'].items():
    if 'pool' in k or 'stem' in k: print(k, v)
"
