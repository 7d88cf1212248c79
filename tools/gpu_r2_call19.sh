#!/bin/bash
mkdir -p gpurun_out/c19
timeout 900 python -m pytest tests/test_stemfuse_gpu.py tests/test_stemconv_gpu.py tests/test_bn_gpu.py tests/test_pool_gpu.py -x -q > gpurun_out/c19/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/c19/pytest.log
TSG_FUSE_STEM_POOL=0 TSG_STEM_STATS=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/c19/bench_0.log 2>&1; tail -1 gpurun_out/c19/bench_0.log | cut -c1-200
TSG_FUSE_STEM_POOL=1 TSG_STEM_STATS=1 timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/c19/bench_1.log 2>&1; tail -1 gpurun_out/c19/bench_1.log | cut -c1-200
tail -1 gpurun_out/c19/bench_1.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
for k, v in d['kernels_last_warmup_step'].items(): print(k, v)
print(d['roofline'])
"
