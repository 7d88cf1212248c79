#!/bin/bash
mkdir -p gpurun_out
timeout 250 python -m pytest tests/test_optim_gpu.py tests/test_stemconv_gpu.py -x -q -k "not full_size" > gpurun_out/round_tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/round_tests.log)"
grep -E "^E |Error" gpurun_out/round_tests.log | head -10
MIOPEN_LOG_LEVEL=1 timeout 100 python tools/bench_stem.py 2>&1 | grep -v amdgpu.ids | head -3
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1; echo "bench rc=$? $(tail -1 gpurun_out/bench_quick.log | cut -c1-120)"
