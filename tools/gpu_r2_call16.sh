#!/bin/bash
mkdir -p gpurun_out/c16
timeout 900 python -m pytest tests/test_headline_gpu.py tests/test_families_gpu.py tests/test_fused_head_gpu.py tests/test_ce_gpu.py tests/test_graph_gpu.py -x -q > gpurun_out/c16/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/c16/pytest.log
TSG_FUSE_HEAD=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/c16/bench_head0.log 2>&1; tail -1 gpurun_out/c16/bench_head0.log | cut -c1-200
TSG_FUSE_HEAD=1 timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/c16/bench_head1.log 2>&1; tail -1 gpurun_out/c16/bench_head1.log | cut -c1-200
