#!/bin/bash
mkdir -p gpurun_out/final
timeout 900 python -m pytest tests/test_bnconv_gpu.py tests/test_stemconv_gpu.py tests/test_stemfuse_gpu.py tests/test_headline_gpu.py tests/test_families_gpu.py tests/test_graph_gpu.py -x -q > gpurun_out/final/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/final/pytest_subset.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
timeout 600 python bench.py > gpurun_out/final/bench_default.log 2>&1; tail -1 gpurun_out/final/bench_default.log | cut -c1-200
bash tools/prof_bench.sh > gpurun_out/final/prof_bench.out 2>&1; tail -1 gpurun_out/final/prof_bench.out | cut -c1-80
bash tools/pmc_traffic.sh > gpurun_out/final/pmc_traffic.out 2>&1; tail -1 gpurun_out/final/pmc_traffic.out | cut -c1-80
