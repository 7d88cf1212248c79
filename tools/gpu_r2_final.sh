#!/bin/bash
# end-of-round evidence: full GPU suite, smoke, default bench, kernel-trace stats and PMC traffic of the bench command
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/final/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/final/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
timeout 600 python bench.py > gpurun_out/final/bench_default.log 2>&1; tail -1 gpurun_out/final/bench_default.log | cut -c1-400
bash tools/prof_bench.sh > gpurun_out/final/prof_bench.out 2>&1; tail -1 gpurun_out/final/prof_bench.out | cut -c1-100
bash tools/pmc_traffic.sh > gpurun_out/final/pmc_traffic.out 2>&1; tail -1 gpurun_out/final/pmc_traffic.out | cut -c1-100
