#!/bin/bash
# Round-end verification on one MI355X: full GPU test suite, smoke, default bench, kernel-stats profile, PMC traffic.
mkdir -p gpurun_out
s=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_full.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 gpurun_out/pytest_gpu_full.log) [$(( $(date +%s) - s ))s]"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 gpurun_out/smoke.log)"
s=$(date +%s); timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench default rc=$? wall=$(( $(date +%s) - s ))s"; tail -1 gpurun_out/bench_default.log
bash tools/prof_bench.sh > gpurun_out/prof_bench.out 2>&1; head -3 gpurun_out/prof_bench.out
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.out 2>&1; tail -25 gpurun_out/pmc_traffic.out | cut -c1-260
