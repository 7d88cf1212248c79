#!/bin/bash
# quick check after a kernel change: the touched parity tests, two short bench runs, the kernel table of the step
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
T=${1:-"tests/test_bn_gpu.py tests/test_bnconv_gpu.py"}
Q="--no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0"
python -m pytest $T -x -q 2>&1 | tail -3
for i in 1 2; do python bench.py --steps 20 --warmup 8 $Q --no-kernel-timing 2>/dev/null | grep -o '"value": [0-9.]*, "unit": "img/s", "n_gpus": 1, "steps": 20, "warmup": 8, "ms_per_step": [0-9.]*' | head -1; done
bash tools/prof_bench.sh 2>&1 | grep -E "kernels total|bn_|ohem_up" | cut -c1-140
cp gpurun_out/prof/kernel_stats_compact.csv gpurun_out/kernel_stats_latest.csv
