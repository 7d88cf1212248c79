#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4o; mkdir -p $O
export TMPDIR=/tmp
bash tools/prof_bench.sh > $O/prof_bench.out 2>&1; cp gpurun_out/prof/kernel_stats_compact.csv $O/kernel_stats.csv; grep "kernels total" $O/prof_bench.out
( timeout 600 python tools/prof_eager.py ) > $O/eager_ops.txt 2>&1; head -5 $O/eager_ops.txt
