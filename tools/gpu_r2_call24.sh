#!/bin/bash
mkdir -p gpurun_out/c24
for g in 0 2 1; do
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --graph $g > gpurun_out/c24/bench_g$g.log 2>&1; echo "graph=$g"; tail -1 gpurun_out/c24/bench_g$g.log | cut -c1-330
done
