#!/bin/bash
# round 6, call 8: gate backward in two phases (pool.gated_scale) — parity, drop-in bit equality, step A/B
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out; O=gpurun_out/r6_call8.txt; : > $O
( timeout 1500 python -m pytest tests/test_pool_gpu.py tests/test_dropin_gpu.py tests/test_graph_gpu.py tests/test_families_gpu.py -x -q 2>&1 | tail -n 8 ) >> $O
for i in 1 2; do
  tools/r6/q.sh "TSG_GATE_SPLIT=0 " TSG_GATE_SPLIT=0 -- >> $O
  tools/r6/q.sh "TSG_GATE_SPLIT=1 " TSG_GATE_SPLIT=1 -- >> $O
done
cat $O
