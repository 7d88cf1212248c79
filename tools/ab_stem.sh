#!/bin/bash
# A/B of the stem-conv swap inside the whole training step (same box, back to back), plus the smoke parity check
mkdir -p gpurun_out
for v in 1 0 1; do
  TSG_STEM_CONV=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_stem_$v.log 2>&1
  echo "TSG_STEM_CONV=$v rc=$? $(tail -1 gpurun_out/ab_stem_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
