#!/bin/bash
mkdir -p gpurun_out
for cfg in "--graph 0 --optimizer torch" "--graph 0 --optimizer fused" "--graph 1 --optimizer fused" "--graph 1 --optimizer torch"; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $cfg > gpurun_out/bench_ab.log 2>&1
  echo "== $cfg rc=$?"; tail -1 gpurun_out/bench_ab.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'), d['roofline']['kernel'], d['roofline']['frac'])" 2>/dev/null || tail -5 gpurun_out/bench_ab.log
done
