#!/bin/bash
for cfg in "--batch 4 --size 256 --steps 5" "--batch 16 --size 512 --steps 5" "--batch 16 --size 1024 --steps 1 --warmup 3" "--batch 16 --size 1024 --steps 5 --warmup 3 --no-kernel-timing"; do
  timeout 600 python bench.py --graph 1 --no-cpu-baseline $cfg > gpurun_out/bench_ab.log 2>&1
  echo "== $cfg rc=$?"; tail -1 gpurun_out/bench_ab.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))" 2>/dev/null || tail -5 gpurun_out/bench_ab.log
done
