#!/bin/bash
mkdir -p gpurun_out
for f in 0 1; do
  s=$(date +%s)
  timeout 900 python bench.py --steps 20 --warmup 5 --miopen-find $f --no-cpu-baseline > gpurun_out/bench_find$f.log 2>&1
  e=$(date +%s)
  echo "find=$f wall=$((e-s))s"; tail -1 gpurun_out/bench_find$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline'))"
done
s=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 --miopen-find 0 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_notimer.log 2>&1; e=$(date +%s)
echo "find=0 notimer wall=$((e-s))s"; tail -1 gpurun_out/bench_notimer.log | cut -c1-160
