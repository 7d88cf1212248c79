#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
s=$(date +%s); timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; e=$(date +%s); echo "bench default rc=$? wall=$((e-s))s"; tail -1 gpurun_out/bench_default.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline'])"
TSG_FORCE_COLLECTIVES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_forced_coll.log 2>&1; echo "forced-collectives rc=$?"; tail -1 gpurun_out/bench_forced_coll.log | cut -c1-330
TSG_FORCE_COLLECTIVES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --optimizer torch > gpurun_out/bench_forced_coll2.log 2>&1; echo "forced-collectives torch-sgd rc=$?"; tail -1 gpurun_out/bench_forced_coll2.log | cut -c1-330
