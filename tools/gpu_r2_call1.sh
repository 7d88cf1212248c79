#!/bin/bash
# round 2, GPU call 1: new parity tests, comm path, hipGraph root-causing, benches
O=gpurun_out/c1; mkdir -p $O
timeout 1000 python -m pytest tests/test_headline_gpu.py tests/test_ce_gpu.py tests/test_comm_gpu.py tests/test_upsample_gpu.py tests/test_ohem_gpu.py tests/test_families_gpu.py tests/test_optim_gpu.py -q -m gpu -s > $O/pytest_new.log 2>&1
echo "== pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|headline|fp32 fused|bf16 \[" $O/pytest_new.log | tail -30
for v in "--opt torch" "--opt fused" "--opt fused --inside 1"; do
  echo "== graph debug $v"; timeout 300 python tools/debug_graph2.py $v --steps 8 2>&1 | grep -v Warning | tail -12
done > $O/graph_debug.log 2>&1
cat $O/graph_debug.log
timeout 600 python bench.py > $O/bench_default.log 2>&1; echo "== bench default rc=$?"; tail -1 $O/bench_default.log | cut -c1-600
for e in "TSG_COMM=1" "TSG_COMM=0" "TSG_COMM=1 TSG_XGMI_ONESHOT=1"; do
  env TSG_FORCE_COLLECTIVES=1 $e timeout 400 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 10 > $O/bench_force_$(echo $e | tr ' =' '__').log 2>&1
  echo "== forced collectives $e rc=$?"; tail -1 $O/bench_force_$(echo $e | tr ' =' '__').log | cut -c1-200
done
