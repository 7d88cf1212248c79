#!/bin/bash
# round 3, call E: k-th-branch tail after the wave-parallel histogram walk, OHEM / headline tests, full kernel trace of the
# bench, PSA tile order
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_ohem_gpu.py tests/test_fused_head_gpu.py tests/test_ce_gpu.py tests/test_bnconv_gpu.py tests/test_psa_gpu.py -x -q -m gpu ) > $O/pytest_ohem.log 2>&1
( time timeout 900 python -m pytest tests/test_headline_gpu.py -x -q -m gpu -s ) > $O/pytest_headline.log 2>&1
( REPS=5 timeout 200 python tools/bench_head_kth.py ) > $O/kth.log 2>&1
bash tools/prof_bench.sh > $O/prof_bench.out 2>&1
cp gpurun_out/prof/kernel_stats_compact.csv $O/kernel_stats.csv
for ord in n m; do
  ( TSG_PSA_ORDER=$ord PSA_QUICK=1 timeout 200 python tools/bench_psa.py ) > $O/psa_order_$ord.log 2>&1
  ( TSG_PSA_CFG=split128x128x1 TSG_PSA_ORDER=$ord PSA_QUICK=1 timeout 200 python tools/bench_psa.py ) > $O/psa128_order_$ord.log 2>&1
done
tail -n 4 $O/pytest_ohem.log; grep -E "passed|failed|batch 16|band|head [0-9]" $O/pytest_headline.log | tail -n 12
tail -n 1 $O/kth.log | cut -c1-700
tail -n 3 $O/prof_bench.out | cut -c1-200
for f in $O/psa*_order_*.log; do echo "== $f"; grep -v amdgpu.ids $f | cut -c1-200; done
head -n 70 $O/kernel_stats.csv | cut -c1-170
