#!/bin/bash
# round 4, call B: exact fp32 convolutions (kernel parity, headline + families + smoke in the exact mode, fp64-truth
# diagnostic at 1024^2), weight gradient with a slot's pairs on two XCDs (512 -> 512), s_setprio around the MFMA clusters
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_exactconv_gpu.py -x -q -m gpu ) > $O/pytest_exact.log 2>&1; tail -n 5 $O/pytest_exact.log
( time timeout 900 python -m pytest tests/test_headline_gpu.py::test_fp32_logits_loss_and_kept_mask_at_1024 tests/test_families_gpu.py -x -q -m gpu -s ) > $O/pytest_fp32.log 2>&1; tail -n 5 $O/pytest_fp32.log; grep -E "^head|headline fp32|loss|grad" $O/pytest_fp32.log | head -30
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -n 3 $O/smoke.log
( SIZE=1024 BATCH=2 timeout 600 python tools/diag_fp64_truth.py ) > $O/fp64_truth_1024.log 2>&1; tail -n 9 $O/fp64_truth_1024.log
( time timeout 600 python -m pytest tests/test_convwrw_gpu.py tests/test_conv3g_gpu.py -x -q -m gpu ) > $O/pytest_conv.log 2>&1; tail -n 3 $O/pytest_conv.log
ONLY=layer4 python tools/bench_conv3wrw.py 2>&1 | tail -n 3 | tee $O/wrw_layer4_xs2.log
ONLY=layer4 TSG_CONV_WRW_XS2=0 python tools/bench_conv3wrw.py 2>&1 | tail -n 3 | tee $O/wrw_layer4_xs1.log
Q="--steps 20 --warmup 8 --no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --no-kernel-timing"
for rep in 1 2; do
  for v in "" "TSG_MFMA_PRIO=1" "TSG_CONV_WRW_XS2=0"; do
    ( env $v timeout 300 python bench.py $Q ) 2>&1 | grep '^{' | tail -n 1 | V="$v" python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('%-24s' % (os.environ['V'] or 'default'), d['value'], d['ms_per_step'])"
  done
done 2>&1 | tee $O/ab.log
