#!/bin/bash
# PSA: NT A operand in MFMA fragment order from global (TSG_PSA_CFG=af<BM>x<BN>x<PF>): parity tests + microbench against the default
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
CFGS=${CFGS:-"af256x64x2 af256x64x3 af128x64x2"}
for c in $CFGS; do
  ( TSG_PSA_CFG=$c timeout 400 python -m pytest tests/test_psa_gpu.py -x -q -m gpu ) > $O/pytest_$c.log 2>&1; echo "$c: $(tail -n 1 $O/pytest_$c.log)"
done
for rep in 1 2; do for c in split128x64x1 $CFGS; do
  echo "== $c: $(PSA_QUICK=1 TSG_PSA_CFG=$c timeout 200 python tools/bench_psa.py 2>&1 | grep bfloat16 | cut -c1-170)"
done; done
