#!/bin/bash
# PSA dA: untracked prefetch of both staged tiles, TSG_PSA_UT_PF register sets (1 = tracked loads, one set: round 3 until now)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3q; mkdir -p $O
export TMPDIR=/tmp
for pf in 2 3; do
  ( TSG_PSA_UT_PF=$pf timeout 300 python -m pytest tests/test_psa_gpu.py -x -q -m gpu ) > $O/pytest_pf$pf.log 2>&1; echo "ut_pf=$pf: $(tail -n 1 $O/pytest_pf$pf.log)"
done
for rep in 1 2; do for pf in 1 2 3; do
  echo "== ut_pf=$pf: $(PSA_QUICK=1 TSG_PSA_UT_PF=$pf timeout 200 python tools/bench_psa.py 2>&1 | grep bfloat16 | cut -c1-170)"
done; done
