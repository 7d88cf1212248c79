#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4r; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_convwrw_gpu.py tests/test_bnconv_gpu.py tests/test_stemfuse_gpu.py -q -m gpu ) > $O/pytest_a.log 2>&1; tail -n 4 $O/pytest_a.log
Q="--steps 20 --warmup 8 --no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --no-kernel-timing"
for rep in 1 2 3; do
  for v in "" "TSG_CONV_WRW_IMPL=tr"; do
    ( env $v timeout 300 python bench.py $Q ) 2>&1 | grep '^{' | tail -n 1 | V="$v" python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('%-24s' % (os.environ['V'] or 'default'), d['value'], d['ms_per_step'], d['config']['final_loss'])"
  done
done 2>&1 | tee $O/ab.log
