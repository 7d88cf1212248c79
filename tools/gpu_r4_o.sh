#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4o; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_conv3g_gpu.py tests/test_vecconv_gpu.py -q -m gpu ) > $O/pytest_f.log 2>&1; tail -n 6 $O/pytest_f.log
( timeout 300 python tools/bench_s2dgrad.py ) 2>&1 | tail -n 3 | tee $O/s2dgrad2.log
Q="--steps 20 --warmup 8 --no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --no-kernel-timing"
for rep in 1 2; do
  for v in "" "TSG_CONV_S2_DGRAD=0"; do
    ( env $v timeout 300 python bench.py $Q ) 2>&1 | grep '^{' | tail -n 1 | V="$v" python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('%-24s' % (os.environ['V'] or 'default'), d['value'], d['ms_per_step'], d['config']['final_loss'])"
  done
done 2>&1 | tee $O/ab6.log
