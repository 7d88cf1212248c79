#!/bin/bash
# round 4, call E: conv64 statistics epilogue requested by the autograd nodes: parity tests of every node it touches + A/B
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4e; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_conv64_gpu.py tests/test_bnconv_gpu.py tests/test_stemfuse_gpu.py tests/test_conv3g_gpu.py tests/test_graph_gpu.py -x -q -m gpu ) > $O/pytest_a.log 2>&1; tail -n 6 $O/pytest_a.log
( time timeout 900 python -m pytest "tests/test_headline_gpu.py::test_bf16_step_at_1024_tracks_the_oracle" "tests/test_headline_gpu.py::test_ten_step_trajectory_matches_stock_torch" -x -q -m gpu ) > $O/pytest_b.log 2>&1; tail -n 4 $O/pytest_b.log
Q="--steps 20 --warmup 8 --no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --no-kernel-timing"
for rep in 1 2; do
  for v in "" "TSG_CONV_C64_STATS=0"; do
    ( env $v timeout 300 python bench.py $Q ) 2>&1 | grep '^{' | tail -n 1 | V="$v" python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('%-24s' % (os.environ['V'] or 'default'), d['value'], d['ms_per_step'], d['config']['final_loss'])"
  done
done 2>&1 | tee $O/ab.log
