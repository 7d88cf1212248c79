#!/bin/bash
# round 3, call B: the general conv kernel (parity + microbench), the bench with it, DFN config, split upsample backward
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_conv3g_gpu.py tests/test_upsample_gpu.py tests/test_optim_gpu.py -x -q -m gpu ) > $O/pytest_conv3g.log 2>&1
( time timeout 300 python tools/bench_conv3g.py ) > $O/bench_conv3g.log 2>&1
( time timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_bisenet_gen1.log 2>&1
( time TSG_CONV_GEN=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ohem-probe ) > $O/bench_bisenet_gen0.log 2>&1
( time TSG_CONV_GEN_STATS=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ohem-probe ) > $O/bench_bisenet_gen1_nostats.log 2>&1
for c in dfn pspnet; do
  ( time timeout 400 python bench.py --config $c --steps 10 --warmup 5 --no-cpu-baseline ) > $O/bench_$c.log 2>&1
done
tail -n 6 $O/pytest_conv3g.log; cat $O/bench_conv3g.log | tail -n 14
for f in $O/bench_bisenet_*.log $O/bench_dfn.log $O/bench_pspnet.log; do echo "== $f"; grep -o '"value": [0-9.]*' $f | head -1; grep -c Traceback $f; done
