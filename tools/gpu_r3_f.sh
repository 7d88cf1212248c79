#!/bin/bash
# round 3, call F: skip-gradient fusion (tests + bench A/B), smoke
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_conv3g_gpu.py tests/test_conv64_gpu.py tests/test_convwrw_gpu.py tests/test_bnconv_gpu.py tests/test_stemfuse_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1
( time timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-ohem-probe ) > $O/bench_skip1.log 2>&1
( time TSG_FUSE_SKIP_GRAD=0 timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-ohem-probe ) > $O/bench_skip0.log 2>&1
( time timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-ohem-probe ) > $O/bench_skip1b.log 2>&1
tail -n 5 $O/pytest.log; tail -n 2 $O/smoke.log
for f in $O/bench_*.log; do echo "== $f"; grep -o '"value": [0-9.]*' $f | head -1; grep -c Traceback $f; done
