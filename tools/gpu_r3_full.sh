#!/bin/bash
# the driver's round-end sequence: full GPU suite, smoke, default bench
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3full; mkdir -p $O
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -n 6 $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver_cmd.log 2>&1; tail -n 4 $O/bench_driver_cmd.log | cut -c1-600
