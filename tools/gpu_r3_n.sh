#!/bin/bash
# conv3g: filter fragments of the next tap prefetched (sched_barrier-pinned): parity, microbench, headline bench
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3n; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_conv3g_gpu.py tests/test_bnconv_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1; tail -n 2 $O/pytest.log
( timeout 200 python tools/bench_conv3g.py ) > $O/conv3g.log 2>&1; grep -v amdgpu.ids $O/conv3g.log | cut -c1-200
for rep in 1 2; do
  echo "bench: $(timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-ohem-probe 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"
done
