#!/bin/bash
# conv3g: filter slab by LDS-DMA (TSG_CONV3G_GLDS=1) vs registers + ds_write: parity tests, microbench, bench A/B
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3j; mkdir -p $O
export TMPDIR=/tmp
( TSG_CONV3G_GLDS=1 timeout 600 python -m pytest tests/test_conv3g_gpu.py tests/test_bnconv_gpu.py -x -q -m gpu ) > $O/pytest_glds.log 2>&1; tail -n 2 $O/pytest_glds.log
for g in 0 1; do
  ( TSG_CONV3G_GLDS=$g timeout 200 python tools/bench_conv3g.py ) > $O/conv3g_glds$g.log 2>&1
  echo "== GLDS=$g"; grep -v amdgpu.ids $O/conv3g_glds$g.log | cut -c1-200
done
for rep in 1 2; do for g in 0 1; do
  echo "bench GLDS=$g: $(TSG_CONV3G_GLDS=$g timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-ohem-probe 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"
done; done
echo "dfn GLDS=0: $(TSG_CONV3G_GLDS=0 timeout 300 python bench.py --config dfn --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"
echo "dfn GLDS=1: $(TSG_CONV3G_GLDS=1 timeout 300 python bench.py --config dfn --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"
