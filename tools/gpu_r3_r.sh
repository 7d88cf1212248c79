#!/bin/bash
# conv3 weight gradient two tiles ahead with untracked loads (TSG_CONV_WRW_PF2=1, stride 1) against PF 0 / 1 (=0): parity,
# microbench, headline bench A/B on one box
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3r; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_convwrw_gpu.py tests/test_bnconv_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1; tail -n 1 $O/pytest.log
( timeout 200 python tools/bench_conv3wrw.py ) > $O/wrw_pf2.log 2>&1; grep -v amdgpu.ids $O/wrw_pf2.log | cut -c1-120
for pf in 0 1; do
  echo "bench pf2=$pf: $(TSG_CONV_WRW_PF2=$pf timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-ohem-probe 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"
done
