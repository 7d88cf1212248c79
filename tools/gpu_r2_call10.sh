#!/bin/bash
O=gpurun_out/c10; mkdir -p $O
timeout 600 python -m pytest tests/test_convwrw_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "== tests rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/pytest.log | cut -c1-220 | head -12
timeout 300 python tools/bench_conv3wrw.py 2>&1 | grep -v Warn | tee $O/bench_conv3wrw.log
for m in 64 512; do
  TSG_CONV_WRW_MAXC=$m timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 10 > $O/bench_maxc$m.log 2>&1; echo "== bench MAXC=$m: $(tail -1 $O/bench_maxc$m.log | cut -c1-150)"
done
