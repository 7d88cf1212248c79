#!/bin/bash
O=gpurun_out/c13; mkdir -p $O
timeout 600 python -m pytest tests/test_convwrw_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "== tests rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/pytest.log | cut -c1-220 | head -12
ONLY=s2 timeout 300 python tools/bench_conv3wrw.py 2>&1 | grep -v Warn | tee $O/bench_conv3wrw_s2.log
b() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 10 > $O/bench_$name.log 2>&1; echo "== $name: $(tail -1 $O/bench_$name.log | cut -c60-150)"; }
b base X=1
