#!/bin/bash
O=gpurun_out/c12; mkdir -p $O
b() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 10 > $O/bench_$name.log 2>&1; echo "== $name: $(tail -1 $O/bench_$name.log | cut -c60-150)"; }
b base X=1
b bnrev TSG_BN_REVERSE=1
b wrw512 TSG_CONV_WRW_BLOCKS=512
b wrw128 TSG_CONV_WRW_BLOCKS=128
b base2 X=1
b bnrev2 TSG_BN_REVERSE=1
timeout 300 python -m pytest tests/test_bn_gpu.py tests/test_convwrw_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "== tests rc=$?"; tail -1 $O/pytest.log
TSG_BN_REVERSE=1 timeout 300 python -m pytest tests/test_bn_gpu.py -q -m gpu -x > $O/pytest_rev.log 2>&1; echo "== tests (reverse) rc=$?"; tail -1 $O/pytest_rev.log
