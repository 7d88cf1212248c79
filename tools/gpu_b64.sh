for occ in 1 2; do echo "== OCC=$occ"; TSG_C64_OCC=$occ timeout 200 python tools/bench_conv64.py 2>&1 | grep "H="; done
TSG_C64_OCC=2 timeout 300 python -m pytest tests/test_conv64_gpu.py -x -q 2>&1 | tail -2
