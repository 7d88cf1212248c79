for occ in 1 2; do echo "== TSG_CONV_WRW_OCC=$occ"; TSG_CONV_WRW_OCC=$occ timeout 300 python tools/bench_conv3wrw.py 2>&1 | grep -v "s2\|amdgpu" | cut -c1-70; done
TSG_CONV_WRW_OCC=2 timeout 600 python -m pytest tests/test_convwrw_gpu.py -x -q 2>&1 | tail -2
