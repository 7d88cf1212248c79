#!/bin/bash
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_convwrw_gpu.py -x -q > gpurun_out/convwrw_tests.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/convwrw_tests.log)"; grep -E "^E " gpurun_out/convwrw_tests.log | head -8
MIOPEN_LOG_LEVEL=1 timeout 100 python tools/bench_conv3wrw.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_conv3wrw.log
for v in 1 0; do
  TSG_CONV_WRW=$v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_convwrw_$v.log 2>&1
  echo "TSG_CONV_WRW=$v rc=$? $(tail -1 gpurun_out/ab_convwrw_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))" 2>&1 | tail -1)"
done
