#!/bin/bash
# round 3, call C: conv3g configs (8 waves x 128 / 64 oc, 4 waves x 64 oc), its counters, PSA tile configs, tests
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_conv3g_gpu.py tests/test_upsample_gpu.py tests/test_optim_gpu.py tests/test_psa_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1
( timeout 200 python tools/bench_conv3g.py ) > $O/conv3g_default.log 2>&1
( TSG_CONV3G_BN=64 timeout 200 python tools/bench_conv3g.py ) > $O/conv3g_64x4.log 2>&1
( TSG_CONV3G_BN=64 TSG_CONV3G_NW=8 timeout 200 python tools/bench_conv3g.py ) > $O/conv3g_64x8.log 2>&1
for cfg in 128x1 128x128x1 128x128x2; do
  ( TSG_PSA_CFG=$cfg PSA_QUICK=1 timeout 200 python tools/bench_psa.py ) > $O/psa_$cfg.log 2>&1
done
LAYER=layer2 bash tools/pmc_kernel.sh tools/bench_conv3g.py conv3g_fwd_k c3g_default > $O/pmc_conv3g_default.txt 2>&1
LAYER=layer2 TSG_CONV3G_BN=64 bash tools/pmc_kernel.sh tools/bench_conv3g.py conv3g_fwd_k c3g_64x4 > $O/pmc_conv3g_64x4.txt 2>&1
PSA_QUICK=1 bash tools/pmc_kernel.sh tools/bench_psa.py psa_mm psa > $O/pmc_psa.txt 2>&1
( time timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_bisenet.log 2>&1
tail -n 4 $O/pytest.log
for f in $O/conv3g_*.log; do echo "== $f"; grep -v amdgpu.ids $f | cut -c1-230; done
for f in $O/psa_*.log; do echo "== $f"; grep -v amdgpu.ids $f | cut -c1-300; done
cat $O/pmc_*.txt
grep -o '"value": [0-9.]*' $O/bench_bisenet.log | head -1
grep -o '"ohem_kth_branch".*"selection_tail_us": [0-9.]*' $O/bench_bisenet.log | cut -c1-600
