#!/bin/bash
# round 6, call 2: BatchNorm backward sums in the 64-channel data gradients — parity, microbench, step A/B
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out; O=gpurun_out/r6_call2.txt; : > $O
( timeout 900 python -m pytest tests/test_conv64_gpu.py tests/test_bnconv_gpu.py -x -q 2>&1 | tail -n 15 ) >> $O
( timeout 300 python tools/bench_bsum.py 2>&1 | tail -n 8 ) >> $O
for i in 1 2; do
  tools/r6/q.sh "TSG_BN_BSUM=0 " TSG_BN_BSUM=0 -- >> $O
  tools/r6/q.sh "TSG_BN_BSUM=1 " TSG_BN_BSUM=1 -- >> $O
done
cat $O
