#!/bin/bash
# round 6, call 7: weight-shadow launch with the data-gradient image written by destination
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out; O=gpurun_out/r6_call7.txt; : > $O
( timeout 900 python -m pytest tests/test_shadow_gpu.py tests/test_optim_gpu.py tests/test_conv3g_gpu.py -x -q 2>&1 | tail -n 6 ) >> $O
bash tools/r6/tl_ab.sh TSG_SHADOW_WF1_PASS 0 1 2>&1 | grep -E "step wall|total A|weight_shadow|sgd_multi" >> $O
for i in 1 2; do
  tools/r6/q.sh "TSG_SHADOW_WF1_PASS=0 " TSG_SHADOW_WF1_PASS=0 -- >> $O
  tools/r6/q.sh "TSG_SHADOW_WF1_PASS=1 " TSG_SHADOW_WF1_PASS=1 -- >> $O
done
cat $O
