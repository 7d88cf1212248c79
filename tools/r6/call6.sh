#!/bin/bash
# round 6, call 6: shortcut on the sub-sampled map — parity, drop-in bit equality, step A/B
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out; O=gpurun_out/r6_call6.txt; : > $O
( timeout 1500 python -m pytest tests/test_conv3g_gpu.py tests/test_pwconv_gpu.py tests/test_dropin_gpu.py tests/test_graph_gpu.py -x -q 2>&1 | tail -n 12 ) >> $O
for i in 1 2; do
  tools/r6/q.sh "TSG_SKIP_SUBSAMPLE=0 " TSG_SKIP_SUBSAMPLE=0 -- >> $O
  tools/r6/q.sh "TSG_SKIP_SUBSAMPLE=1 " TSG_SKIP_SUBSAMPLE=1 -- >> $O
done
cat $O
