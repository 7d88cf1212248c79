#!/bin/bash
# round 6, call 5: reproducible pointwise weight gradients — parity, drop-in bit equality, step A/B, families A/B
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out; O=gpurun_out/r6_call5.txt; : > $O
( timeout 1500 python -m pytest tests/test_pwconv_gpu.py tests/test_dropin_gpu.py tests/test_graph_gpu.py -x -q 2>&1 | tail -n 12 ) >> $O
for i in 1 2; do
  tools/r6/q.sh "TSG_PW_CONV=0 " TSG_PW_CONV=0 -- >> $O
  tools/r6/q.sh "TSG_PW_CONV=1 " TSG_PW_CONV=1 -- >> $O
done
for c in pspnet dfn psanet; do
  for v in 0 1; do
    tools/r6/q.sh "$c TSG_PW_CONV=$v " TSG_PW_CONV=$v -- --config $c --steps 15 --warmup 8 >> $O
  done
done
cat $O
