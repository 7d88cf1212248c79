#!/bin/bash
# round 6, call 4: bit-mask block tails + adaptive fold — parity, then the step A/B
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out; O=gpurun_out/r6_call4.txt; : > $O
( timeout 1200 python -m pytest tests/test_bn_gpu.py tests/test_convwrw_gpu.py tests/test_bnconv_gpu.py tests/test_conv64_gpu.py -x -q 2>&1 | tail -n 8 ) >> $O
for i in 1 2; do
  tools/r6/q.sh "round-5 forms (fold 1, y mask) " TSG_CONV_WRW_FOLD=1 TSG_BN_MASKBITS=0 -- >> $O
  tools/r6/q.sh "fold 2                         " TSG_CONV_WRW_FOLD=2 TSG_BN_MASKBITS=0 -- >> $O
  tools/r6/q.sh "fold 2 + bit masks             " TSG_CONV_WRW_FOLD=2 TSG_BN_MASKBITS=1 -- >> $O
done
cat $O
