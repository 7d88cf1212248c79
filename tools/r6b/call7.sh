#!/bin/bash
# round 6b, call 7: pooled layers (conv1x1 + BN + ReLU / sigmoid) as one launch per direction; gap-backward scaling in one launch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b_call7.txt
{
timeout 900 python -m pytest tests/test_vecconv_gpu.py tests/test_pool_gpu.py -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_dropin_gpu.py tests/test_headline_gpu.py tests/test_graph_gpu.py -x -q 2>&1 | tail -8
for i in 1 2 3; do
tools/r6/q.sh "TSG_POOLED_LAYER=0" TSG_POOLED_LAYER=0 --
tools/r6/q.sh "TSG_POOLED_LAYER=1" TSG_POOLED_LAYER=1 --
done
} > $O 2>&1
cat $O
