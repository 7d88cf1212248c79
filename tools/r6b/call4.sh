#!/bin/bash
# round 6b, call 4: the ResNet stem as one node — 2 (gradient never stored) / 1 (nothing stored) / 0 (three modules)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b_call4.txt
{
timeout 900 python -m pytest tests/test_stempool_gpu.py tests/test_dropin_gpu.py tests/test_graph_gpu.py -x -q 2>&1 | tail -15
for i in 1 2 3; do
tools/r6/q.sh "TSG_STEM_RECOMPUTE=0" TSG_STEM_RECOMPUTE=0 --
tools/r6/q.sh "TSG_STEM_RECOMPUTE=2" TSG_STEM_RECOMPUTE=2 --
done
tools/r6/q.sh "TSG_STEM_RECOMPUTE=1" TSG_STEM_RECOMPUTE=1 --
} > $O 2>&1
cat $O
