#!/bin/bash
# round 6b, call 3: the recomputing ResNet stem — parity, kernel times, step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b_call3.txt
{
timeout 900 python -m pytest tests/test_stempool_gpu.py -x -q 2>&1 | tail -15
timeout 300 python tools/r6b/bench_stem_pool.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
tools/r6/q.sh "TSG_STEM_RECOMPUTE=0" TSG_STEM_RECOMPUTE=0 --
tools/r6/q.sh "TSG_STEM_RECOMPUTE=1" TSG_STEM_RECOMPUTE=1 --
done
} > $O 2>&1
cat $O
