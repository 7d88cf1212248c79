#!/bin/bash
# round 6b, call 5: pooled stem weight gradient that READS y (3 waves with spills / 2 waves) against the re-evaluating one
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b_call5.txt
{
timeout 900 python -m pytest tests/test_stempool_gpu.py -x -q 2>&1 | tail -5
timeout 300 python tools/r6b/bench_stem_pool.py 2>&1 | grep -v amdgpu.ids
TSG_STEM_POOL_WAVES=2 timeout 300 python tools/r6b/bench_stem_pool.py 2>&1 | grep "waves"
for i in 1 2; do
tools/r6/q.sh "TSG_STEM_RECOMPUTE=0" TSG_STEM_RECOMPUTE=0 --
tools/r6/q.sh "TSG_STEM_RECOMPUTE=2 waves 3" TSG_STEM_RECOMPUTE=2 --
tools/r6/q.sh "TSG_STEM_RECOMPUTE=2 waves 2" TSG_STEM_RECOMPUTE=2 TSG_STEM_POOL_WAVES=2 --
done
} > $O 2>&1
cat $O
