#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b_call6.txt
{
timeout 300 python tools/r6b/bench_stem_pool.py 2>&1 | grep -v amdgpu.ids
TSG_STEM_POOL_WAVES=2 timeout 300 python tools/r6b/bench_stem_pool.py 2>&1 | grep "waves"
for i in 1 2 3; do
tools/r6/q.sh "TSG_STEM_RECOMPUTE=0               " TSG_STEM_RECOMPUTE=0 --
tools/r6/q.sh "TSG_STEM_RECOMPUTE=2 (y remade)    " TSG_STEM_RECOMPUTE=2 --
tools/r6/q.sh "TSG_STEM_RECOMPUTE=2 (y read)      " TSG_STEM_RECOMPUTE=2 TSG_STEM_WRW_READS_Y=1 --
done
} > $O 2>&1
cat $O
