#!/bin/bash
# round 6 (second session), call 1: BN NHWC passes with uniform addressing / 4 waves per SIMD vs the previous library; stem wrw_bn grid
cd $GRAFT_REPO_ROOT
export BENCH_BN_ONLY=1
O=gpurun_out/r6b_call1.txt
{
echo "### new library"; python tools/bench_bn.py
echo "### new library, TSG_BN_BLOCKS_APPLY=2048"; TSG_BN_BLOCKS_APPLY=2048 python tools/bench_bn.py
echo "### new library, TSG_BN_BLOCKS_REDUCE=2048 TSG_BN_BLOCKS_APPLY=1536"; TSG_BN_BLOCKS_REDUCE=2048 TSG_BN_BLOCKS_APPLY=1536 python tools/bench_bn.py
echo "### stem wrw_bn grid 768 / 512 / 1024"
python tools/r6b/bench_stem_bn.py; TSG_STEM_BN_GRID=512 python tools/r6b/bench_stem_bn.py; TSG_STEM_BN_GRID=1024 python tools/r6b/bench_stem_bn.py
cp torchseg_amd/libtsg_hip.so build/libtsg_new.so
cp build/libtsg_base.so torchseg_amd/libtsg_hip.so
echo "### previous library"; python tools/bench_bn.py
for i in 1 2; do
cp build/libtsg_base.so torchseg_amd/libtsg_hip.so
tools/r6/q.sh "previous library  " --
cp build/libtsg_new.so torchseg_amd/libtsg_hip.so
tools/r6/q.sh "new library       " --
tools/r6/q.sh "new, stem grid 512" TSG_STEM_BN_GRID=512 --
done
} > $O 2>&1
tail -8 $O
