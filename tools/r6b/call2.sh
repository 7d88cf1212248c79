#!/bin/bash
# round 6b, call 2: stem_wrw_k with __launch_bounds__(256, 3) (BN-on-load variant 180 -> 160 VGPRs: 3 blocks per CU like the plain one)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b_call2.txt
{
python tools/r6b/bench_stem_bn.py; TSG_STEM_BN_GRID=512 python tools/r6b/bench_stem_bn.py
timeout 600 python -m pytest tests/test_bnconv_gpu.py tests/test_stemconv_gpu.py -x -q 2>&1 | tail -3
cp torchseg_amd/libtsg_hip.so build/libtsg_new.so
for i in 1 2; do
cp build/libtsg_base.so torchseg_amd/libtsg_hip.so
tools/r6/q.sh "previous library  " --
cp build/libtsg_new.so torchseg_amd/libtsg_hip.so
tools/r6/q.sh "new library       " --
done
} > $O 2>&1
cat $O | grep -v amdgpu.ids
