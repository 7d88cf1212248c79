#!/bin/bash
mkdir -p gpurun_out/c18
for h in 0 1; do
TSG_FUSE_HEAD=$h timeout 600 python -m pytest tests/test_headline_gpu.py -x -q -s -k ten_step > gpurun_out/c18/traj_head$h.log 2>&1; echo "head=$h rc=$?"; grep -a "^fp32\|^bf16" gpurun_out/c18/traj_head$h.log | cut -c1-400
done
