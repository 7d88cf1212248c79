#!/bin/bash
O=gpurun_out/c4; mkdir -p $O
run() { tag=$1; shift; echo "=== $tag"; env "$@" timeout 300 python tools/debug_graph4.py --tag "$tag" ${ARGS} 2>&1 | grep "^\[" | cut -c1-330; }
{
ARGS="--steps 6" run base X=1
ARGS="--steps 6 --opt torch" run torchopt X=1
ARGS="--steps 4" run nofeat TSG_STEM_CONV=0 TSG_CONV_WRW=0 TSG_SPLIT_BIAS=0
ARGS="--steps 4 --batch 4" run batch4 X=1
ARGS="--steps 4 --size 512" run size512 X=1
ARGS="--steps 4 --side 0" run noside X=1
} > $O/graph4.log 2>&1
cat $O/graph4.log
