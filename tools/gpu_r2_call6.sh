#!/bin/bash
O=gpurun_out/c6; mkdir -p $O
run() { tag=$1; shift; echo "=== $tag"; env "$@" timeout 300 python tools/debug_graph4.py --tag "$tag" ${ARGS} 2>&1 | grep "^\[.*GRAPH" | cut -c1-150; }
{ ARGS="--steps 3 --side 3" run lazyOnD_restOnY X=1; ARGS="--steps 3 --side 4" run allOnY X=1; ARGS="--steps 3 --side 5" run lastStepOnD X=1; } > $O/graph4.log 2>&1; cat $O/graph4.log
