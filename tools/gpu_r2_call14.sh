#!/bin/bash
O=gpurun_out/c14; mkdir -p $O
b() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 10 > $O/bench_$name.log 2>&1; echo "== $name: $(tail -1 $O/bench_$name.log | cut -c60-150)"; }
b dgrad1 TSG_CONV_DGRAD_FWD=1
b dgrad2 TSG_CONV_DGRAD_FWD=2
b dgrad2_find TSG_CONV_DGRAD_FWD=2 TSG_MIOPEN_FIND=1
b graph2 TSG_GRAPH=2
