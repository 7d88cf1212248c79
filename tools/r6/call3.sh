#!/bin/bash
# round 6, call 3: SpatialPath fork inside the captured graph
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out; O=gpurun_out/r6_call3.txt; : > $O
for i in 1 2; do
  tools/r6/q.sh "graph, one stream          " -- >> $O
  tools/r6/q.sh "graph, spatial path forked " TSG_FORK_SPATIAL=1 TSG_FORK_SPATIAL_IN_GRAPH=1 -- >> $O
done
tools/r6/q.sh "eager, spatial path forked " TSG_FORK_SPATIAL=1 -- --graph 0 >> $O
cat $O
