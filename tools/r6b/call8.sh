#!/bin/bash
# round 6b, call 8: SpatialPath forked onto a side stream INSIDE the captured graph (one fork / join per direction)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b_call8.txt
{
for i in 1 2; do
tools/r6/q.sh "graph, no fork            " --
tools/r6/q.sh "graph, spatial path forked" TSG_FORK_SPATIAL=1 TSG_FORK_IN_GRAPH=1 --
done
tools/r6/q.sh "eager, spatial path forked" TSG_FORK_SPATIAL=1 -- --graph 0
tools/r6/q.sh "eager                     " -- --graph 0
} > $O 2>&1
cat $O
