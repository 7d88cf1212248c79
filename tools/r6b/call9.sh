#!/bin/bash
# round 6b, call 9: auxiliary heads forked onto side streams inside the captured graph
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b_call9.txt
{
for i in 1 2; do
tools/r6/q.sh "graph, no fork                 " --
tools/r6/q.sh "graph, heads forked            " TSG_FORK_HEADS=1 TSG_FORK_IN_GRAPH=1 --
tools/r6/q.sh "graph, heads + spatial forked  " TSG_FORK_HEADS=1 TSG_FORK_SPATIAL=1 TSG_FORK_IN_GRAPH=1 --
done
tools/r6/q.sh "eager, heads forked            " TSG_FORK_HEADS=1 -- --graph 0
tools/r6/q.sh "eager                          " -- --graph 0
} > $O 2>&1
cat $O
