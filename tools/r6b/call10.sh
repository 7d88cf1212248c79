#!/bin/bash
# round 6b, call 10: eager step with the auxiliary heads (and the spatial path) on side streams vs plain eager vs graph replay
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b_call10.txt
{
for i in 1 2 3; do
tools/r6/q.sh "graph                          " --
tools/r6/q.sh "eager                          " -- --graph 0
tools/r6/q.sh "eager, heads forked            " TSG_FORK_HEADS=1 -- --graph 0
tools/r6/q.sh "eager, heads + spatial forked  " TSG_FORK_HEADS=1 TSG_FORK_SPATIAL=1 -- --graph 0
done
} > $O 2>&1
cat $O
