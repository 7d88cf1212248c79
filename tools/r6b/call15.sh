#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b_call15.txt
{
for i in 1 2 3; do
tools/r6/q.sh "eager + spatial path behind layer1    " TSG_FORK_SPATIAL=2 -- --graph 0
tools/r6/q.sh "eager + spatial path behind layer2    " TSG_FORK_SPATIAL=3 -- --graph 0
done
} > $O 2>&1
cat $O
