#!/bin/bash
# round 6b, call 14: SpatialPath on auxiliary head 0's stream (no fifth stream), from the start / behind layer1 — eager
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b_call14.txt
{
for i in 1 2 3; do
tools/r6/q.sh "eager (heads forked)                  " -- --graph 0
tools/r6/q.sh "eager + spatial path from the start   " TSG_FORK_SPATIAL=1 -- --graph 0
tools/r6/q.sh "eager + spatial path behind layer1    " TSG_FORK_SPATIAL=2 -- --graph 0
done
} > $O 2>&1
cat $O
