#!/bin/bash
# round 6, first GPU call: the ADVICE fixes on the device + eager vs hipGraph replay with the weight-gradient side stream inside the graph
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out; O=gpurun_out/r6_call1.txt; : > $O
( timeout 900 python -m pytest tests/test_wrw_stream_gpu.py tests/test_comm_gpu.py tests/test_graph_gpu.py tests/test_convwrw_gpu.py -x -q 2>&1 | tail -n 15 ) >> $O
for i in 1 2; do
  tools/r6/q.sh "eager                       " -- >> $O
  tools/r6/q.sh "graph2 side-stream in graph " TSG_WRW_IN_GRAPH=1 -- --graph 2 >> $O
  tools/r6/q.sh "graph2 no side stream       " TSG_WRW_IN_GRAPH=0 -- --graph 2 >> $O
  tools/r6/q.sh "graph1 side-stream in graph " TSG_WRW_IN_GRAPH=1 -- --graph 1 >> $O
done
tools/r6/q.sh "eager forced collectives    " TSG_FORCE_COLLECTIVES=1 -- >> $O
tools/r6/q.sh "graph2 forced collectives   " TSG_FORCE_COLLECTIVES=1 TSG_WRW_STREAM=1 -- --graph 2 >> $O
cat $O
