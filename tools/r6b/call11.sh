#!/bin/bash
# round 6b, call 11: forked heads as the eager default + the bench's mode probe
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b_call11.txt
{
timeout 1200 python -m pytest tests/test_graph_gpu.py tests/test_dropin_gpu.py tests/test_headline_gpu.py tests/test_wrw_stream_gpu.py -x -q 2>&1 | tail -5
( time timeout 600 python bench.py ) > gpurun_out/r6b_bench_default.log 2>&1
grep "^{" gpurun_out/r6b_bench_default.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['value'], d['ms_per_step'], 'hip_graph', c['hip_graph'], c['mode_probe'])
print('eager_steps', c['eager_steps']); print('host', c['host_enqueue_ms_per_step'])
print('roofline', d['roofline']['kernel'], d['roofline']['frac'], 'timed_region' in d['roofline'])
"
grep real gpurun_out/r6b_bench_default.log
} > $O 2>&1
cat $O
