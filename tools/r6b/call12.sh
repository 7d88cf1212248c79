#!/bin/bash
# round 6b, call 12: the replayed step as five linear graphs (heads side by side)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b_call12.txt
{
( time timeout 600 python bench.py --no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --forced-steps 0 ) > gpurun_out/r6b_bench_seg.log 2>&1
tail -3 gpurun_out/r6b_bench_seg.log | cut -c1-300
grep "^{" gpurun_out/r6b_bench_seg.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(d['value'], d['ms_per_step'], 'hip_graph', c['hip_graph'], c['hip_graph_fallback']); print(c['mode_probe'])
print('loss', c['final_loss'])
"
} > $O 2>&1
cat $O
