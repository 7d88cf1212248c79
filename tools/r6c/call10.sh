#!/bin/bash
# round 6c, call 10: fusion module + main head + fusion backward as ONE graph (TSG_SEG_MERGE=1) against three (0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c_call10.txt
Q="--no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --forced-steps 0 --no-kernel-timing"
line() { grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; m=c.get('mode_probe') or {}
print('%-10s %.1f img/s %.3f ms  chosen %s whole %s segmented %s eager %s loss %s fallback %s' % ('$1', d['value'], d['ms_per_step'], m.get('chosen'), m.get('whole_graph_ms_per_step'), m.get('segmented_ms_per_step'), m.get('eager_ms_per_step'), c.get('final_loss'), str(c.get('hip_graph_fallback'))[:200]))
"; }
{
python -m pytest -q -x -m gpu "tests/test_graph_gpu.py::test_segmented_replay_follows_the_same_trajectory_as_one_graph" -p no:cacheprovider 2>&1 | tail -3
for i in 1 2 3 4; do
TSG_SEG_MERGE=0 python bench.py $Q 2>/dev/null | line "merge 0"
TSG_SEG_MERGE=1 python bench.py $Q 2>/dev/null | line "merge 1"
done
} > $O 2>&1
cat $O | cut -c1-300
