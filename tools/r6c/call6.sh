#!/bin/bash
# round 6c, call 6: weight gradients under the segmented replay — one block per CU (TSG_CONV_WRW_OCC=1) and fewer blocks (TSG_CONV_WRW_BLOCKS): fewer partials, CUs left to the other queue
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c_call6.txt
Q="--no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --forced-steps 0 --no-kernel-timing"
line() { grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; m=c.get('mode_probe') or {}
print('%-26s %.1f img/s %.3f ms  chosen %s whole %s segmented %s eager %s loss %s fallback %s' % ('$1', d['value'], d['ms_per_step'], m.get('chosen'), m.get('whole_graph_ms_per_step'), m.get('segmented_ms_per_step'), m.get('eager_ms_per_step'), c.get('final_loss'), str(c.get('hip_graph_fallback'))[:200]))
"; }
{
for i in 1 2 3; do
python bench.py $Q 2>/dev/null | line "default"
TSG_CONV_WRW_OCC=1 python bench.py $Q 2>/dev/null | line "occ1"
TSG_CONV_WRW_OCC=1 TSG_CONV_WRW_BLOCKS=192 python bench.py $Q 2>/dev/null | line "occ1 blocks192"
TSG_CONV_WRW_OCC=1 TSG_CONV_WRW_BLOCKS=128 python bench.py $Q 2>/dev/null | line "occ1 blocks128"
TSG_CONV_WRW_BLOCKS=128 python bench.py $Q 2>/dev/null | line "occ2 blocks128 (256 for the big maps)"
done
} > $O 2>&1
cat $O | cut -c1-300
