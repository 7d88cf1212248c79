#!/bin/bash
# round 6c, call 3: env A/Bs under the segmented replay (normalise-on-load for the general 3x3 layers; forced collectives under one graph), then the full GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c_call3.txt
Q="--no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --forced-steps 0 --no-kernel-timing"
line() { grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; m=c.get('mode_probe') or {}
print('$1: %.1f img/s %.3f ms  chosen %s whole %s segmented %s eager %s graph %s fallback %s' % (d['value'], d['ms_per_step'], m.get('chosen'), m.get('whole_graph_ms_per_step'), m.get('segmented_ms_per_step'), m.get('eager_ms_per_step'), c.get('hip_graph'), str(c.get('hip_graph_fallback'))[:200]))
"; }
{
for i in 1 2; do
python bench.py $Q 2>/dev/null | line "default"
TSG_CONV_GEN_BN_ON_LOAD=1 python bench.py $Q 2>/dev/null | line "gen_bn_on_load"
done
TSG_FORCE_COLLECTIVES=1 python bench.py $Q 2>/dev/null | line "forced eager"
TSG_FORCE_COLLECTIVES=1 timeout 300 python bench.py $Q --graph 2 2>&1 | tail -n 3 | line "forced graph2"
python -m pytest tests -m gpu -q -rf -p no:cacheprovider 2>&1 | tail -n 40
} > $O 2>&1
cat $O | cut -c1-300
