#!/bin/bash
# round 6b, call 13: mode probe with the segmented replay among the candidates (three runs of the default bench, quick flags)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b_call13.txt
{
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --forced-steps 0 --no-kernel-timing 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; m=c['mode_probe']
print('%.1f img/s %.3f ms  chosen %-17s whole %.3f  segmented %s  eager %.3f  fallback %s' % (d['value'], d['ms_per_step'], m['chosen'], m['whole_graph_ms_per_step'], m['segmented_ms_per_step'], m['eager_ms_per_step'], c['hip_graph_fallback']))
"
done
} > $O 2>&1
cat $O
