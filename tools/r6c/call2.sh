#!/bin/bash
# round 6c, call 2: optimizer in two graphs (TSG_SEG_EARLY_OPT) and auxiliary heads started early (TSG_SEG_EARLY_HEADS): tests, then interleaved A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c_call2.txt
{
python -m pytest -q -x -m gpu tests/test_optim_gpu.py "tests/test_graph_gpu.py::test_segmented_replay_follows_the_same_trajectory_as_one_graph" -p no:cacheprovider 2>&1 | tail -15
for i in 1 2 3; do
for cfg in "0 0" "1 0" "0 1" "1 1"; do
set -- $cfg
TSG_SEG_EARLY_OPT=$1 TSG_SEG_EARLY_HEADS=$2 python bench.py --no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --forced-steps 0 --no-kernel-timing 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; m=c['mode_probe']
print('early_opt $1 early_heads $2: %.1f img/s %.3f ms  chosen %-17s whole %.3f  segmented %s  eager %.3f  loss %s fallback %s' % (d['value'], d['ms_per_step'], m['chosen'], m['whole_graph_ms_per_step'], m['segmented_ms_per_step'], m['eager_ms_per_step'], c['final_loss'], str(c['hip_graph_fallback'])[:150]))
"
done
done
} > $O 2>&1
cat $O
