#!/bin/bash
# round 6c, call 1: the optimizer in two graphs (TSG_SEG_EARLY_OPT) — tests, then interleaved A/B of the segmented replay
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c_call1.txt
{
python -m pytest -q -x -m gpu tests/test_optim_gpu.py tests/test_wrw_stream_gpu.py "tests/test_graph_gpu.py::test_segmented_replay_follows_the_same_trajectory_as_one_graph" -p no:cacheprovider 2>&1 | tail -15
for i in 1 2 3; do
for e in 0 1; do
TSG_SEG_EARLY_OPT=$e python bench.py --no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --forced-steps 0 --no-kernel-timing 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; m=c['mode_probe']
print('early_opt $e: %.1f img/s %.3f ms  chosen %-17s whole %.3f  segmented %s  eager %.3f  fallback %s' % (d['value'], d['ms_per_step'], m['chosen'], m['whole_graph_ms_per_step'], m['segmented_ms_per_step'], m['eager_ms_per_step'], c['hip_graph_fallback']))
"
done
done
} > $O 2>&1
cat $O
