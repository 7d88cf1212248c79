#!/bin/bash
# round 5, GPU call 1: the reference's unchanged network.py on the device (tests + bench both ways), smoke gradient
# diagnosis, weight-shadow A/B, kernel trace of the reference-network step
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r5c1; mkdir -p $O
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0"
( time timeout 900 python -m pytest tests/test_dropin_gpu.py -x -q ) > $O/pytest_dropin.txt 2>&1; tail -5 $O/pytest_dropin.txt
( time timeout 600 python bench.py --steps 30 --warmup 10 $Q --ref-steps 20 --fp32-steps 2 ) > $O/bench_native_with_ref.log 2>&1
grep "^{" $O/bench_native_with_ref.log | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('native', d['value'], d['ms_per_step'], 'host', c['host_enqueue_ms_per_step'])
print('reference_network', json.dumps(c['reference_network']))
print('fp32_mode', json.dumps(c['fp32_mode']))
print('roofline', json.dumps(d['roofline'])[:1500])
"
( time timeout 600 python bench.py --network reference --steps 30 --warmup 10 $Q --ref-steps 0 --fp32-steps 0 ) > $O/bench_reference.log 2>&1
grep "^{" $O/bench_reference.log | tail -n 1 | cut -c1-400
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 8 $Q --ref-steps 0 --fp32-steps 0 --no-kernel-timing > $O/ab_base_$i.log 2>&1; grep -o '"value": [0-9.]*' $O/ab_base_$i.log | head -1
  TSG_WEIGHT_SHADOW=1 timeout 300 python bench.py --steps 20 --warmup 8 $Q --ref-steps 0 --fp32-steps 0 --no-kernel-timing > $O/ab_shadow_$i.log 2>&1; grep -o '"value": [0-9.]*' $O/ab_shadow_$i.log | head -1
done
( time timeout 600 python tools/diag_smoke_grads.py ) > $O/diag_smoke_grads.txt 2>&1; tail -n 80 $O/diag_smoke_grads.txt | cut -c1-160
out=$PWD/gpurun_out/prof_ref; rm -rf $out; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o ref -- python $OLDPWD/bench.py --network reference --steps 10 --warmup 5 $Q --ref-steps 0 --fp32-steps 0 --no-kernel-timing > $out.log 2>&1)
f=$(find $out -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_reference_raw.csv
find $out -name "*.csv" -size +8M -delete
ls -la $O
