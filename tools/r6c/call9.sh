#!/bin/bash
# round 6c, call 9: the default bench line three times (roofline from three instrumented warm-up steps), family config once
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c_call9.txt
{
for i in 1 2 3; do
( time timeout 600 python bench.py ) > gpurun_out/r6c_bench_$i.log 2>&1
grep "^{" gpurun_out/r6c_bench_$i.log | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.1f img/s %.3f ms | roofline %s bound %s launches %d total_ms %.3f frac %.4f traffic %s | %s' % (d['value'], d['ms_per_step'], r['kernel'], r['bound'], r['launches'], r['total_ms'], r['frac'], r['traffic'], [(t['kernel'], t['launches'], t['total_ms'], t['frac']) for t in r['top_families']]))
print('   ', r.get('measured_over'), '| timed_region', {k: r['timed_region'][k] for k in ('kernel','launches','total_ms','frac')} if r.get('timed_region') else None)
"
grep real gpurun_out/r6c_bench_$i.log
done
timeout 600 python bench.py --config psanet --steps 10 --warmup 6 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('psanet %.1f img/s | roofline %s %d launches %.3f ms frac %.4f' % (d['value'], r['kernel'], r['launches'], r['total_ms'], r['frac']))
"
} > $O 2>&1
cat $O | cut -c1-400
