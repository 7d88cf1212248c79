#!/bin/bash
# how much of the step do kernels of different streams run at the same time?  (kernel trace of the default bench)
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_ov; rm -rf $out; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o b -- python $OLDPWD/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-psa-probe --no-ohem-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 > $out.log 2>&1)
python - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(f))]
rows.sort()
# last 3 steps: split on sgd_multi_k
ends = [e for s, e, n, q in rows if "sgd_multi_k" in n]
t0, t1 = ends[-4], ends[-1]
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
tot = sum(e - s for s, e, n, q in sel)
# union
cur_s, cur_e, uni = None, None, 0
for s, e, n, q in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: uni += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
uni += cur_e - cur_s
print("3 steps: span %.2f ms/step, sum of kernel durations %.2f ms/step, union (GPU busy) %.2f ms/step, overlapped %.2f ms/step" % ((t1 - t0) / 3e6, tot / 3e6, uni / 3e6, (tot - uni) / 3e6))
qs = collections.Counter(q for s, e, n, q in sel)
print("queues:", dict(qs))
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in sel:
    if "wrw" in n: a = agg[n[:48]]; a[0] += 1; a[1] += (e - s) / 1e3
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("  %5.1f x %7.1f us  %s" % (n / 3, us / n, k))
PY
find $out -name "*.csv" -size +4M -delete
