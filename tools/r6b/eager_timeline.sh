#!/bin/bash
# eager steps (weight gradients + auxiliary heads on side streams): wall, sum of kernel durations, union busy time, and who overlaps whom
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
out=$PWD/gpurun_out/etl; rm -rf $out; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $OLDPWD/bench.py --graph 0 --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-psa-probe --no-ohem-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --forced-steps 0 "$@" > $out.log 2>&1)
grep -o '"value": [0-9.]*' $out.log | head -1
OUT=$out python - <<'PY' > gpurun_out/r6b_eager_timeline.txt
import csv, glob, os, re, collections
f = glob.glob(os.environ["OUT"] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "sgd_multi_k" in r[2]]
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n); return n[:60]
for j in range(len(idx) - 3, len(idx)):
    st = rows[idx[j - 1] + 1:idx[j] + 1]
    t0 = rows[idx[j - 1]][1]
    wall = (st[-1][1] - t0) / 1e3
    dsum = sum(e - s for s, e, *_ in st) / 1e3
    # union
    ev = sorted((s, e) for s, e, *_ in st)
    busy, cs, ce = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > ce:
            busy += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print("step %d: wall %.1f us, kernels %d, sum of durations %.1f us, union busy %.1f us, idle %.1f us, overlapped %.1f us" % (j, wall, len(st), dsum, busy / 1e3, wall - busy / 1e3, dsum - busy / 1e3))
st = rows[idx[-2] + 1:idx[-1] + 1]
qs = collections.Counter(r[3] for r in st)
print("queues:", dict(qs))
# per kernel family: time spent overlapped with another kernel
ov = collections.defaultdict(float); tot = collections.defaultdict(float)
for i, (s, e, n, q, _) in enumerate(st):
    o = 0
    for s2, e2, n2, q2, _ in st:
        if q2 != q and s2 < e and e2 > s:
            o += min(e, e2) - max(s, s2)
    ov[short(n)] += min(o, e - s) / 1e3; tot[short(n)] += (e - s) / 1e3
for k in sorted(tot, key=lambda k: -tot[k])[:40]:
    print("%-62s total %8.1f us  overlapped with another queue %8.1f us" % (k, tot[k], ov[k]))
PY
find $out -name "*.csv" -size +8M -delete
head -8 gpurun_out/r6b_eager_timeline.txt
