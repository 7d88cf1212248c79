#!/bin/bash
# the segmented replay as a timeline: per step wall / sum of kernel durations / union busy / idle (no kernel on any queue), and for the fastest step the largest idle gaps
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
out=$PWD/gpurun_out/segtl; rm -rf $out; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $OLDPWD/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-psa-probe --no-ohem-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --forced-steps 0 "$@" > $out.log 2>&1)
grep -o '"value": [0-9.]*' $out.log | head -1
grep -o '"mode_probe": {[^}]*}' $out.log | cut -c1-300
OUT=$out python - <<'PY' > gpurun_out/r6c_segmented_timeline.txt
import csv, glob, os, re, collections
f = glob.glob(os.environ["OUT"] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "sgd_multi_k" in r[2]]
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n); return n[:56]
steps = []
for j in range(1, len(idx)):
    st = rows[idx[j - 1] + 1:idx[j] + 1]
    t0 = rows[idx[j - 1]][1]
    wall = (st[-1][1] - t0) / 1e3
    dsum = sum(e - s for s, e, *_ in st) / 1e3
    ev = sorted((s, e) for s, e, *_ in st)
    busy, cs, ce = 0, ev[0][0], ev[0][1]
    gaps = [(ev[0][0] - t0, ev[0])]
    for s, e in ev[1:]:
        if s > ce:
            busy += ce - cs; gaps.append((s - ce, (s, e))); cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    nq = len(set(r[3] for r in st))
    steps.append((wall, j, len(st), dsum, busy / 1e3, nq, gaps, st))
for wall, j, n, dsum, busy, nq, _, _ in steps:
    print("step %2d: wall %8.1f us, kernels %d on %d queues, sum of durations %8.1f, union busy %8.1f, idle %6.1f, overlapped %7.1f" % (j, wall, n, nq, dsum, busy, wall - busy, dsum - busy))
multi = [s for s in steps if s[5] >= 3 and s[2] < 500]
best = min(multi or steps)
print("\nfastest multi-queue step: %d (wall %.1f us)" % (best[1], best[0]))
st = best[7]
byq = collections.defaultdict(list)
for s, e, n, q in st: byq[q].append((s, e, n))
for q, l in byq.items():
    print("queue %s: %d kernels, %.1f us busy, first %s, last %s" % (q, len(l), sum(e - s for s, e, _ in l) / 1e3, short(l[0][2]), short(l[-1][2])))
print("-- idle gaps > 2 us (no kernel on any queue), with the kernel that ends the gap")
name_at = {(s, e): n for s, e, n, q in st}
tot = 0
for g, (s, e) in sorted(best[6], reverse=True)[:25]:
    if g > 2000:
        tot += g
        print("  %7.1f us before %s" % (g / 1e3, short(name_at.get((s, e), "?"))))
print("sum of those gaps: %.1f us" % (tot / 1e3))
# concurrency profile: time with 1, 2, 3+ kernels in flight
evs = []
for s, e, n, q in st: evs += [(s, 1), (e, -1)]
evs.sort()
cur, last, hist = 0, evs[0][0], collections.Counter()
for t, d in evs:
    hist[cur] += t - last; last = t; cur += d
print("time with k kernels in flight:", {k: round(v / 1e3, 1) for k, v in sorted(hist.items())})
PY
find $out -name "*.csv" -size +8M -delete
cat gpurun_out/r6c_segmented_timeline.txt
