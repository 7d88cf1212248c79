#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command (no counters, no sys tracing)
#   tools/prof_bench.sh [config]      config: bisenet (default) | pspnet | dfn | psanet -> gpurun_out/prof[_config]/kernel_stats_compact.csv
export TMPDIR=/tmp
CFG=${1:-bisenet}
SUF=""; [ "$CFG" != "bisenet" ] && SUF="_$CFG"
out=$PWD/gpurun_out/prof$SUF
rm -rf $out; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python $OLDPWD/bench.py --config $CFG --steps 10 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-psa-probe --no-ohem-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --forced-steps 0 > $out.log 2>&1)
echo "rc=$?"; tail -1 $out.log | cut -c1-200
PROF_OUT=$out python - <<'PY'
import csv, glob, collections
import os
OUT = os.environ["PROF_OUT"]
f = glob.glob(OUT + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 10 steps: find the span of the last 10 occurrences of the SGD kernel boundaries via time: use last 40% of the trace
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = r["Kernel_Name"]
    a = agg[k]; a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
with open(OUT + "/kernel_stats_compact.csv", "w") as o:
    o.write("Name,Calls,TotalUs,AvgUs,Pct\n")
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write('"%s",%d,%.1f,%.2f,%.2f\n' % (k[:160].replace('"', "'"), n, us, us / n, 100 * us / tot))
print("kernels total %.1f ms over %d launches; trace span %.1f ms" % (tot / 1e3, len(rows), (t1 - t0) / 1e6))
PY
find $out -name "*kernel_trace.csv" -size +6M -delete; find $out -name "*.db" -delete
head -50 $out/kernel_stats_compact.csv | cut -c1-150
