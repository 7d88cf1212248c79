#!/bin/bash
# kernel trace of the N > 1 code path on one rank (TSG_FORCE_COLLECTIVES=1) beside the plain step: which launches the path adds
export TMPDIR=/tmp
for mode in plain forced; do
  out=$PWD/gpurun_out/prof_$mode; rm -rf $out; mkdir -p $out
  if [ $mode = forced ]; then export TSG_FORCE_COLLECTIVES=1; else unset TSG_FORCE_COLLECTIVES; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-psa-probe --no-ohem-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 > $out.log 2>&1)
  tail -1 $out.log | cut -c1-120
  PROF_OUT=$out python - <<'PY'
import csv, glob, collections, os
OUT = os.environ["PROF_OUT"]
f = glob.glob(OUT + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    a = agg[r["Kernel_Name"]]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
with open(OUT + "_stats.csv", "w") as o:
    o.write("Name,Calls,TotalUs,AvgUs\n")
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write('"%s",%d,%.1f,%.2f\n' % (k[:140].replace('"', "'"), n, us, us / n))
print("kernels total %.1f ms over %d launches" % (sum(v[1] for v in agg.values()) / 1e3, len(rows)))
PY
  rm -rf $out
done
