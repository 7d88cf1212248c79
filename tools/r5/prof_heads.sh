#!/bin/bash
# kernel-trace of the head microbench for both forward forms
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
for f in 1 2; do
  out=$PWD/gpurun_out/prof_heads_$f; rm -rf $out; mkdir -p $out
  (cd /tmp && TSG_HEAD_FWD=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o h -- python $OLDPWD/tools/r5/bench_heads.py > $out.log 2>&1)
  python - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[r["Kernel_Name"][:70]]; a[0] += 1; a[1] += d
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%6d x %8.1f us  %s" % (n, us / n, k))
PY
  find $out -name "*.csv" -size +4M -delete
done
