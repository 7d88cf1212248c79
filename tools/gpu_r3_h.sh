#!/bin/bash
# round 3, call H: the N > 1 code path on one rank (forced collectives: SyncBN exchanges + DDP buckets through tsg_comm),
# its reduce-scatter / separate-communicator / mailbox variants, and family traces after the pooling / find-db work
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; ( time env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ohem-probe ) > $O/bench_$name.log 2>&1; echo "$name: $(grep -o '"value": [0-9.]*' $O/bench_$name.log | head -1) tracebacks $(grep -c Traceback $O/bench_$name.log)"; }
run plain A=1
run forced TSG_FORCE_COLLECTIVES=1
run forced_rs TSG_FORCE_COLLECTIVES=1 TSG_DDP_RS=1
run forced_separate TSG_FORCE_COLLECTIVES=1 TSG_DDP_COMM=separate
run forced_torch TSG_FORCE_COLLECTIVES=1 TSG_DDP_COMM=torch TSG_COMM=0
run forced_oneshot TSG_FORCE_COLLECTIVES=1 TSG_XGMI_ONESHOT=1
for c in pspnet dfn psanet; do
  o2=$PWD/gpurun_out/prof_$c; rm -rf $o2; mkdir -p $o2
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o2 -o b -- python $OLDPWD/bench.py --config $c --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > $o2.log 2>&1)
  C=$c python - <<'PY'
import csv, glob, os
c = os.environ["C"]
for f in glob.glob("gpurun_out/prof_%s/**/*kernel_stats.csv" % c, recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
    n = sum(int(r["Calls"]) for r in rows)
    with open("gpurun_out/r3h/kernel_stats_%s.csv" % c, "w") as o:
        o.write("# bench.py --config %s --steps 5 --warmup 3: %.1f ms of kernels over %d launches in 8 steps\n" % (c, tot, n))
        o.write("Name,Calls,TotalUs,AvgUs,Pct\n")
        for r in rows[:60]:
            o.write('"%s",%s,%.1f,%.2f,%s\n' % (r["Name"][:150].replace('"', "'"), r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    print(c, "kernels %.1f ms / 8 steps, %d launches" % (tot, n))
PY
  find $o2 -name "*.csv" -size +4M -delete; find $o2 -name "*.db" -delete
done
