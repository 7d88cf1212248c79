#!/bin/bash
# round 3 evidence: default bench, kernel-trace stats, PMC traffic + MFMA busy of the bench command, configs 3-5 (bench line
# with cpu_baseline + kernel stats), the tests that the -x run stopped before
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/final3; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_metric_gpu.py tests/test_pool_gpu.py tests/test_shadow_gpu.py tests/test_stemconv_gpu.py tests/test_integration_doc.py -x -q -m gpu ) > $O/pytest_rest.log 2>&1; tail -n 3 $O/pytest_rest.log
( time timeout 600 python bench.py ) > $O/bench_default.log 2>&1; tail -n 4 $O/bench_default.log | cut -c1-300
bash tools/prof_bench.sh > $O/prof_bench.out 2>&1; cp gpurun_out/prof/kernel_stats_compact.csv $O/kernel_stats.csv; grep "kernels total" $O/prof_bench.out
bash tools/pmc_traffic.sh > $O/pmc_traffic.out 2>&1; cp gpurun_out/pmc/summary.txt $O/pmc_summary.txt; cp gpurun_out/pmc/traffic_by_kernel.json $O/traffic_by_kernel.json
out=$PWD/gpurun_out/pmc_busy; rm -rf $out; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-ohem-probe > $out.log 2>&1)
python tools/pmc_mfma_busy.py $out $O/mfma_busy.json
find $out -name "*.csv" -size +8M -delete
python tools/make_traffic_json.py $O/traffic_by_kernel.json $O/traffic.json r03 $O/mfma_busy.json > /dev/null
for c in pspnet dfn psanet; do
  ( time timeout 900 python bench.py --config $c --steps 20 --warmup 10 ) > $O/bench_$c.log 2>&1
  grep -o '"value": [0-9.]*' $O/bench_$c.log | head -2
  o2=$PWD/gpurun_out/prof_$c; rm -rf $o2; mkdir -p $o2
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o2 -o b -- python $OLDPWD/bench.py --config $c --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > $o2.log 2>&1)
  C=$c python - <<'PY'
import csv, glob, os
c = os.environ["C"]
for f in glob.glob("gpurun_out/prof_%s/**/*kernel_stats.csv" % c, recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("gpurun_out/final3/kernel_stats_%s.csv" % c, "w") as o:
        o.write("Name,Calls,TotalUs,AvgUs,Pct\n")
        for r in rows[:60]:
            o.write('"%s",%s,%.1f,%.2f,%s\n' % (r["Name"][:150].replace('"', "'"), r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  find $o2 -name "*.csv" -size +4M -delete; find $o2 -name "*.db" -delete
done
ls -la $O
