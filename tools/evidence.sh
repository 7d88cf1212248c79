#!/bin/bash
# The evidence run of a round (what profiles/rNN_final_* are made from), at the tree it is called from:
#   default bench line, rocprofv3 kernel-trace stats of the bench command, PMC traffic (FETCH_SIZE / WRITE_SIZE, separate
#   passes) + MFMA-busy of the bench command -> traffic.json stamped with the round tag and the commit, optionally the
#   three family configs (bench line WITH cpu_baseline + kernel stats).
# usage: tools/evidence.sh <tag e.g. r04> [families]      (run through gpurun from the repo root; writes gpurun_out/<tag>_evidence)
cd "$(dirname "$0")/.." || exit 1
TAG=${1:-r04}; FAM=${2:-}
O=gpurun_out/${TAG}_evidence; mkdir -p $O
export TMPDIR=/tmp
HEAD=$(cat .git_head 2>/dev/null || echo unknown)
( time timeout 600 python bench.py ) > $O/bench_default.log 2>&1; grep "^{" $O/bench_default.log | tail -n 1 | cut -c1-300
bash tools/prof_bench.sh > $O/prof_bench.out 2>&1; cp gpurun_out/prof/kernel_stats_compact.csv $O/kernel_stats.csv; grep "kernels total" $O/prof_bench.out
# the same with the weight gradients on the compute stream: per-kernel durations that do not contain a concurrent kernel
TSG_WRW_STREAM=0 TSG_FORK_HEADS=0 TSG_FORK_SPATIAL=0 TSG_SEGMENTED_GRAPH=0 bash tools/prof_bench.sh > $O/prof_bench_serial.out 2>&1; cp gpurun_out/prof/kernel_stats_compact.csv $O/kernel_stats_serial.csv; grep "kernels total" $O/prof_bench_serial.out
export TSG_WRW_STREAM=0 TSG_FORK_HEADS=0 TSG_FORK_SPATIAL=0 TSG_SEGMENTED_GRAPH=0     # counter passes: one kernel at a time, or FETCH_SIZE / WRITE_SIZE / MFMA-busy of a kernel contain its neighbour's
bash tools/pmc_traffic.sh > $O/pmc_traffic.out 2>&1; cp gpurun_out/pmc/summary.txt $O/pmc_summary.txt; cp gpurun_out/pmc/traffic_by_kernel.json $O/traffic_by_kernel.json
out=$PWD/gpurun_out/pmc_busy; rm -rf $out; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-ohem-probe --no-psa-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --forced-steps 0 > $out.log 2>&1)
python tools/pmc_mfma_busy.py $out $O/mfma_busy.json
find $out -name "*.csv" -size +8M -delete
python tools/make_traffic_json.py $O/traffic_by_kernel.json $O/traffic.json "$TAG@$HEAD" $O/mfma_busy.json > /dev/null
unset TSG_WRW_STREAM TSG_FORK_HEADS TSG_FORK_SPATIAL TSG_SEGMENTED_GRAPH
if [ -n "$FAM" ]; then
for c in pspnet dfn psanet; do
  ( time timeout 900 python bench.py --config $c --steps 20 --warmup 10 ) > $O/bench_$c.log 2>&1
  grep "^{" $O/bench_$c.log | tail -n 1 > $O/bench_$c.json; grep -o '"value": [0-9.]*' $O/bench_$c.log | head -2
  bash tools/prof_bench.sh $c > $O/prof_bench_$c.out 2>&1; cp gpurun_out/prof_$c/kernel_stats_compact.csv $O/kernel_stats_$c.csv
done
fi
ls -la $O
