#!/bin/bash
# round 3, final tree: kernel-trace stats and MFMA-busy counters of the bench command, psanet config line
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/final3b; mkdir -p $O
export TMPDIR=/tmp
bash tools/prof_bench.sh > $O/prof_bench.out 2>&1; cp gpurun_out/prof/kernel_stats_compact.csv $O/kernel_stats.csv; grep "kernels total" $O/prof_bench.out
out=$PWD/gpurun_out/pmc_busy; rm -rf $out; mkdir -p $out
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-ohem-probe > $out.log 2>&1)
python tools/pmc_mfma_busy.py $out $O/mfma_busy.json
find $out -name "*.csv" -size +8M -delete
( timeout 200 python bench.py --config psanet --steps 20 --warmup 10 --no-cpu-baseline ) > $O/bench_psanet.log 2>&1; grep -o '"value": [0-9.]*' $O/bench_psanet.log | head -1
