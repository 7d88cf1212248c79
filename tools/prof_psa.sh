#!/bin/bash
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_psa
rm -rf $out; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o psa -- python $OLDPWD/tools/bench_psa.py > $out.log 2>&1)
f=$(find $out -name "*kernel_stats.csv" | head -1); head -30 "$f" | cut -c1-220
find $out -name "*kernel_trace.csv" -size +8M -delete
