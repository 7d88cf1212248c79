#!/bin/bash
# HBM traffic of our kernels from PMC counters (separate passes, no sys/hip tracing):
#   pass 1: FETCH_SIZE   pass 2: WRITE_SIZE    -> gpurun_out/pmc/{fetch,write}
export TMPDIR=/tmp
base=$PWD/gpurun_out/pmc
rm -rf $base; mkdir -p $base
for c in FETCH_SIZE WRITE_SIZE; do
  out=$base/$c
  (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-psa-probe --no-ohem-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --forced-steps 0 > $out.log 2>&1)
  echo "$c rc=$?"; find $out -name "*.csv" | head -5
done
python tools/pmc_summarize.py $base > $base/summary.txt 2>&1; head -40 $base/summary.txt
find $base -name "*.csv" -size +8M -delete
