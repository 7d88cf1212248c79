#!/bin/bash
# kernel timeline of the N > 1 code path on one rank (TSG_FORCE_COLLECTIVES=1) and of the plain step: keeps the trace CSVs
export TMPDIR=/tmp
for mode in plain forced; do
  out=$PWD/gpurun_out/trace_$mode; rm -rf $out; mkdir -p $out
  if [ $mode = forced ]; then export TSG_FORCE_COLLECTIVES=1; else unset TSG_FORCE_COLLECTIVES; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o bench -- python $OLDPWD/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-psa-probe --no-ohem-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 > $out.log 2>&1)
  tail -1 $out.log | cut -c1-120
  f=$(find $out -name "*kernel_trace.csv" | head -1); cp $f $PWD/gpurun_out/trace_$mode.csv; rm -rf $out
done
ls -la gpurun_out/trace_*.csv
