#!/bin/bash
# PSA per-kernel breakdown (rocprofv3 kernel trace) against the contraction length
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=$PWD
for c in ${CFGS:-af256x64x2}; do for k in 512 1792 3600; do
  out=$R/gpurun_out/r3l/$c.$k; rm -rf $out; mkdir -p $out
  (cd /tmp && PSA_K=$k TSG_PSA_CFG=$c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o psa -- python $R/tools/bench_psa_k.py > $out.log 2>&1)
  f=$(find $out -name "*kernel_stats.csv" | head -1); echo "== $c K=$k"; grep -E "psa_" "$f" | awk -F'","' '{printf "%-90s calls %s avg %s ns\n", substr($1,2,90), $2, $4}'
  find $out -name "*kernel_trace.csv" -delete
done; done
