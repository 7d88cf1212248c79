#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, --kernel-trace only) of the tsg:: kernels of `python $1`
# usage: tools/pmc_traffic_script.sh tools/bench_conv3wrw.py tag
export TMPDIR=/tmp
script=$1; tag=${2:-x}
base=$PWD/gpurun_out/pmct_$tag
rm -rf $base; mkdir -p $base
for c in FETCH_SIZE WRITE_SIZE; do
  out=$base/$c
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o pmc -- python $OLDPWD/$script > $out.log 2>&1)
  echo "$c rc=$?"
done
python tools/pmc_summarize.py $base 2>&1 | cut -c1-260 | head -30
find $base -name "*.csv" -size +8M -delete
