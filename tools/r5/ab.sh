#!/bin/bash
# interleaved A/B of one environment switch on the default bench: tools/r5/ab.sh VAR A_VALUE B_VALUE [pairs]
cd "$(dirname "$0")/../.." || exit 1
V=$1; A=$2; B=$3; N=${4:-2}
Q="--no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --no-kernel-timing --steps 30 --warmup 10"
for i in $(seq $N); do
  for val in $A $B; do
    r=$(env $V=$val python bench.py $Q 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
    h=$(env $V=$val python bench.py $Q 2>/dev/null | grep -o '"host_enqueue_ms_per_step": [0-9.]*' | head -1)
    echo "$V=$val  $r  $h"
  done
done
