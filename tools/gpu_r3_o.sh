#!/bin/bash
# conv3 weight gradient: x fragments shared between tile rows (TSG_WRW_SHARE=1) against one fragment per MFMA (=0):
# (historical: the TSG_WRW_SHARE switch this script toggles was removed after the comparison; kept as the record of how profiles/r03_conv3wrw_shared_fragments.txt was made)
# parity, headline bench A/B on one box, per-kernel times
cd "$(dirname "$0")/.." || exit 1
O=$PWD/gpurun_out/r3o; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( TSG_WRW_SHARE=0 timeout 600 python -m pytest tests/test_convwrw_gpu.py -x -q -m gpu ) > $O/pytest0.log 2>&1; tail -n 1 $O/pytest0.log
for rep in 1 2; do for sh in 0 1; do
  echo "bench share=$sh: $(TSG_WRW_SHARE=$sh timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-ohem-probe 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"
done; done
for sh in 0 1; do
  out=$O/trace$sh; rm -rf $out; mkdir -p $out
  (cd /tmp && TSG_WRW_SHARE=$sh timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b -- python $R/bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-ohem-probe > $out.log 2>&1)
  f=$(find $out -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_share$sh.csv
  echo "== share=$sh"; grep -E "wrw" "$f" | awk -F'","' '{n=$1; sub(/^"/,"",n); printf "%-70s calls %s avg %s ns\n", substr(n,1,70), $2, $4}'
  find $out -name "*kernel_trace.csv" -delete
done
