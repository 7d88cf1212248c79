#!/bin/bash
# fold kernel A/B by rocprofv3 kernel trace (HIP-event timing of the microbench is host-bound for the short kernels)
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_convwrw_gpu.py -x -q 2>&1 | tail -2
for f in 1 2; do
  out=$PWD/gpurun_out/fold_$f; rm -rf $out; mkdir -p $out
  (cd /tmp && TSG_CONV_WRW_FOLD=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $OLDPWD/tools/bench_conv3wrw.py > $out.log 2>&1)
  echo "== TSG_CONV_WRW_FOLD=$f"; grep "per step" $out.log
  OUT=$out python - <<'PY'
import csv, glob, os, collections
for f in glob.glob(os.environ["OUT"] + "/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "conv3_wrw_gen" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a fold follows its gen kernel: group folds by the grid of the preceding gen kernel
    agg = collections.OrderedDict()
    prev = None
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if "fold" in r["Kernel_Name"]:
            k = (prev, r["Kernel_Name"].split("(")[0][-24:], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X"))
            agg.setdefault(k, []).append(d)
        else:
            prev = r["Kernel_Name"].split("(")[0][-34:] + " grid " + str(r.get("Grid_Size") or r.get("Grid_Size_X"))
    for k, v in agg.items():
        print("   after %-60s %s grid %s: %d launches, avg %.1f us" % (k[0], k[1], k[2], len(v), sum(v) / len(v)))
PY
  find $out -name "*.csv" -size +8M -delete
done
