#!/bin/bash
# SQ counters of the stem-conv kernels (one pass, 8 SQ slots; no sys/hip tracing)
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_c3
rm -rf $out; mkdir -p $out
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
(cd /tmp && MIOPEN_LOG_LEVEL=1 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $out -o pmc -- python $OLDPWD/tools/bench_conv3wrw.py > $out.log 2>&1)
echo "rc=$?"
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pmc_c3/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "conv3_" not in k: continue
        d = res[k.split("(")[0][-24:]][row["Counter_Name"]]
        d[0] += float(row["Counter_Value"]); d[1] += 1
for k, v in res.items():
    print(k, {c: round(a / max(n, 1)) for c, (a, n) in v.items()}, "launches", max(n for _, n in v.values()))
PY
find $out -name "*.csv" -size +8M -delete
