#!/bin/bash
# SQ counters (one pass, 8 SQ slots, --kernel-trace only) of the kernels whose name contains $2, over `python $1` run with
# the environment of the caller.  usage: tools/pmc_kernel.sh tools/bench_conv3g.py conv3g_fwd_k [tag]
export TMPDIR=/tmp
script=$1; pat=$2; tag=${3:-pmc}
out=$PWD/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
C=${PMC_C:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"}
(cd /tmp && MIOPEN_LOG_LEVEL=1 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $out -o pmc -- python $OLDPWD/$script > $out.log 2>&1)
echo "rc=$?"
PAT=$pat OUT=$out python - <<'PY'
import csv, glob, collections, os
res = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.environ["OUT"] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if os.environ["PAT"] not in k: continue
        d = res[k.split("(")[0][-40:]][row["Counter_Name"]]
        d[0] += float(row["Counter_Value"]); d[1] += 1
for k, v in res.items():
    m = {c: a / max(n, 1) for c, (a, n) in v.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    print(k, "launches", max(n for _, n in v.values()))
    print("   ", {c: round(x) for c, x in m.items()})
    print("    of wave cycles: WAIT_ANY %.3f  WAIT_INST_ANY %.3f (LDS %.3f)  ACTIVE_INST_ANY %.3f | MFMA busy / (4 x busy cycles) %.3f | LDS conflict cycles %.0f"
          % (m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_WAIT_INST_LDS", 0) / wc,
             m.get("SQ_ACTIVE_INST_ANY", 0) / wc, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4.0 * (m.get("SQ_BUSY_CYCLES", 0) or 1)),
             m.get("SQ_LDS_BANK_CONFLICT", 0)))
PY
find $out -name "*.csv" -size +8M -delete
