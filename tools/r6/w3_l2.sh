#!/bin/bash
# L2 hit rate and memory-side traffic of the weight-gradient kernel per layer (separate --pmc pass, --kernel-trace only)
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_w3l2
rm -rf $out; mkdir -p $out
(cd /tmp && ONLY="${ONLY:-layer}" MIOPEN_LOG_LEVEL=1 timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d $out/run -o pmc -- python $OLDPWD/tools/bench_conv3wrw.py > $out/run.log 2>&1)
grep -E "ours" $out/run.log | sed 's/MIOpen.*//'
OUT=$out/run python - <<'PY'
import csv, glob, collections, os
rows = []
for f in glob.glob(os.environ["OUT"] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "conv3_wrw_gen_k" in row["Kernel_Name"]:
            rows.append(row)
disp = collections.OrderedDict()
for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0][-40:], "grid": r.get("Grid_Size")})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
# consecutive dispatches with the same (name, grid) = one layer of the bench loop
groups = []
for k, d in disp.items():
    key = (d["name"], d["grid"])
    if not groups or groups[-1][0] != key:
        groups.append([key, []])
    groups[-1][1].append(d)
for key, ds in groups:
    n = len(ds)
    avg = lambda c: sum(d.get(c, 0.0) for d in ds) / n
    h, m = avg("TCC_HIT_sum"), avg("TCC_MISS_sum")
    print(key[0], "grid", key[1], "launches", n, "L2 hit rate %.3f  req %.2fM  hit %.2fM miss %.2fM  EA read %.1f MB (64 B per request)" %
          (h / max(h + m, 1), avg("TCC_REQ_sum") / 1e6, h / 1e6, m / 1e6, avg("TCC_EA0_RDREQ_sum") * 64 / 1e6))
PY
find $out -name "*.csv" -size +8M -delete
