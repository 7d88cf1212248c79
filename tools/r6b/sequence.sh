#!/bin/bash
# the ordered kernel sequence of ONE replayed step (index, start offset, duration, name) -> gpurun_out/r6b_sequence.txt
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
out=$PWD/gpurun_out/seq; rm -rf $out; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $OLDPWD/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-psa-probe --no-ohem-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --forced-steps 0 "$@" > $out.log 2>&1)
OUT=$out python - <<'PY' > gpurun_out/r6b_sequence.txt
import csv, glob, os, re
f = glob.glob(os.environ["OUT"] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "sgd_multi_k" in r[2]]
best = None
for j in range(1, len(idx)):
    st = rows[idx[j - 1] + 1:idx[j] + 1]
    wall = st[-1][1] - rows[idx[j - 1]][1]
    dsum = sum(e - s for s, e, _ in st)
    if dsum <= wall * 1.002 and (best is None or wall < best[0]):
        best = (wall, j)
a, b = idx[best[1] - 1], idx[best[1]]
step = rows[a + 1:b + 1]
t0 = rows[a][1]
print("step wall %.1f us, %d kernels" % ((step[-1][1] - t0) / 1e3, len(step)))
for i, (s, e, n) in enumerate(step):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    print("%4d  %9.1f  %7.1f  %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, n[:110]))
PY
find $out -name "*.csv" -size +8M -delete
head -3 gpurun_out/r6b_sequence.txt
