#!/bin/bash
# per-kernel difference of one replayed step between two settings of an environment switch: tools/r6/tl_ab.sh VAR A B
cd "$(dirname "$0")/../.." || exit 1
V=$1; A=$2; B=$3
env $V=$A TOPN=0 bash tools/r6/timeline.sh > gpurun_out/tl_a.txt 2>&1; cp gpurun_out/timeline/step_kernels.csv gpurun_out/tl_a.csv
env $V=$B TOPN=0 bash tools/r6/timeline.sh > gpurun_out/tl_b.txt 2>&1; cp gpurun_out/timeline/step_kernels.csv gpurun_out/tl_b.csv
grep "step wall" gpurun_out/tl_a.txt gpurun_out/tl_b.txt
python - <<'PY'
import csv
a = {r["Name"]: (int(r["Calls"]), float(r["TotalUs"])) for r in csv.DictReader(open("gpurun_out/tl_a.csv"))}
b = {r["Name"]: (int(r["Calls"]), float(r["TotalUs"])) for r in csv.DictReader(open("gpurun_out/tl_b.csv"))}
rows = []
for k in set(a) | set(b):
    ca, ua = a.get(k, (0, 0.0)); cb, ub = b.get(k, (0, 0.0))
    rows.append((ub - ua, k, ca, ua, cb, ub))
rows.sort()
print("total A %.1f  B %.1f" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
for d, k, ca, ua, cb, ub in rows[:14] + rows[-14:]:
    print("%+8.1f us  %-64s  A x%-3d %8.1f   B x%-3d %8.1f" % (d, k, ca, ua, cb, ub))
PY
