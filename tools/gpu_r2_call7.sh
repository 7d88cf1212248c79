#!/bin/bash
O=gpurun_out/c7; mkdir -p $O
timeout 300 python -m pytest tests/test_psa_gpu.py -q -m gpu -x > $O/pytest_psa.log 2>&1; echo "== psa pytest rc=$?"; tail -2 $O/pytest_psa.log
for e in "TSG_PSA_CFG=128x1" "TSG_PSA_CFG=256x1" "TSG_PSA_CFG=256x2" "TSG_PSA_CFG=128x2"; do
  echo "== PSA $e"; env $e PSA_QUICK=1 timeout 120 python tools/bench_psa.py 2>&1 | grep bfloat16 | cut -c1-200
done | tee $O/bench_psa_cfgs.log
TSG_PSA_CFG=256x1 timeout 200 python -m pytest tests/test_psa_gpu.py -q -m gpu -k "bf16" > $O/pytest_psa_256.log 2>&1; tail -1 $O/pytest_psa_256.log
(export TMPDIR=/tmp; out=$PWD/$O/pmc_psa; mkdir -p $out; cd /tmp && PSA_QUICK=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out -o pmc -- python $OLDPWD/tools/bench_psa.py > $out.log 2>&1)
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/c7/pmc_psa/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "psa_mm" not in k: continue
        d = res[k.split("(")[0][-40:]][row["Counter_Name"]]
        d[0] += float(row["Counter_Value"]); d[1] += 1
for k, v in res.items():
    print(k, {c: round(a / max(n, 1)) for c, (a, n) in v.items()})
PY
find $O/pmc_psa -name "*.csv" -size +4M -delete
traj() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 50 --trace-loss "$@" > $O/traj_$name.out 2> $O/traj_$name.err; 
  echo "== $name: $(grep -c '^step' $O/traj_$name.err) steps; losses: $(grep '^step' $O/traj_$name.err | awk 'NR%5==0{printf "%.3f ", $4}' | cut -c1-400)"; tail -1 $O/traj_$name.out | cut -c1-160; }
traj eager_fused --warmup 10 --optimizer fused
traj graph1_fused --warmup 8 --graph 1 --optimizer fused
traj graph1_torch --warmup 8 --graph 1 --optimizer torch
traj graph2_fused --warmup 8 --graph 2 --optimizer fused
timeout 900 python -m pytest tests/test_headline_gpu.py tests/test_graph_gpu.py -q -m gpu -s -k "1024 or graph" > $O/pytest_headline.log 2>&1; echo "== headline+graph rc=$?"; grep -E "passed|failed|^head|headline|^E  |^eager|^graph" $O/pytest_headline.log | cut -c1-250 | head -30
