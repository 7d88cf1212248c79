#!/bin/bash
# round 3, call D: PSA wave specialisation, wgrad XCD mapping (+ its traffic), normalise-on-load for the general conv,
# the k-th-branch head under the kernel trace, bench
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_conv3g_gpu.py tests/test_bnconv_gpu.py tests/test_convwrw_gpu.py tests/test_psa_gpu.py tests/test_upsample_gpu.py tests/test_optim_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1
for cfg in split128x128x1 split128x128x2 split128x64x1 split256x64x1 128x1; do
  ( TSG_PSA_CFG=$cfg PSA_QUICK=1 timeout 200 python tools/bench_psa.py ) > $O/psa_$cfg.log 2>&1
done
PSA_QUICK=1 bash tools/pmc_kernel.sh tools/bench_psa.py psa_mm psa2 > $O/pmc_psa_split.txt 2>&1
( timeout 300 python tools/bench_conv3wrw.py ) > $O/conv3wrw.log 2>&1
ONLY=layer2 bash tools/pmc_traffic_script.sh tools/bench_conv3wrw.py wrw_l2 > $O/traffic_wrw_layer2.txt 2>&1
ONLY=layer3 bash tools/pmc_traffic_script.sh tools/bench_conv3wrw.py wrw_l3 > $O/traffic_wrw_layer3.txt 2>&1
(cd /tmp && REPS=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/kth_trace -o kth -- python $OLDPWD/tools/bench_head_kth.py > $OLDPWD/$O/kth.log 2>&1)
python - <<'PY' > $O/kth_stats.txt 2>&1
import csv, glob
for f in glob.glob("gpurun_out/kth_trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
find gpurun_out/kth_trace -name "*.csv" -size +4M -delete
( time timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ohem-probe ) > $O/bench_bisenet.log 2>&1
( time TSG_CONV_GEN_BN_ON_LOAD=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ohem-probe ) > $O/bench_bisenet_noaff.log 2>&1
( time timeout 300 python bench.py --config psanet --steps 10 --warmup 5 --no-cpu-baseline ) > $O/bench_psanet.log 2>&1
tail -n 5 $O/pytest.log
for f in $O/psa_*.log; do echo "== $f"; grep -v amdgpu.ids $f | cut -c1-260; done
cat $O/pmc_psa_split.txt
grep -v amdgpu.ids $O/conv3wrw.log
cat $O/traffic_wrw_layer2.txt $O/traffic_wrw_layer3.txt
cat $O/kth_stats.txt; tail -n 2 $O/kth.log | cut -c1-400
for f in $O/bench_*.log; do echo "== $f"; grep -o '"value": [0-9.]*' $f | head -1; grep -c Traceback $f; done
