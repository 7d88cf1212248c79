#!/bin/bash
# round 4, call A: the new parity tests, the default bench line at HEAD (psa_probe, int64 labels, oracle beside the k-th
# branch), conv64-vs-conv3g A/B for the 64 -> 64 stride-1 layers, SQ counters of the fused heads, fp64-truth diagnostic
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest "tests/test_fused_head_gpu.py" "tests/test_conv3g_gpu.py::test_bench_geometry_forward_stats_dgrad_addend" tests/test_ohem_gpu.py -x -q -m gpu -s ) > $O/pytest_new.log 2>&1; tail -n 4 $O/pytest_new.log; grep "^head " $O/pytest_new.log
( time timeout 600 python bench.py ) > $O/bench_default.log 2>&1; tail -n 1 $O/bench_default.log > $O/bench_default.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/r4a/bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["config"].get("labels_i64"), d.get("psa_probe"), d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("frac"))
print(d.get("ohem_kth_branch", {}).get("trained_like"))
PY
Q="--steps 20 --warmup 8 --no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --no-kernel-timing"
for rep in 1 2; do
  ( timeout 300 python bench.py $Q ) 2>&1 | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default      ', d['value'], d['ms_per_step'])"
  ( TSG_CONV_C64_S1=0 timeout 300 python bench.py $Q ) 2>&1 | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c64 s1 on gen', d['value'], d['ms_per_step'])"
  ( TSG_CONV_C64_S1=0 TSG_CONV_GEN_BN_ON_LOAD=1 timeout 300 python bench.py $Q ) 2>&1 | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c64 s1 on gen + bn on load', d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/ab_c64.log
PMC_C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" bash tools/pmc_kernel.sh tools/bench_head.py ohem_up heads > $O/pmc_heads.txt 2>&1; tail -n 12 $O/pmc_heads.txt
( SIZE=512 BATCH=2 timeout 600 python tools/diag_fp64_truth.py ) > $O/fp64_truth_512.log 2>&1; cat $O/fp64_truth_512.log | tail -n 8
