#!/bin/bash
# round 2, GPU call 2: hipGraph bisect, fp32 logits diagnostic, fused PSA kernels (tests, bench, kernel stats)
O=gpurun_out/c2; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 200 python tools/debug_graph3.py --tag "$tag" ${ARGS} 2>&1 | grep "^\[" ; }
{
ARGS="" run base X=1
ARGS="" run nofeat TSG_STEM_CONV=0 TSG_CONV_WRW=0 TSG_SPLIT_BIAS=0
ARGS="--warm default" run warmdefault X=1
ARGS="--opt torch" run torchopt X=1
ARGS="" run nchw TSG_CHANNELS_LAST=0
ARGS="--mode thread_local" run threadlocal X=1
ARGS="" run fp32 TSG_DTYPE=fp32
} > $O/graph3.log 2>&1
cat $O/graph3.log
timeout 400 python tools/diag_fp32_logits.py > $O/diag_logits.log 2>&1; grep -v Warn $O/diag_logits.log | tail -9
timeout 600 python -m pytest tests/test_psa_gpu.py tests/test_bn_multirank_gpu.py "tests/test_ohem_gpu.py::test_ohem_selection_bit_exact_given_device_probs" -q -m gpu > $O/pytest.log 2>&1
echo "== pytest rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/pytest.log | head -20
timeout 300 python tools/bench_psa.py > $O/bench_psa.log 2>&1; cat $O/bench_psa.log | grep -v Warn
bash tools/prof_psa.sh > $O/prof_psa.txt 2>&1; head -24 $O/prof_psa.txt
cp gpurun_out/prof_psa/*/*kernel_stats.csv $O/psa_kernel_stats.csv 2>/dev/null
