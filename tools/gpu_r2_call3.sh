#!/bin/bash
# round 2, GPU call 3: PSA tile configurations, hipGraph vs eager trajectories at the bench shape, fp32 logits vs MIOpen solver families
O=gpurun_out/c3; mkdir -p $O
timeout 300 python -m pytest tests/test_psa_gpu.py -q -m gpu -x > $O/pytest_psa.log 2>&1; echo "== psa pytest rc=$?"; tail -3 $O/pytest_psa.log
for cfg in 128x1 128x2 256x1 256x2; do
  echo "== PSA cfg $cfg"; TSG_PSA_CFG=$cfg PSA_QUICK=1 timeout 120 python tools/bench_psa.py 2>&1 | grep bfloat16
done | tee $O/bench_psa_cfgs.log
TSG_PSA_CFG=128x1 timeout 200 python -m pytest tests/test_psa_gpu.py -q -m gpu -k "bf16" > $O/pytest_psa_128.log 2>&1; tail -1 $O/pytest_psa_128.log
TSG_PSA_CFG=256x2 timeout 200 python -m pytest tests/test_psa_gpu.py -q -m gpu -k "bf16" > $O/pytest_psa_256x2.log 2>&1; tail -1 $O/pytest_psa_256x2.log
(export TMPDIR=/tmp; out=$PWD/$O/prof_psa; mkdir -p $out; cd /tmp && PSA_QUICK=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o psa -- python $OLDPWD/tools/bench_psa.py > $out.log 2>&1)
f=$(find $O/prof_psa -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200; find $O/prof_psa -name "*kernel_trace.csv" -delete
traj() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --trace-loss "$@" > $O/traj_$name.out 2> $O/traj_$name.err; 
  echo "== $name: $(grep -c '^step' $O/traj_$name.err) steps; losses: $(grep '^step' $O/traj_$name.err | awk '{printf "%.3f ", $4}' | cut -c1-400)"; tail -1 $O/traj_$name.out | cut -c1-160; }
traj eager_fused --warmup 10 --optimizer fused
traj eager_torch --warmup 10 --optimizer torch
traj graph1_fused --warmup 8 --graph 1 --optimizer fused
traj graph1_torch --warmup 8 --graph 1 --optimizer torch
traj graph2_fused --warmup 8 --graph 2 --optimizer fused
for e in "X=1" "MIOPEN_DEBUG_CONV_WINOGRAD=0" "MIOPEN_DEBUG_CONV_WINOGRAD=0 MIOPEN_DEBUG_CONV_FFT=0 MIOPEN_DEBUG_CONV_GEMM=0"; do
  echo "== logits fp32, $e"; env $e DIAG_BF16=0 timeout 300 python tools/diag_fp32_logits.py 2>&1 | grep -E "^stock fp32 tf32=False nchw|^ours  fp32|^stock fp32 tf32=False ch"
done | tee $O/diag_logits_solvers.log
