#!/bin/bash
# GPU-side check run: per-file pytest logs, smoke, short bench (+ optional rocprof) -> gpurun_out/
mkdir -p gpurun_out
for f in ${TESTS-bn ohem upsample focal psa pool fused_head optim}; do
  timeout 900 python -m pytest tests/test_${f}_gpu.py -x -q -m gpu > gpurun_out/test_$f.log 2>&1
  echo "== $f rc=$?"; grep -E "passed|failed|error|Fatal|fault|^E  " gpurun_out/test_$f.log | tail -8
done
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "== smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 ${BENCH_ARGS---no-cpu-baseline} > gpurun_out/bench.log 2>&1; echo "== bench rc=$?"; tail -2 gpurun_out/bench.log
if [ -n "$PROF" ]; then
  export TMPDIR=/tmp
  out=$PWD/gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $out -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $out.log 2>&1)
  echo "== rocprof rc=$?"; find $out -name "*kernel_stats*" | head -3
  f=$(find $out -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f"
  # keep only the small summaries
  find $out -name "*kernel_trace.csv" -size +20M -delete
fi
