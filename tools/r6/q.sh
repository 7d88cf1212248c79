#!/bin/bash
# one quick bench line (value, ms/step, host ms/step): tools/r6/q.sh "<label>" [ENV=VAL ...] -- [bench args]
cd "$(dirname "$0")/../.." || exit 1
label=$1; shift
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "$1" == "--" ] && shift
Q="--no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --no-kernel-timing --forced-steps 0 --steps 30 --warmup 10"
out=$(env "${envs[@]}" timeout 600 python bench.py $Q "$@" 2>gpurun_out/q_last.err | grep '^{' | tail -n 1)
if [ -z "$out" ]; then echo "$label  FAILED: $(tail -n 3 gpurun_out/q_last.err | tr '\n' ' ')"; exit 0; fi
echo "$label  $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; print("%.1f img/s  %.3f ms/step  host %.3f ms  graph=%s loss=%s" % (d["value"], d["ms_per_step"], c["host_enqueue_ms_per_step"], c["hip_graph"], c["final_loss"]))')"
