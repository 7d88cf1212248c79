#!/bin/bash
# The weight-gradient side stream and the hardware queue it lands on: plain step and the N > 1 code path on one rank
# (TSG_FORCE_COLLECTIVES=1), by priority class of the side stream and by the number of hardware queues the HIP runtime may open
B="python bench.py --steps 20 --warmup 10 --ref-steps 0 --fp32-steps 0 --no-cpu-baseline --no-ohem-probe --no-psa-probe --no-kernel-timing --i64-steps 0"
run() {  # label, env...
  label=$1; shift
  v=$(env "$@" $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'))")
  echo "$label: $v"
}
run "plain  normal        " TSG_WRW_PRIO=normal
run "forced normal        " TSG_WRW_PRIO=normal TSG_FORCE_COLLECTIVES=1
run "forced low           " TSG_WRW_PRIO=low TSG_FORCE_COLLECTIVES=1
run "forced high          " TSG_WRW_PRIO=high TSG_FORCE_COLLECTIVES=1
run "forced normal 8 queues" TSG_WRW_PRIO=normal TSG_FORCE_COLLECTIVES=1 GPU_MAX_HW_QUEUES=8
run "forced no side stream" TSG_WRW_STREAM=0 TSG_FORCE_COLLECTIVES=1
run "plain  low           " TSG_WRW_PRIO=low
run "plain  high          " TSG_WRW_PRIO=high
run "plain  normal 8 queues" TSG_WRW_PRIO=normal GPU_MAX_HW_QUEUES=8
run "plain  normal (again)" TSG_WRW_PRIO=normal
