#!/bin/bash
# One-off: let MIOpen's find mode populate a user perf/find db for the bench's conv shapes,
# then check what the immediate-mode (benchmark=False) run gains from it.
mkdir -p gpurun_out/miopen_db
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db
timeout 1200 python bench.py --steps 3 --warmup 2 --miopen-find 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_find1.log 2>&1; echo "find1 rc=$?"; tail -1 gpurun_out/bench_find1.log | cut -c1-200
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_db.log 2>&1; echo "db rc=$?"; tail -1 gpurun_out/bench_db.log | cut -c1-200
ls -la gpurun_out/miopen_db | head; du -sh gpurun_out/miopen_db
