#!/bin/bash
# One-off on the GPU box: let MIOpen's find mode extend the shipped user find-db with the convolution shapes of the
# current code path (TSG_CONV_DGRAD_FWD=2: data gradients of every stride-1 3x3 layer as forward convolutions), then
# measure the immediate-mode run (what bench.py does by default) on the new db.  Result: gpurun_out/miopen_db/*.
rm -rf gpurun_out/miopen_db; mkdir -p gpurun_out/miopen_db
cp torchseg_amd/miopen_db/* gpurun_out/miopen_db/
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db
TSG_CONV_DGRAD_FWD=2 timeout 1200 python bench.py --steps 3 --warmup 2 --miopen-find 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_find1.log 2>&1; echo "find rc=$?"; tail -1 gpurun_out/bench_find1.log | cut -c60-150
for d in 1 2; do
  TSG_CONV_DGRAD_FWD=$d timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_db_d$d.log 2>&1; echo "immediate mode on the new db, DGRAD_FWD=$d: $(tail -1 gpurun_out/bench_db_d$d.log | cut -c60-150)"
done
ls -la gpurun_out/miopen_db; wc -l gpurun_out/miopen_db/*.txt torchseg_amd/miopen_db/*.txt
