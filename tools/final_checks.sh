#!/bin/bash
# round-end style checks on one GPU box: full gpu test suite, smoke, default bench (with cpu baseline),
# and the N>1 code path forced on one rank (RCCL communicator of size 1).
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
s=$(date +%s); timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; e=$(date +%s); echo "bench default rc=$? wall=$((e-s))s"; tail -1 gpurun_out/bench_default.log | cut -c1-2500
TSG_FORCE_COLLECTIVES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_forced_coll.log 2>&1; echo "forced-collectives rc=$?"; tail -1 gpurun_out/bench_forced_coll.log | cut -c1-400
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_torchrun1.log 2>&1; echo "torchrun-1 rc=$?"; tail -1 gpurun_out/bench_torchrun1.log | cut -c1-300
