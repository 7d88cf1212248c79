#!/bin/bash
mkdir -p gpurun_out/c25
TSG_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/c25/bench_forced.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/c25/bench_forced.log | cut -c1-300
timeout 600 python -m pytest tests/test_bn_multirank_gpu.py tests/test_comm_gpu.py -x -q 2>&1 | tail -2
