#!/bin/bash
mkdir -p gpurun_out/c27
B="bench.py --gpus 1 --steps 30 --warmup 10 --no-cpu-baseline"
echo "(a) env-only forced"; MASTER_ADDR=127.0.0.1 MASTER_PORT=29521 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 TSG_FORCE_COLLECTIVES=1 timeout 300 python $B 2>&1 | tail -1 | cut -c1-130
echo "(b) torchrun forced"; TSG_FORCE_COLLECTIVES=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29523 $B 2>&1 | tail -1 | cut -c1-130
echo "(c) torchrun plain"; timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29525 $B 2>&1 | tail -1 | cut -c1-130
echo "(d) torchrun forced OMP=8"; OMP_NUM_THREADS=8 TSG_FORCE_COLLECTIVES=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29527 $B 2>&1 | tail -1 | cut -c1-130
echo "(e) plain python"; timeout 300 python $B 2>&1 | tail -1 | cut -c1-130
