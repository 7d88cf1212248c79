#!/bin/bash
mkdir -p gpurun_out/c26
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
TSG_FORCE_COLLECTIVES=1 timeout 600 python tools/host_profile.py > gpurun_out/c26/host_forced.log 2>&1; echo rc=$?
unset MASTER_ADDR MASTER_PORT RANK LOCAL_RANK WORLD_SIZE
timeout 600 python tools/host_profile.py > gpurun_out/c26/host_plain.log 2>&1; echo rc=$?
