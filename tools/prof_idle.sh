#!/bin/bash
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof
rm -rf $out; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-kernel-timing > $out.log 2>&1)
tail -1 $out.log | cut -c1-200
timeout 600 python tools/host_profile.py > gpurun_out/host_profile.log 2>&1; tail -5 gpurun_out/host_profile.log | cut -c1-200
