#!/bin/bash
# round 3, call A: the new bench configs + the new tests.  Everything lands in gpurun_out/r3a/.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/dev.log 2>&1
nproc > $O/nproc.log
( time timeout 300 python bench.py --steps 20 --warmup 5 ) > $O/bench_bisenet.log 2>&1
for c in pspnet dfn psanet; do
  ( time timeout 400 python bench.py --config $c --steps 10 --warmup 5 --no-cpu-baseline ) > $O/bench_$c.log 2>&1
done
( time timeout 900 python -m pytest tests/test_comm_gpu.py tests/test_integration_doc.py tests/test_evaluator_gpu.py tests/test_upsample_gpu.py -x -q -m gpu ) > $O/pytest_small.log 2>&1
( time timeout 900 python -m pytest tests/test_families_gpu.py -x -q -m gpu -s ) > $O/pytest_families.log 2>&1
tail -3 $O/bench_*.log | cut -c1-600
tail -5 $O/pytest_*.log
