#!/bin/bash
# model-level parity on the round's final library
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3t; mkdir -p $O
export TMPDIR=/tmp
( timeout 140 python -m pytest tests/test_headline_gpu.py tests/test_families_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1; tail -n 1 $O/pytest.log
