#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4o; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_conv3g_gpu.py tests/test_bnconv_gpu.py tests/test_headline_gpu.py -q -m gpu ) > $O/pytest_b.log 2>&1; tail -n 6 $O/pytest_b.log
