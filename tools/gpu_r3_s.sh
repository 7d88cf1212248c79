#!/bin/bash
# last sanity of the round's final library: the refactored default weight-gradient path, PSA, one bench line
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3s; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_convwrw_gpu.py tests/test_bnconv_gpu.py tests/test_psa_gpu.py tests/test_bn_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1; tail -n 1 $O/pytest.log
echo "bench: $(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ohem-probe 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"
