#!/bin/bash
# round 4, call C: fp64 statistics accumulators for fp32 tensors (hi / lo partial rows) + fp64 global average pooling:
# does the fp32 path now meet north_star's 1e-4 on the logits?  BN / pool / headline / families / trajectory tests.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4c; mkdir -p $O
export TMPDIR=/tmp
( SIZE=1024 BATCH=2 timeout 600 python tools/diag_fp64_truth.py ) > $O/fp64_truth_1024.log 2>&1; tail -n 9 $O/fp64_truth_1024.log
( time timeout 900 python -m pytest tests/test_bn_gpu.py tests/test_bn_multirank_gpu.py tests/test_syncbn_oracle.py tests/test_pool_gpu.py tests/test_stemfuse_gpu.py -x -q -m gpu ) > $O/pytest_bn.log 2>&1; tail -n 4 $O/pytest_bn.log
( time timeout 900 python -m pytest tests/test_headline_gpu.py -x -q -m gpu -s ) > $O/pytest_headline.log 2>&1; tail -n 4 $O/pytest_headline.log; grep -E "^head|headline fp32|^fp32|^bf16|batch 16" $O/pytest_headline.log | head -30
( time timeout 900 python -m pytest tests/test_families_gpu.py -x -q -m gpu -s ) > $O/pytest_families.log 2>&1; tail -n 4 $O/pytest_families.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -n 3 $O/smoke.log
