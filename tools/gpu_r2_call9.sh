#!/bin/bash
O=gpurun_out/c9; mkdir -p $O
timeout 900 python -m pytest tests/test_augment_gpu.py tests/test_evaluator_gpu.py tests/test_stemconv_gpu.py tests/test_convwrw_gpu.py tests/test_fused_head_gpu.py tests/test_metric_gpu.py tests/test_focal_gpu.py tests/test_pool_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; echo "== tests rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/pytest.log | cut -c1-220 | head -20
timeout 300 python tools/probe_conv2.py 2>&1 | grep -v Warn | tee $O/probe_conv2.log
