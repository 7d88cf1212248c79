#!/bin/bash
O=gpurun_out/c8; mkdir -p $O
timeout 900 python -m pytest tests/test_augment_gpu.py tests/test_evaluator_gpu.py tests/test_graph_gpu.py tests/test_headline_gpu.py -q -m gpu -s -k "not bn_kernels and not trajectory" > $O/pytest_new.log 2>&1; echo "== new tests rc=$?"; grep -E "passed|failed|^FAILED|^E  |headline|^head" $O/pytest_new.log | cut -c1-220 | head -30
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_augment_gpu.py --deselect tests/test_evaluator_gpu.py --deselect tests/test_graph_gpu.py --deselect tests/test_headline_gpu.py > $O/pytest_all.log 2>&1; echo "== rest of the suite rc=$?"; tail -5 $O/pytest_all.log | cut -c1-200
