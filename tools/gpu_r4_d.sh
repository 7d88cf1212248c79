#!/bin/bash
# round 4, call D: sliced exact weight gradient, optimistic PSA forward (parity incl. the guarded path, probe), families timing
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4d; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_exactconv_gpu.py tests/test_psa_gpu.py -x -q -m gpu ) > $O/pytest_a.log 2>&1; tail -n 6 $O/pytest_a.log
PSA_QUICK=1 python tools/bench_psa.py 2>&1 | tail -n 2 | tee $O/psa_optimistic.log
PSA_QUICK=1 TSG_PSA_OPTIMISTIC=0 python tools/bench_psa.py 2>&1 | tail -n 2 | tee $O/psa_classic.log
( time timeout 900 python -m pytest tests/test_families_gpu.py -x -q -m gpu -s --durations=5 ) > $O/pytest_families.log 2>&1; tail -n 12 $O/pytest_families.log | cut -c1-200
( timeout 300 python bench.py --config psanet --steps 20 --warmup 10 --no-cpu-baseline ) 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('psanet', d['value'], d['ms_per_step'], d.get('psa_probe'))"
