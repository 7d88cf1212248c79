#!/bin/bash
O=gpurun_out/c11; mkdir -p $O
timeout 600 python -m pytest tests/test_convwrw_gpu.py -q -m gpu -x -k "rot180 or dgrad or module" > $O/pytest.log 2>&1; echo "== tests rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/pytest.log | cut -c1-220 | head -12
for d in 0 1; do
  TSG_CONV_DGRAD_FWD=$d timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 10 > $O/bench_dgradfwd$d.log 2>&1; echo "== bench DGRAD_FWD=$d: $(tail -1 $O/bench_dgradfwd$d.log | cut -c1-150)"
done
bash tools/prof_bench.sh > $O/prof.txt 2>&1; tail -60 $O/prof.txt | cut -c1-170
cp gpurun_out/prof/kernel_stats_compact.csv $O/kernel_stats_compact.csv
