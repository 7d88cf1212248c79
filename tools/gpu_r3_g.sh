#!/bin/bash
# round 3, call G: adaptive average pooling + split upsample backward tests; MIOpen find mode over configs 3-5 to extend the
# shipped find-db; immediate-mode runs on the new db
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3g; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_pool_gpu.py tests/test_upsample_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
rm -rf gpurun_out/miopen_db; mkdir -p gpurun_out/miopen_db
cp torchseg_amd/miopen_db/* gpurun_out/miopen_db/
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db
for c in pspnet psanet dfn; do
  ( time timeout 700 python bench.py --config $c --steps 3 --warmup 3 --miopen-find 1 --no-cpu-baseline --no-kernel-timing ) > $O/find_$c.log 2>&1
  echo "find $c: $(grep -o '"value": [0-9.]*' $O/find_$c.log | head -1) $(grep real $O/find_$c.log)"
done
for c in pspnet psanet dfn; do
  ( time timeout 300 python bench.py --config $c --steps 20 --warmup 10 --miopen-find 0 --no-cpu-baseline ) > $O/bench_$c.log 2>&1
  echo "immediate mode on the new db, $c: $(grep -o '"value": [0-9.]*' $O/bench_$c.log | head -1)"
done
( time timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ohem-probe ) > $O/bench_bisenet.log 2>&1
echo "bisenet on the new db: $(grep -o '"value": [0-9.]*' $O/bench_bisenet.log | head -1)"
ls -la gpurun_out/miopen_db; wc -l gpurun_out/miopen_db/*.txt torchseg_amd/miopen_db/*.txt
