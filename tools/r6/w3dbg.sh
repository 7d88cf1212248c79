#!/bin/bash
# where does a weight-gradient tile's time go?  TSG_MFMA_PRIO bits (debug build only): 2 = no MFMAs, 4 = no LDS staging, 8 = no fetch
cd "$(dirname "$0")/../.." || exit 1
for m in 0 2 4 8 12 14 6; do
  echo "== TSG_MFMA_PRIO=$m"
  TSG_MFMA_PRIO=$m ONLY="${ONLY:-layer}" timeout 200 python tools/bench_conv3wrw.py 2>&1 | grep -E "layer1|layer2 |layer3 |layer4 " | sed 's/MIOpen.*//'
done
