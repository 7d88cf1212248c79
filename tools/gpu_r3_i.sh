#!/bin/bash
# PSA ablations (diagnosis): which part of the K tile costs the time?  TSG_PSA_ABLATE bits: 1 no A reloads, 2 no B reloads,
# 4 no exp, 8 no MFMA.  Results are numerically wrong by construction; only the times count.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3i; mkdir -p $O
for cfg in split128x64x1 128x1 split128x128x1; do
  for ab in 0 4 2 1 3 7 8 15; do
    echo "cfg $cfg ablate $ab: $(TSG_PSA_CFG=$cfg TSG_PSA_ABLATE=$ab PSA_QUICK=1 timeout 100 python tools/bench_psa.py 2>&1 | grep bfloat16 | cut -c1-100)"
  done
done 2>&1 | tee $O/psa_ablate.txt
