#!/bin/bash
# PSA af256x64x2: parity, colstat chunk sweep, per-kernel times, SQ counters of psa_mm
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3m; mkdir -p $O
export PSA_QUICK=1 TSG_PSA_CFG=${TSG_PSA_CFG:-af256x64x2}
( unset PSA_QUICK; timeout 400 python -m pytest tests/test_psa_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1; echo "pytest: $(tail -n 1 $O/pytest.log)"
for cc in 240 120 80 60; do
  echo "== colchunks $cc: $(TSG_PSA_COLCHUNKS=$cc timeout 200 python tools/bench_psa.py 2>&1 | grep bfloat16 | cut -c1-170)"
done
for cc in 240 80; do
  out=$O/trace$cc; rm -rf $out; mkdir -p $out
  (cd /tmp && TSG_PSA_COLCHUNKS=$cc timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o psa -- python $R/tools/bench_psa.py > $out.log 2>&1)
  f=$(find $out -name "*kernel_stats.csv" | head -1); echo "== kernels, colchunks $cc"; grep -E "psa_" "$f" | awk -F'","' '{n=$1; sub(/^"/,"",n); printf "%-100s calls %s avg %.1f us\n", substr(n,1,100), $2, $4/1000}'
  find $out -name "*kernel_trace.csv" -delete
done
bash tools/pmc_kernel.sh tools/bench_psa.py psa_mm psa_a 2>&1 | cut -c1-330
PMC_C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS" bash tools/pmc_kernel.sh tools/bench_psa.py psa_mm psa_b 2>&1 | cut -c1-330
PMC_C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_WAVES" bash tools/pmc_kernel.sh tools/bench_psa.py psa_mm psa_c 2>&1 | cut -c1-330
