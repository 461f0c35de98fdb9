#!/bin/bash
# The 317M stress test faulted twice in the full suite (r4z) with the XCD rectangles: how often, and does the XCD-local order
# (one XCD per row block: provably no circular wait for two launches at <= 16 column blocks) ever?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r4za_stress_xcd_order.txt; : > $OUT
for rep in 1 2; do
for g in 2 1; do
  echo "== rep $rep TDMPC2_GEMM_W_XCD_ROWS=$g" >> $OUT
  TDMPC2_GEMM_W_XCD_ROWS=$g TDMPC2_STRESS_STAGES=3000 timeout 600 python -m pytest tests/test_gpu_layered.py -m gpu -q -s --tb=line -p no:cacheprovider -k "never_starve and c4" 2>&1 | grep -E "stress|passed|failed" >> $OUT
done
done
cat $OUT
