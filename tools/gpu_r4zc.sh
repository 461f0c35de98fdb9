#!/bin/bash
# ks_rollout phase clocks with the heads split in three (k-loop of the head GEMM, staging of the logits, row routines), read on
# wave 0 (older wave of its SIMD; computes a head column tile) and on wave 4 (younger; idles through the head GEMMs)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r4zc_ks_rollout_head_clocks.txt; : > $OUT
for v in tim0 tim4; do
  echo "== $v" >> $OUT
  TDMPC2_TIMING=1 TDMPC2_BENCH_EXACT_STEPS=1 TDMPC2_PLAN_LIB=build/ablate/lib_$v.so timeout 300 python bench.py --steps 3 --warmup 1 --skip-cpu-baseline --skip-extra-configs --skip-traffic 2>&1 >/dev/null | grep "tdmpc2_plan timing" >> $OUT
done
cat $OUT
