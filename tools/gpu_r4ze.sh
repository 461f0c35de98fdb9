#!/bin/bash
# ks_rollout with ONE reciprocal for two Mish elements (-DMISH_PAIR_RCP, timing + parity of the bench line only; not adopted here)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r4ze_mish_pair_rcp_ab.txt; : > $OUT
for rep in 1 2; do
  for v in fbase mishp; do
    TDMPC2_BENCH_EXACT_STEPS=1 TDMPC2_PLAN_LIB=build/ablate/lib_${v}.so timeout 200 python bench.py --steps 10 --warmup 3 --skip-cpu-baseline --skip-extra-configs --skip-traffic 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['extra'].get('parity',{}); print('$v', 'plans/s', d['value'], 'launch_ms', d['roofline']['avg_launch_ms'], 'lat1_ms', d['extra'].get('latency_ms_single_env'), 'action mse vs reference', p.get('action_mse_vs_reference'), 'max', p.get('action_max_abs_diff'))" >> $OUT
  done
done
cat $OUT
