#!/bin/bash
# steady-state loop of the exact-fp32 contraction (kloop_f32): c2 throughput in fp32 mode, A/B; parity of the fp32 cases
mkdir -p gpurun_out
out=gpurun_out/r03j_f32_steady.txt; : > $out
TDMPC2_PLAN_LIB=build/ablate/lib_hs.so timeout 600 python -m pytest tests/test_gpu_planner.py -q -m gpu -x --timeout 300 -k "c2 and fp32" 2>&1 | tail -2 >> $out
for v in hns hs hns hs; do
  echo "== $v" >> $out
  TDMPC2_PLAN_LIB=build/ablate/lib_$v.so timeout 300 python bench.py --precision fp32 --steps 4 --warmup 1 --skip-cpu-baseline --skip-extra-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plans/s', d['value'], 'rollout_ms', d['roofline']['avg_launch_ms'])" >> $out
done
cat $out
