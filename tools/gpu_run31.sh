#!/bin/bash
# steady-state loop of the narrow heads' contraction (kloop_tile_s): c2 throughput + single-plan latency, A/B; c2 parity
mkdir -p gpurun_out
out=gpurun_out/r03i_head_steady.txt; : > $out
TDMPC2_PLAN_LIB=build/ablate/lib_hs.so timeout 600 python -m pytest tests/test_gpu_planner.py -q -m gpu -x --timeout 300 -k "c2" 2>&1 | tail -2 >> $out
for v in hns hs hns hs; do
  echo "== $v" >> $out
  TDMPC2_PLAN_LIB=build/ablate/lib_$v.so timeout 300 python bench.py --steps 10 --warmup 3 --skip-cpu-baseline --skip-extra-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plans/s', d['value'], 'rollout_ms', d['roofline']['avg_launch_ms'], {k:v for k,v in d['extra'].items() if 'latency_ms_single_env' in k and 'obs' not in k}, 'td_us', d['extra'].get('td_target_us_768_rows'))" >> $out
done
cat $out
