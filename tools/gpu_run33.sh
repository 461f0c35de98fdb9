#!/bin/bash
# is the r02ac bench line (8.7 k plans/s, 1.75 ms) the box or the library?  in-tree library vs the 48-only A/B builds, same box
mkdir -p gpurun_out
out=gpurun_out/r03k_recheck.txt; : > $out
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|mclk|Power" | head -6 >> $out
for v in tree hns hs tree; do
  echo "== $v" >> $out
  if [ $v = tree ]; then unset TDMPC2_PLAN_LIB; else export TDMPC2_PLAN_LIB=build/ablate/lib_$v.so; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --skip-cpu-baseline --skip-extra-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plans/s', d['value'], 'rollout_ms', d['roofline']['avg_launch_ms'], {k:v for k,v in d['extra'].items() if 'latency_ms_single_env' in k and 'obs' not in k}, 'fp32', d['extra']['exact_fp32_mode']['value'])" >> $out
done
cat $out
