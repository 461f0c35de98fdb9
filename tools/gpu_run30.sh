#!/bin/bash
# branch-free steady-state k-loop of g_gemm_s: layered parity, single-plan latency and throughput legs of c3 / c4, A/B
mkdir -p gpurun_out
out=gpurun_out/r03h_steady.txt; : > $out
TDMPC2_PLAN_LIB=build/ablate/lib_steady.so timeout 900 python -m pytest tests/test_gpu_layered.py tests/test_gpu_td_target.py -q -m gpu -x --timeout 600 2>&1 | tail -2 >> $out
for v in nosteady steady nosteady steady; do
  for cfgname in c3 c4; do
    echo "== $v $cfgname single plan" >> $out
    TDMPC2_PLAN_LIB=build/ablate/lib_$v.so timeout 300 python tools/probes/graph_probe.py $cfgname 2>&1 | grep "eager" >> $out
  done
done
for v in nosteady steady; do
  echo "== $v c3 E=30" >> $out
  TDMPC2_PLAN_LIB=build/ablate/lib_$v.so timeout 300 python bench.py --config c3 --envs 30 --steps 8 --warmup 2 --skip-cpu-baseline --skip-extra-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plans/s', d['value'], 'stage_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])" >> $out
  echo "== $v c4 E=8" >> $out
  TDMPC2_PLAN_LIB=build/ablate/lib_$v.so timeout 300 python bench.py --config c4 --envs 8 --steps 4 --warmup 1 --skip-cpu-baseline --skip-extra-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plans/s', d['value'], 'stage_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])" >> $out
done
cat $out
