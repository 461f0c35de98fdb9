#!/bin/bash
# LDS staging write permutation of g_gemm_s: c3 / c4 throughput legs, A/B of variant libraries + layered parity
mkdir -p gpurun_out
out=gpurun_out/r03f_lds_perm.txt; : > $out
for v in lin perm lin perm; do
  for cfgname in c3 c4; do
    echo "== $v $cfgname" >> $out
    TDMPC2_PLAN_LIB=build/ablate/lib_$v.so timeout 300 python bench.py --config $cfgname --steps 5 --warmup 2 --skip-cpu-baseline --skip-extra-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plans/s', d['value'], 'stage_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])" >> $out
  done
done
cat $out
TDMPC2_PLAN_LIB=build/ablate/lib_perm.so timeout 900 python -m pytest tests/test_gpu_layered.py -q -m gpu -x --timeout 600 2>&1 | tail -3
