#!/bin/bash
# few-row GEMM tiles: both k16-blocks' row fragments read before the chunk's first MFMA -- c3 / c4 single-plan latency A/B + parity
mkdir -p gpurun_out
out=gpurun_out/r03l_frags.txt; : > $out
TDMPC2_PLAN_LIB=build/ablate/lib_both.so timeout 600 python -m pytest tests/test_gpu_layered.py -q -m gpu -x --timeout 600 2>&1 | tail -2 >> $out
for v in fpb both fpb both; do
  for cfgname in c3 c4; do
    echo "== $v $cfgname single plan" >> $out
    TDMPC2_PLAN_LIB=build/ablate/lib_$v.so timeout 300 python tools/probes/graph_probe.py $cfgname 2>&1 | grep "eager" >> $out
  done
done
cat $out
