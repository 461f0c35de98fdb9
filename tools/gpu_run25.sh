#!/bin/bash
# deep row-operand staging (and split-K) for the layered family's few-row GEMMs: parity + single-plan latency of c3 / c4
mkdir -p gpurun_out
export TDMPC2_PLAN_LIB=build/ablate/lib_sk.so
timeout 1500 python -m pytest tests/test_gpu_layered.py tests/test_gpu_td_target.py tests/test_gpu_edge.py -q -m gpu -x --timeout 600 > gpurun_out/r03c_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03c_pytest.log; tail -4 gpurun_out/r03c_pytest.log
out=gpurun_out/r03c_staging.txt; : > $out
for cfgname in c3 c4; do
  for v in "TDMPC2_GEMM_SD1=1" "X=1" "TDMPC2_SPLIT_K=4" "TDMPC2_SPLIT_K=8"; do
    echo "== $cfgname $v" >> $out
    env $v timeout 300 python tools/probes/graph_probe.py $cfgname 2>&1 | grep "eager" >> $out
  done
done
cat $out
