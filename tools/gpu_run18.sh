#!/bin/bash
# round 2, call 18/19: which of the cluster-path changes pays? (variant libraries, c2 E=1 latency) + phase timers of "all"
mkdir -p gpurun_out
out=gpurun_out/r02o_ablate.txt; : > $out
for v in v1 bias wpf fast all all8 v1; do
  echo "== $v" >> $out
  CLUSTER_MODES=1,0 CLUSTER_ENVS=1 TDMPC2_PLAN_LIB=build/ablate/lib_$v.so timeout 120 python tools/probes/cluster_latency.py c2 2>&1 | grep -v amdgpu.ids >> $out
done
echo "== timing (all)" >> $out
CLUSTER_MODES=1 CLUSTER_ENVS=1 TDMPC2_PLAN_LIB=build/ablate/lib_timing.so TDMPC2_TIMING=1 timeout 120 python tools/probes/cluster_latency.py c2 2>&1 | grep -v amdgpu.ids >> $out
cat $out
