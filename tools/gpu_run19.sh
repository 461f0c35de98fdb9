#!/bin/bash
# A/B of variant libraries on the c2 E=1 latency probe: VARIANTS="a b c"
mkdir -p gpurun_out
out=gpurun_out/${TAG:-r02p}_ablate.txt; : > $out
for v in $VARIANTS; do
  echo "== $v" >> $out
  CLUSTER_MODES=${MODES:-1,0} CLUSTER_ENVS=${ENVS:-1} TDMPC2_PLAN_LIB=build/ablate/lib_$v.so timeout 120 python tools/probes/cluster_latency.py ${CASES:-c2} 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
