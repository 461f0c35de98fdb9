#!/bin/bash
# round 4: ablation-by-removal of the cluster path (c2, E = 1), refreshed on the round's library (variants built with tools/variant.sh)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r4g_cluster_ablation_by_removal.txt; : > $out
for rep in 1 2; do
for v in clfull clnohead clnowait clnoepi clnogemm clnone; do
  echo -n "$v: " >> $out
  CLUSTER_ENVS=1 CLUSTER_MODES=1 TDMPC2_PLAN_LIB=build/ablate/lib_${v}.so timeout 120 python tools/probes/cluster_latency.py c2 2>/dev/null | tail -1 >> $out
done
done
cat $out
