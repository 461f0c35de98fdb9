#!/bin/bash
# r03k hunt, fourth step: tools/probes/r3k_probe.py variants, one process each
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r4q_r3k_probe.txt; : > $OUT
for v in nolayered base fault base+sleep base+nocluster base+sidestream base+c1first base+noclose base+nowork base+plan fault+plan; do
  timeout 120 python tools/probes/r3k_probe.py $v 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400 >> $OUT
done
for e in TDMPC2_ONE_STREAM=1 AMD_SERIALIZE_KERNEL=3 GPU_MAX_HW_QUEUES=1 GPU_MAX_HW_QUEUES=2 HIP_FORCE_DEV_KERNARG=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; do
  echo "-- $e" >> $OUT
  env $e timeout 120 python tools/probes/r3k_probe.py base 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400 >> $OUT
done
cat $OUT
