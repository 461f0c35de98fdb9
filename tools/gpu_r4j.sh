#!/bin/bash
# round 4: ks_rollout_cl2 (two clusters per tile for a single plan): goldens, bit-identity vs one cluster, latency
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_planner.py tests/test_gpu_philox.py tests/test_gpu_boundary.py -q --tb=short -p no:cacheprovider -x -k "cluster or golden or bit_for_bit or act or graph") > gpurun_out/r4j_pytest.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r4j_pytest.log | tail -8
for m in 2 1 2 1; do
  CLUSTER_ENVS=1 CLUSTER_MODES=$m timeout 120 python tools/probes/cluster_latency.py c2 c1 2>/dev/null | tee -a gpurun_out/r4j_cluster2_latency.txt
done
