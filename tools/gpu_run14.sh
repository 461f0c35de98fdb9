#!/bin/bash
# round 2, call 14: cluster hand-over probe (tools/probes/cluster_probe.hip) + the agent-level td_target / pixel act() tests
mkdir -p gpurun_out
timeout 120 ./build/probes/cluster_probe > gpurun_out/r02k_cluster_probe.txt 2>&1
echo "probe rc=$?" >> gpurun_out/r02k_cluster_probe.txt
timeout 600 python -m pytest tests/test_gpu_boundary.py -q -m gpu -x -k "agent_td_target or pixel" > gpurun_out/r02k_pytest.log 2>&1
tail -5 gpurun_out/r02k_pytest.log
cat gpurun_out/r02k_cluster_probe.txt
