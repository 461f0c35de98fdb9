#!/bin/bash
# round 2, call 15: first run of the cluster path (single-plan latency)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_planner.py -q -m gpu -x --timeout 300 -k "cluster or golden or deterministic" > gpurun_out/r02l_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02l_pytest.log
tail -25 gpurun_out/r02l_pytest.log
timeout 300 python tools/probes/cluster_latency.py > gpurun_out/r02l_latency.txt 2>&1
cat gpurun_out/r02l_latency.txt
