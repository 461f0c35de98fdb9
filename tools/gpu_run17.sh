#!/bin/bash
# round 2, call 17: cluster path v2 (weight / bias prefetch, XCD-aware stores) + heads over 8 waves: tests, latency, timers, c2 bench
mkdir -p gpurun_out
./build/probes/xcc_t > gpurun_out/r02n_xcc.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_planner.py tests/test_gpu_edge.py tests/test_gpu_td_target.py -q -m gpu -x --timeout 300 > gpurun_out/r02n_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02n_pytest.log
tail -15 gpurun_out/r02n_pytest.log
timeout 300 python tools/probes/cluster_latency.py > gpurun_out/r02n_latency.txt 2>&1
cat gpurun_out/r02n_latency.txt
CLUSTER_MODES=1 CLUSTER_ENVS=1 TDMPC2_PLAN_LIB=build/ablate/lib_cl48_timing.so TDMPC2_TIMING=1 timeout 300 python tools/probes/cluster_latency.py c2 > gpurun_out/r02n_timing.txt 2>&1
cat gpurun_out/r02n_timing.txt
timeout 600 python bench.py --skip-extra-configs > gpurun_out/r02n_bench.json 2> gpurun_out/r02n_bench.err
cat gpurun_out/r02n_bench.json
cat gpurun_out/r02n_xcc.txt
