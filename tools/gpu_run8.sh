#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
TDMPC2_PLAN_LIB=$PWD/build/ablate/lib_rt48.so python tools/probes/refit_probe.py
python tools/probes/fold_probe.py
timeout 600 python -m pytest tests/test_gpu_planner.py tests/test_gpu_edge.py tests/test_gpu_dist.py tests/test_gpu_layered.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r02g_pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r02g_pytest.log | head
