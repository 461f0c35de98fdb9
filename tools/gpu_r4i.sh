#!/bin/bash
# round 4: two-hot in the head GEMM's epilogue + no z0 broadcast when the shared z0 products exist: tests, A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_layered.py tests/test_gpu_td_target.py tests/test_gpu_dist.py -q --tb=short -p no:cacheprovider -x) > gpurun_out/r4i_pytest.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r4i_pytest.log | tail -5
bash tools/gpu_env_ab.sh r4i "c3 30 8;c4 8 4" "A=0" "TDMPC2_TWOHOT_UNFUSED=1"
