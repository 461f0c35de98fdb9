#!/bin/bash
# round 4: paired wide GEMMs (reward || dynamics, Q || Q as one grid): tests + A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_layered.py tests/test_gpu_philox.py -q --tb=short -p no:cacheprovider -x -k "switches_tiles or 317m or (golden and split) or benched_layered or starve") > gpurun_out/r4k_pytest.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r4k_pytest.log | tail -5
bash tools/gpu_env_ab.sh r4k "c3 30 8;c4 8 4" "A=0" "TDMPC2_GEMM_PAIR=0"
