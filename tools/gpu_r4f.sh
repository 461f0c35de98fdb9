#!/bin/bash
# round 4: g_gemm_wn (256 x 224 / 192 tiles for the 48M model's 56 / 24 column tiles): bit-exactness vs the small tiles, A/B vs 256 x 256
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_layered.py tests/test_gpu_philox.py -q --tb=short -p no:cacheprovider -k "switches_tiles or 317m or (golden and split) or benched_layered or epilogue") > gpurun_out/r4f_pytest.log 2>&1
tail -4 gpurun_out/r4f_pytest.log
bash tools/gpu_env_ab.sh r4f "c3 30 8" "A=0" "TDMPC2_GEMM_W_NT=8"
AB_REPS=1 bash tools/gpu_env_ab.sh r4f4 "c4 8 4" "A=0" "TDMPC2_GEMM_W_NT=8"
