#!/bin/bash
# round 4, first GPU call: the layered family on the fragment-packed operand layout + g_gemm_w (256 x 256, LDS-DMA), tests + A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(time timeout 1200 python -m pytest tests/test_gpu_layered.py tests/test_gpu_td_target.py tests/test_gpu_dist.py tests/test_gpu_philox.py -q --tb=short --maxfail=12 -p no:cacheprovider) > gpurun_out/r4a_pytest_layered.log 2>&1
tail -40 gpurun_out/r4a_pytest_layered.log
bash tools/gpu_env_ab.sh r4a "c3 30 8;c4 8 4" "A=0" "TDMPC2_GEMM_W256_MIN=-1"
