#!/bin/bash
# round 4, second GPU call: kernel traces + PMC (sq1, grbm, fetch) of the c3 / c4 legs on g_gemm_w, tile-order A/B
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
TRACE_HEAD=40 bash tools/gpu_trace.sh r4b_c3 c3 30 4
TRACE_HEAD=40 bash tools/gpu_trace.sh r4b_c4 c4 8 2
PMC_PASSES="sq1 grbm fetch" bash tools/gpu_pmc.sh r4b_c3 --config c3 --envs 30 --steps 2 --warmup 1 --skip-cpu-baseline --skip-extra-configs --skip-traffic > /dev/null
python tools/pmc_summary.py gpurun_out/pmc_r4b_c3 g_gemm > gpurun_out/r4b_c3_pmc.txt 2>&1
PMC_PASSES="sq1 grbm fetch" bash tools/gpu_pmc.sh r4b_c4 --config c4 --envs 8 --steps 1 --warmup 1 --skip-cpu-baseline --skip-extra-configs --skip-traffic > /dev/null
python tools/pmc_summary.py gpurun_out/pmc_r4b_c4 g_gemm > gpurun_out/r4b_c4_pmc.txt 2>&1
grep -A22 "g_gemm_w" gpurun_out/r4b_c3_pmc.txt | head -60
grep -A22 "g_gemm_w" gpurun_out/r4b_c4_pmc.txt | head -60
rm -rf gpurun_out/pmc_r4b_c3/*/ gpurun_out/pmc_r4b_c4/*/ 2>/dev/null
AB_REPS=1 bash tools/gpu_env_ab.sh r4b "c3 30 8;c4 8 4" "TDMPC2_GEMM_W_XCD_ROWS=0" "TDMPC2_GEMM_W_XCD_ROWS=1" "TDMPC2_ONE_STREAM=1"
