#!/bin/bash
# round 4, third GPU call: g_gemm_w variants (ring of 5, DMA issue after two MFMAs, static priority), phase clocks, W256_MIN
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_layered.py -q --tb=short -p no:cacheprovider -k "switches_tiles or 317m or (golden and split and (c3 or c4))") > gpurun_out/r4c_pytest.log 2>&1
tail -3 gpurun_out/r4c_pytest.log
for spec in "c3 30 8" "c4 8 4"; do
  set -- $spec
  TDMPC2_GW_TIMING=1 TDMPC2_PLAN_LIB=build/ablate/lib_timing.so timeout 300 python bench.py --config $1 --envs $2 --steps $3 --warmup 2 --skip-cpu-baseline --skip-extra-configs --skip-traffic 2>&1 >/dev/null | grep "g_gemm_w timing" | tee -a gpurun_out/r4c_gw_timing.txt
done
bash tools/gpu_ab.sh r4c_c3 "base ns5 late prio" --config c3 --envs 30 --steps 8 --warmup 2 --skip-cpu-baseline --skip-extra-configs --skip-traffic
bash tools/gpu_ab.sh r4c_c4 "base ns5 late prio" --config c4 --envs 8 --steps 4 --warmup 2 --skip-cpu-baseline --skip-extra-configs --skip-traffic
AB_REPS=1 bash tools/gpu_env_ab.sh r4c_env "c3 30 8" "A=0" "TDMPC2_GEMM_W256_MIN=128"
