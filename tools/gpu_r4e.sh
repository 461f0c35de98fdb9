#!/bin/bash
# round 4: ks_rollout -- heads contracted by all 8 waves (HEAD_SPLITK) vs base, same call; per-phase clocks of the current kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/gpu_ab.sh r4e "fbase hsk" --steps 10 --warmup 3 --skip-cpu-baseline --skip-extra-configs --skip-traffic
TDMPC2_TIMING=1 TDMPC2_PLAN_LIB=build/ablate/lib_ftim.so timeout 200 python bench.py --steps 5 --warmup 2 --skip-cpu-baseline --skip-extra-configs --skip-traffic 2>&1 >/dev/null | grep "tdmpc2_plan timing" | tee gpurun_out/r4e_split_timing.txt
TDMPC2_PLAN_LIB=build/ablate/lib_hsk.so timeout 300 python -m pytest tests/test_gpu_planner.py -q -x --tb=short -p no:cacheprovider -k "golden or oracle" 2>&1 | tail -3
