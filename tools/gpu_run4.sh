#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TDMPC2_PLAN_LIB=$PWD/build/ablate/lib_regen48.so
(time timeout 900 python -m pytest "tests/test_gpu_planner.py" -m gpu -q --tb=short -p no:cacheprovider -k "c2 or refit or hand_over or philox") > gpurun_out/r02d_pytest_gpu.log 2>&1
tail -15 gpurun_out/r02d_pytest_gpu.log
unset TDMPC2_PLAN_LIB
run() { # name, lib, extra env
  env $3 TDMPC2_PLAN_LIB=$2 timeout 300 python bench.py --steps 10 --warmup 3 --skip-cpu-baseline > gpurun_out/r02d_bench_$1.json 2> gpurun_out/r02d_bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02d_bench_$1.json").read().strip().splitlines()[-1])
    print("$1", "plans/s", d["value"], "ms_per_step", d["ms_per_step"], "rollout_ms", d["roofline"]["avg_launch_ms"], "lat1_ms", d["extra"].get("latency_ms_single_env"))
except Exception as e:
    print("$1 FAILED", e); print(open("gpurun_out/r02d_bench_$1.err").read()[-1500:])
PY
}
run regen $PWD/build/ablate/lib_regen48.so A=1
run regenoff $PWD/build/ablate/lib_regen48.so TDMPC2_FOLD_REFIT=0
run regen2 $PWD/build/ablate/lib_regen48.so A=1
