#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(time timeout 900 python -m pytest "tests/test_gpu_planner.py" "tests/test_gpu_boundary.py" -m gpu -q --tb=short -p no:cacheprovider) > gpurun_out/r02e_pytest_gpu.log 2>&1
tail -12 gpurun_out/r02e_pytest_gpu.log
run() { # name, lib, extra env
  env $3 TDMPC2_PLAN_LIB=$2 timeout 300 python bench.py --steps 10 --warmup 3 --skip-cpu-baseline > gpurun_out/r02e_bench_$1.json 2> gpurun_out/r02e_bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02e_bench_$1.json").read().strip().splitlines()[-1])
    print("$1", "plans/s", d["value"], "ms_per_step", d["ms_per_step"], "rollout_ms", d["roofline"]["avg_launch_ms"], "lat1_ms", d["extra"].get("latency_ms_single_env"))
except Exception as e:
    print("$1 FAILED", e); print(open("gpurun_out/r02e_bench_$1.err").read()[-1500:])
PY
}
run fold $PWD/tdmpc2_amd/libtdmpc2_plan.so A=1
run foldoff $PWD/tdmpc2_amd/libtdmpc2_plan.so TDMPC2_FOLD_REFIT=0
run fold2 $PWD/tdmpc2_amd/libtdmpc2_plan.so A=1
