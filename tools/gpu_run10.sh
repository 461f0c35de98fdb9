#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
TDMPC2_PLAN_LIB=$PWD/build/ablate/lib_asmh48.so timeout 600 python -m pytest tests/test_gpu_planner.py tests/test_gpu_layers.py tests/test_gpu_td_target.py -m gpu -q --tb=short -p no:cacheprovider -k "c2" 2>&1 | grep -E "passed|failed|Abort|fault|FAILED" | head -5
run() { # name, lib
  TDMPC2_PLAN_LIB=$2 timeout 300 python bench.py --steps 10 --warmup 3 --skip-cpu-baseline --skip-extra-configs > gpurun_out/r02h_bench_$1.json 2> gpurun_out/r02h_bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02h_bench_$1.json").read().strip().splitlines()[-1])
    print("$1", "plans/s", d["value"], "rollout_ms", d["roofline"]["avg_launch_ms"], "lat1_ms", d["extra"].get("latency_ms_single_env"), "parity", d["extra"].get("parity",{}).get("action_max_abs_diff"))
except Exception as e:
    print("$1 FAILED", e); print(open("gpurun_out/r02h_bench_$1.err").read()[-800:])
PY
}
run asm $PWD/build/ablate/lib_asm48.so
run asmh $PWD/build/ablate/lib_asmh48.so
run asm2 $PWD/build/ablate/lib_asm48.so
run asmh2 $PWD/build/ablate/lib_asmh48.so
