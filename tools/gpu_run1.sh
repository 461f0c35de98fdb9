#!/bin/bash
# round-2 GPU session 1: full -m gpu suite (parity report), then A/B of kernel variants on the bench workload
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider) > gpurun_out/r02a_pytest_gpu.log 2>&1
tail -25 gpurun_out/r02a_pytest_gpu.log
for v in base48 pair48 base48; do
  TDMPC2_PLAN_LIB=$PWD/build/ablate/lib_$v.so timeout 300 python bench.py --steps 10 --warmup 3 --skip-cpu-baseline > gpurun_out/r02a_bench_$v.json 2> gpurun_out/r02a_bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02a_bench_$v.json").read().strip().splitlines()[-1])
    print("$v", "plans/s", d["value"], "rollout_ms", d["roofline"]["avg_launch_ms"], "lat1_ms", d["extra"].get("latency_ms_single_env"))
except Exception as e:
    print("$v FAILED", e); print(open("gpurun_out/r02a_bench_$v.err").read()[-1500:])
PY
done
