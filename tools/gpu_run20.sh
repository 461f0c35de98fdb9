#!/bin/bash
# c2-only correctness of variant libraries + latency A/B:  VARIANTS="a b" CHECK="a b"
mkdir -p gpurun_out
for v in $CHECK; do
  echo "== pytest $v"
  TDMPC2_PLAN_LIB=build/ablate/lib_$v.so timeout 600 python -m pytest tests/test_gpu_planner.py -q -m gpu -x --timeout 300 -k "c2 and (golden or cluster)" 2>&1 | tail -4
done
bash tools/gpu_run19.sh
