#!/bin/bash
cd "$(dirname "$0")/.."
for v in timing48 timing48w0; do
TDMPC2_TIMING=1 TDMPC2_PLAN_LIB=$PWD/build/ablate/lib_$v.so timeout 300 python bench.py --steps 5 --warmup 2 --skip-cpu-baseline --skip-extra-configs 2>&1 >/dev/null | grep "tdmpc2_plan timing" | head -2
done
