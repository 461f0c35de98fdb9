#!/bin/bash
# A/B of variant libraries in ONE gpurun call (boxes of the pool differ by up to 15 %: only same-call numbers compare).
# Build the variants here first: ABLATE_VARIANTS="base:;new:-DFLAG" tools/ablate.sh build
# usage: gpurun -- bash tools/gpu_ab.sh <tag> "<variant names>" [bench args]      (each variant is timed twice, interleaved)
cd "$(dirname "$0")/.."
TAG="${1:-ab}"; NAMES="${2:-base new}"; shift 2 || true
ARGS="${*:---steps 10 --warmup 3 --skip-cpu-baseline --skip-extra-configs --skip-traffic}"
mkdir -p gpurun_out
out=gpurun_out/${TAG}_ab.txt; : > $out
for rep in 1 2; do
  for v in $NAMES; do
    TDMPC2_BENCH_EXACT_STEPS=1 TDMPC2_PLAN_LIB=build/ablate/lib_${v}.so timeout 300 python bench.py $ARGS 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'plans/s', d['value'], 'launch_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'lat1_ms', d['extra'].get('latency_ms_single_env'), 'sha', d['extra'].get('action_sha1'), 'parity', (d['extra'].get('parity') or {}).get('action_max_abs_diff'))" >> $out
  done
done
cat $out
