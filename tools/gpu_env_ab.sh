#!/bin/bash
# A/B of ENVIRONMENT switches of one library in ONE gpurun call (interleaved twice; boxes of the pool differ by up to 15 %).
# (AB_REPS: interleaved repetitions, default 2)
# usage: gpurun -- bash tools/gpu_env_ab.sh <tag> "<config envs steps>[;<config envs steps>...]" "<ENV=..>" "<ENV=.. ENV=..>" ...
#   e.g. bash tools/gpu_env_ab.sh r3e "c3 30 8;c4 8 4" "TDMPC2_ONE_STREAM=1" "A=0"
cd "$(dirname "$0")/.."
TAG=$1; SPECS=$2; shift 2
ENVSETS=("$@")
mkdir -p gpurun_out
out=gpurun_out/${TAG}_ab.txt; : > $out
IFS=';' read -r -a SP <<< "$SPECS"
for rep in $(seq 1 ${AB_REPS:-2}); do
  for envset in "${ENVSETS[@]}"; do
    for spec in "${SP[@]}"; do
      set -- $spec
      echo "== [$envset] $1 E=$2" >> $out
      env TDMPC2_BENCH_EXACT_STEPS=1 $envset timeout 300 python bench.py --config $1 --envs $2 --steps $3 --warmup 2 --skip-cpu-baseline --skip-extra-configs --skip-traffic 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plans/s', d['value'], 'stage_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'lat1_ms', d['extra'].get('latency_ms_single_env'), 'sha', d['extra'].get('action_sha1'))" >> $out 2>&1
    done
  done
done
cat $out
