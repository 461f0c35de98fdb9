#!/bin/bash
# A/B of where a phase's LDS-DMA requests are issued (layered_mid.cuh: GM_SPREAD_ISSUE, layered_wide.cuh: GW_SPREAD_ISSUE) -- variant libraries
# under build/ab/ (built with TDMPC2_FLAGS_layered=-DG?_SPREAD_ISSUE=n), all legs in ONE gpurun call, twice.
# usage: gpurun -- bash tools/gpu_issue_placement_ab.sh <tag> "<gm libs>" "<gw libs>"
cd "$(dirname "$0")/.."
TAG=${1:-r6zu}; GM=${2:-"serial gm1"}; GW=${3:-"serial gw1"}
run() { # lib config envs steps
  env TDMPC2_BENCH_EXACT_STEPS=1 TDMPC2_PLAN_LIB=build/ab/lib_$1.so timeout 300 python bench.py --config $2 --envs $3 --steps $4 --warmup 2 --skip-cpu-baseline --skip-extra-configs --skip-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 E=$3 plans/s', d['value'], 'stage_ms', d['roofline']['avg_launch_ms'], 'sha', d['extra'].get('action_sha1'))"
}
mkdir -p gpurun_out
out=gpurun_out/${TAG}_issue_placement_ab.txt; : > $out
for rep in 1 2; do
  for lib in $GM; do run $lib c4 1 12 >> $out; run $lib c3 1 60 >> $out; run $lib c3 4 20 >> $out; done
  for lib in $GW; do run $lib c3 30 8 >> $out; run $lib c4 8 4 >> $out; done
done
cat $out
