#!/bin/bash
# Same-call A/B of two source TREES (e.g. the round-4 library + its own bench.py against HEAD): every leg interleaved, each
# bench line carries the clocks / socket power read under that leg's own load (extra.box_under_load).
# usage: gpurun -- bash tools/gpu_tree_ab.sh <tag> "<name>=<dir> <name>=<dir> ..." "<config envs steps>;..." [reps]
cd "$(dirname "$0")/.."
R=$PWD; TAG=$1; TREES=$2; SPECS=$3; REPS=${4:-2}
mkdir -p gpurun_out
out=$R/gpurun_out/${TAG}_tree_ab.txt; : > $out
IFS=';' read -r -a SP <<< "$SPECS"
for rep in $(seq 1 $REPS); do
  for spec in "${SP[@]}"; do
    set -- $spec
    for t in $TREES; do
      name=${t%%=*}; dir=${t#*=}
      (cd $R/$dir && TDMPC2_BENCH_EXACT_STEPS=1 timeout 400 python bench.py --config $1 --envs $2 --steps $3 --warmup 3 --skip-cpu-baseline --skip-extra-configs --skip-traffic 2>/dev/null \
        | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['extra'].get('box_under_load',{})
print('$name', '$1', 'E=$2', 'plans/s', d['value'], 'unit_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'lat1_ms', d['extra'].get('latency_ms_single_env'), 'faults', d['extra'].get('bounded_wait_faults'), '|', b.get('sclk'), b.get('power_w'))" >> $out 2>&1)
    done
  done
done
cat $out
