#!/bin/bash
# r03k, root cause: the handle's stream hand-over (StreamTurn).  The failing sequence of tools/gpu_r4o.sh and the new test, with
# the hand-over and -- negative control -- without it (TDMPC2_DEBUG_NO_TURN=1); then what the extra event record costs a single plan.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r4r_stream_turn.txt; : > $OUT
SEL='wait or capturable or rebinding or downgrades or td_target_says or graph_replay or another_stream'
PY="timeout 300 python -m pytest -m gpu -q --tb=line -p no:cacheprovider tests/test_gpu_layered.py tests/test_gpu_boundary.py -k"
for rep in 1 2; do
  echo "== rep $rep: with the hand-over" >> $OUT
  $PY "$SEL" 2>&1 | grep -E "passed|failed|^/root" | cut -c1-300 >> $OUT
  echo "== rep $rep: TDMPC2_DEBUG_NO_TURN=1" >> $OUT
  TDMPC2_DEBUG_NO_TURN=1 $PY "$SEL" 2>&1 | grep -E "passed|failed|^/root" | cut -c1-300 >> $OUT
done
for rep in 1 2; do
  for e in A=0 TDMPC2_DEBUG_NO_TURN=1; do
    echo "== latency, $e" >> $OUT
    env $e TDMPC2_BENCH_EXACT_STEPS=1 timeout 300 python bench.py --steps 3 --warmup 1 --skip-cpu-baseline --skip-extra-configs --skip-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']
print('plans/s', d['value'], 'single plan ms', e.get('latency_ms_single_env'), 'from obs', e.get('latency_ms_single_env_from_obs'), 'td_target us', e.get('td_target_us_768_rows'), 'encode us', e.get('encode_us_single_env'))" >> $OUT
  done
done
cat $OUT
