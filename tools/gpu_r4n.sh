#!/bin/bash
# r03k hunt (profiles/README.md "A ROCm runtime trap"): the round-3 failure sequence with the side stream destroyed again
# (TDMPC2_DEBUG_SIDE_DESTROY: 0 pool, 1 destroy, 2 stream-sync + destroy, 3 device-sync + destroy), twice each.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r4n_side_destroy.txt; : > $OUT
SEL='wait or capturable or rebinding or downgrades or td_target_says or graph_replay'
for rep in 1 2; do
for m in 1 2 3 0; do
  echo "== rep $rep TDMPC2_DEBUG_SIDE_DESTROY=$m" >> $OUT
  TDMPC2_DEBUG_SIDE_DESTROY=$m timeout 400 python -m pytest tests/test_gpu_layered.py tests/test_gpu_boundary.py -m gpu -q --tb=line -p no:cacheprovider -k "$SEL" 2>&1 | grep -E "passed|failed|Error|error|^/|assert" | head -12 >> $OUT
done
done
cat $OUT
