#!/bin/bash
# soak of the K-split paths: golden plans of few-tile calls, repeated; stress with the K-split on
cd "$(dirname "$0")/../.."
f=0
for i in 1 2 3 4 5 6; do
  python -m pytest tests/test_gpu_layered.py -q -m gpu -x -k "matches_reference_golden and split and (c3_x4 or c4_x2 or c4_l1024 or c4-split or m19_mt80)" -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -1
done
TDMPC2_STRESS_STAGES=1500 python -m pytest tests/test_gpu_layered.py -q -m gpu -k "never_starve and (c3-30 or c4-1-)" -p no:cacheprovider 2>&1 | grep -E "passed|failed|AssertionError: [0-9]+" | tail -3
