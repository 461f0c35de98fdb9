#!/bin/bash
# (1) the two-chain stress test on both geometries, long form for the 317M model (XCD rectangles: a row block on two XCDs);
# (2) the side chain's stream at a higher / lower priority than the caller's.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(TDMPC2_STRESS_STAGES=2500 timeout 600 python -m pytest tests/test_gpu_layered.py -m gpu -q -s --tb=short -p no:cacheprovider -k never_starve) 2>&1 | grep -E "stress|passed|failed|Error" > gpurun_out/r4u_stress.txt
cat gpurun_out/r4u_stress.txt
bash tools/gpu_env_ab.sh r4u "c3 30 44;c4 8 13" "A=0" "TDMPC2_SIDE_PRIO=high" "TDMPC2_SIDE_PRIO=low" > /dev/null; cat gpurun_out/r4u_ab.txt
