#!/bin/bash
# round 4: new robustness tests (fault re-arm, NaN td_target, graph replay after a smaller call, two-chain stress)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_layered.py -q --tb=short -p no:cacheprovider -s -k "rearm or says_nan or graph_replay or starve or reported_not_hung or earlier_iteration") > gpurun_out/r4h_pytest.log 2>&1
grep -E "passed|failed|stress|Error|error|assert" gpurun_out/r4h_pytest.log | tail -30
