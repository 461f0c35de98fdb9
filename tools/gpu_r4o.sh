#!/bin/bash
# r03k hunt, second step: which preceding test makes the first eager plan of a fresh fused handle come back as garbage
# (tools/gpu_r4n.sh: it does so whatever happens to the side stream at destroy), and which runtime switches hide it.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r4o_bisect.txt; : > $OUT
CAP='tests/test_gpu_boundary.py::test_plan_is_hip_graph_capturable'
run() {  # label, env..., -- pytest args
  local label="$1"; shift
  echo "== $label" >> $OUT
  env "$@" 2>&1 | grep -E "passed|failed|^E   Assertion|Error" | cut -c1-400 | head -6 >> $OUT
}
PY="timeout 300 python -m pytest -m gpu -q --tb=short -p no:cacheprovider"
run "capturable alone" $PY "$CAP"
for t in test_fused_epilogue_wait_that_never_completes_is_reported_not_hung test_sharded_plan_reports_a_wait_that_gave_up_in_an_earlier_iteration \
         test_a_reported_wait_downgrades_the_handle_and_clean_calls_rearm_it test_td_target_says_nan_when_a_wait_gave_up \
         test_graph_replay_after_a_smaller_eager_call_resets_every_arrival_counter test_two_chains_in_flight_never_starve_each_other \
         test_layered_errors_are_loud; do
  run "$t + capturable" $PY "tests/test_gpu_layered.py::$t" "$CAP"
done
SEL='wait or capturable or rebinding or downgrades or td_target_says or graph_replay'
ALL="tests/test_gpu_layered.py tests/test_gpu_boundary.py -k"
run "all six + capturable" $PY $ALL "$SEL"
run "all six + capturable, AMD_SERIALIZE_KERNEL=3" AMD_SERIALIZE_KERNEL=3 $PY $ALL "$SEL"
run "all six + capturable, GPU_MAX_HW_QUEUES=1" GPU_MAX_HW_QUEUES=1 $PY $ALL "$SEL"
run "all six + capturable, TDMPC2_POISON=1" TDMPC2_POISON=1 $PY $ALL "$SEL"
run "all six + capturable, HIP_LAUNCH_BLOCKING=1" HIP_LAUNCH_BLOCKING=1 $PY $ALL "$SEL"
run "all six + capturable, TDMPC2_ONE_STREAM=1" TDMPC2_ONE_STREAM=1 $PY $ALL "$SEL"
cat $OUT
