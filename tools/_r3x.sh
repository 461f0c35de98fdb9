mkdir -p gpurun_out
out=gpurun_out/r3w_two_process_loop.txt; : > $out
for i in 1 2 3 4; do
  timeout 300 python -m pytest tests/test_gpu_dist.py -q -x -s -p no:cacheprovider -k "two_processes" --tb=short 2>&1 | grep -E "passed|failed|Error|2 ranks|re-planned" | head -8 >> $out
done
(time timeout 300 python -m pytest tests/test_gpu_layered.py tests/test_gpu_dist.py -q -p no:cacheprovider -k "sharded_plan_reports or wait_that_never or sharded" --tb=short) 2>&1 | tail -15 >> $out
cat $out
