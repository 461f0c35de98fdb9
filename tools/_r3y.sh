mkdir -p gpurun_out
for i in 1 2 3 4; do
  timeout 300 python -m pytest tests/test_gpu_dist.py -q -x -p no:cacheprovider -k "two_processes and c4" --tb=short 2>&1 | grep -E "passed|failed|AssertionError|2 ranks|bounded waits" | head -5 >> gpurun_out/r3y_flake.txt
done
timeout 300 python -m pytest tests/test_gpu_layered.py -q -p no:cacheprovider -k "sharded_plan_reports or wait_that_never" --tb=short 2>&1 | tail -15 >> gpurun_out/r3y_flake.txt
cat gpurun_out/r3y_flake.txt
bash tools/gpu_env_ab.sh r3y "c3 30 8" "A=0" "TDMPC2_GEMM_WIDE_SD=2" "TDMPC2_GEMM_WIDE_PF=3" > /dev/null; cat gpurun_out/r3y_ab.txt
AB_REPS=1 bash tools/gpu_env_ab.sh r3yb "c4 8 3" "A=0" "TDMPC2_GEMM_WIDE_SD=2" "TDMPC2_GEMM_WIDE_PF=3" > /dev/null; cat gpurun_out/r3yb_ab.txt
