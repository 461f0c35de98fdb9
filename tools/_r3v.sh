mkdir -p gpurun_out
bash tools/gpu_env_ab.sh r3v "c3 30 8;c4 8 3" "A=0" "TDMPC2_GEMM_XCD_ROWS=1" > /dev/null; cat gpurun_out/r3v_ab.txt
for e in "A=0" "TDMPC2_GEMM_XCD_ROWS=1"; do
  echo "== traffic [$e] c3" >> gpurun_out/r3v_traffic.txt
  env $e timeout 300 python bench.py --config c3 --envs 30 --steps 3 --warmup 1 --skip-cpu-baseline --skip-extra-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['traffic'], d['roofline']['traffic_detail'])" >> gpurun_out/r3v_traffic.txt 2>&1
done
cat gpurun_out/r3v_traffic.txt
TDMPC2_GEMM_XCD_ROWS=1 timeout 600 python -m pytest tests/test_gpu_layered.py tests/test_gpu_philox.py -q -x -p no:cacheprovider --tb=short 2>&1 | tail -5 > gpurun_out/r3v_tests_xcd_rows.txt; cat gpurun_out/r3v_tests_xcd_rows.txt
