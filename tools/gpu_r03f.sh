cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_layered.py tests/test_gpu_td_target.py tests/test_gpu_dist.py tests/test_gpu_philox.py tests/test_gpu_packed.py -q -m gpu -x --timeout 600 -p no:cacheprovider > gpurun_out/r03f_layered_tests.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r03f_layered_tests.txt | tail -3
out=gpurun_out/r03f_z0_shared_ab.txt; : > $out
for off in 1 0 1 0; do
  for spec in "c3 30 8" "c4 8 4"; do
    set -- $spec
    echo "== TDMPC2_Z0_SHARED_OFF=$off $1 E=$2" >> $out
    if [ $off = 1 ]; then export TDMPC2_Z0_SHARED_OFF=1; else unset TDMPC2_Z0_SHARED_OFF; fi
    timeout 300 python bench.py --config $1 --envs $2 --steps $3 --warmup 2 --skip-cpu-baseline --skip-extra-configs --skip-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plans/s', d['value'], 'stage_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'lat1', d['extra'].get('latency_ms_single_env'))" >> $out 2>&1
  done
done
unset TDMPC2_Z0_SHARED_OFF
cat $out
