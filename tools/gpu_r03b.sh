cd /root/repo
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_layered.py tests/test_gpu_td_target.py -q -m gpu -x --timeout 600 -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r03b_layered_tests.txt
cat gpurun_out/r03b_layered_tests.txt | tail -8
out=gpurun_out/r03b_fuse_ln_ab.txt; : > $out
for fl in 0 1 0 1; do
  for spec in "c3 30 8" "c4 8 4"; do
    set -- $spec
    echo "== TDMPC2_FUSE_LN=$fl $1 E=$2" >> $out
    TDMPC2_FUSE_LN=$fl timeout 300 python bench.py --config $1 --envs $2 --steps $3 --warmup 2 --skip-cpu-baseline --skip-extra-configs --skip-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plans/s', d['value'], 'stage_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'lat1', d['extra'].get('latency_ms_single_env'))" >> $out 2>&1
  done
done
cat $out
