cd /root/repo
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_layered.py -q -m gpu -x --timeout 600 -p no:cacheprovider -k "normed_linear or golden or fused_epilogue" 2>&1 | tail -4) > gpurun_out/r03d_layered_tests.txt
tail -3 gpurun_out/r03d_layered_tests.txt
export TMPDIR=/tmp
for fl in 1; do
  for spec in "c3 30 3" "c4 8 2"; do
    set -- $spec
    tag=r03d_$1_fl$fl
    (cd /tmp && TDMPC2_FUSE_LN=$fl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_$tag -o $tag -- python /root/repo/bench.py --config $1 --envs $2 --steps $3 --warmup 1 --skip-cpu-baseline --skip-extra-configs --skip-traffic > /dev/null 2>&1)
    KT=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
    python tools/rocprof_summary.py $KT > gpurun_out/${tag}_kernel_stats_by_grid.txt
    rm -rf gpurun_out/prof_$tag
    grep "g_gemm_s" gpurun_out/${tag}_kernel_stats_by_grid.txt | head -8 | cut -c1-150
  done
done
out=gpurun_out/r03d_fuse_ln_ab.txt; : > $out
for fl in 0 1 0 1; do
  for spec in "c3 30 8" "c4 8 4"; do
    set -- $spec
    echo "== TDMPC2_FUSE_LN=$fl $1 E=$2" >> $out
    TDMPC2_FUSE_LN=$fl timeout 300 python bench.py --config $1 --envs $2 --steps $3 --warmup 2 --skip-cpu-baseline --skip-extra-configs --skip-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plans/s', d['value'], 'stage_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'lat1', d['extra'].get('latency_ms_single_env'))" >> $out 2>&1
  done
done
cat $out
