cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for fl in 1 0; do
  for spec in "c3 30 3" "c4 8 2"; do
    set -- $spec
    tag=r03c_$1_fl$fl
    (cd /tmp && TDMPC2_FUSE_LN=$fl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_$tag -o $tag -- python /root/repo/bench.py --config $1 --envs $2 --steps $3 --warmup 1 --skip-cpu-baseline --skip-extra-configs --skip-traffic > /dev/null 2>&1)
    KT=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
    python tools/rocprof_summary.py $KT > gpurun_out/${tag}_kernel_stats_by_grid.txt
    rm -rf gpurun_out/prof_$tag
    head -14 gpurun_out/${tag}_kernel_stats_by_grid.txt | cut -c1-150
  done
done
