#!/bin/bash
# rocprofv3 kernel stats of the round's final library: the default bench line including the c3 / c4 legs
R="${GRAFT_REPO_ROOT:-$PWD}"; export TMPDIR=/tmp; cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02ad -o r02ad -- python $R/bench.py --steps 5 --warmup 2 --skip-cpu-baseline > $R/gpurun_out/r02ad_bench_under_rocprof.json 2> $R/gpurun_out/prof_r02ad.stderr
cd $R
KT=$(find gpurun_out/prof_r02ad -name "*kernel_trace.csv" | head -1)
python tools/rocprof_summary.py $KT > gpurun_out/r02ad_kernel_stats_by_grid.txt
cp $(find gpurun_out/prof_r02ad -name "*kernel_stats.csv" | head -1) gpurun_out/r02ad_rocprofv3_kernel_stats.csv
head -24 gpurun_out/r02ad_kernel_stats_by_grid.txt | cut -c1-150
rm -rf gpurun_out/prof_r02ad
