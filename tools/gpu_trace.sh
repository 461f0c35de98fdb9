#!/bin/bash
# rocprofv3 kernel trace of one bench configuration, split by kernel and grid
# usage: gpurun -- bash tools/gpu_trace.sh <tag> <config> <envs> <steps> [env assignments...]
cd "$(dirname "$0")/.."
R=$PWD; TAG=$1; CFG=$2; E=$3; K=$4; shift 4
mkdir -p gpurun_out; export TMPDIR=/tmp
(cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py --config $CFG --envs $E --steps $K --warmup 1 --skip-cpu-baseline --skip-extra-configs --skip-traffic > /dev/null 2>&1)
KT=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python tools/rocprof_summary.py $KT > gpurun_out/${TAG}_kernel_stats_by_grid.txt
rm -rf gpurun_out/prof_$TAG
head -${TRACE_HEAD:-24} gpurun_out/${TAG}_kernel_stats_by_grid.txt | cut -c1-150
