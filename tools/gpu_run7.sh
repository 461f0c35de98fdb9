#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_fold -o fold -- python $R/tools/probes/fold_probe.py
cd $R
python tools/rocprof_summary.py $(find gpurun_out/prof_fold -name "*kernel_trace.csv" | head -1) | head -12
rm -rf gpurun_out/prof_fold
