#!/bin/bash
# kernel trace of c3 / c4 single plans (layered family, few-row GEMM variants)
R="${GRAFT_REPO_ROOT:-$PWD}"; export TMPDIR=/tmp; cd /tmp
export TDMPC2_PLAN_LIB=$R/build/ablate/lib_sk.so
for cfgname in c3 c4; do
  OUT="$R/gpurun_out/prof_r03d_$cfgname"; mkdir -p "$OUT"
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python "$R/tools/probes/graph_probe.py" $cfgname > "$OUT/stdout.txt" 2>&1
  python "$R/tools/rocprof_summary.py" $(find "$OUT" -name "*kernel_trace.csv" | head -1) > "$R/gpurun_out/r03d_${cfgname}_kernels.txt" 2>&1
  echo "== $cfgname"; head -12 "$R/gpurun_out/r03d_${cfgname}_kernels.txt" | cut -c1-150
done
