#!/bin/bash
# instruction-fetch counters of the cluster kernel (c2, E = 1): is the 160 KB straight-line kernel I-cache bound?
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/pmc_r02u"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
export CLUSTER_MODES=1,0 CLUSTER_ENVS=1 TDMPC2_PLAN_LIB=$R/build/ablate/lib_${LIBV:-sl}.so
pass() { local name="$1"; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python "$R/tools/probes/cluster_latency.py" c2 > "$OUT/$name.stdout" 2> "$OUT/$name.stderr"; echo "pass $name rc=$?"; }
pass ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
pass ic2 SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES
cd "$R"; python tools/pmc_summary.py "$OUT" ks_rollout > gpurun_out/r02u_pmc.txt 2>&1; tail -60 gpurun_out/r02u_pmc.txt
find "$OUT" -name "*counter_collection.csv" | head
