#!/bin/bash
# PMC passes for the planner kernels (run on the GPU box through gpurun).  Counters are collected in their own
# runs (no --stats/--sys-trace combined with --pmc), one pass per counter group, as the MI355X guide prescribes.
# usage: tools/gpu_pmc.sh <tag> [bench args...]
set -u
TAG="${1:-r01}"; shift || true
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/pmc_${TAG}"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="${*:---steps 2 --warmup 1 --skip-cpu-baseline --skip-extra-configs}"
pass() {
  local name="$1"; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- \
      python "$R/bench.py" $ARGS > "$OUT/$name.stdout" 2> "$OUT/$name.stderr"
  echo "pass $name rc=$?"
}
PASSES="${PMC_PASSES:-sq1 sq2 grbm fetch write}"   # PMC_PASSES="sq1 grbm": a short run (bench.py measures fetch / write itself)
pass_if() { [[ " $PASSES " == *" $1 "* ]] && pass "$@"; }
pass_if sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
pass_if sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_WAVES
pass_if grbm GRBM_GUI_ACTIVE GRBM_COUNT
pass_if fetch FETCH_SIZE
pass_if write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
find "$OUT" -name "*.csv" | head -50
