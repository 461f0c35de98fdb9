#!/bin/bash
# Ablation timing of the split rollout kernel: build variants with parts of the kernel disabled (results are wrong by
# construction; only the timing is read) and time the same bench step with each.  Run: build here (no GPU needed),
# then `gpurun -- bash tools/ablate.sh run`.
set -u
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
if [ -n "${ABLATE_VARIANTS:-}" ]; then
  IFS=';' read -r -a VARIANTS <<< "$ABLATE_VARIANTS"
else
  # (the no-MFMA / no-weight-load / no-epilogue ablations of rounds 1-4 need tools/variants/r1_r4_ablation_hooks.patch applied first:
  # the shipped kernels carry no experiment switches; -DSPLIT_TIMING / -DGW_TIMING, the phase clocks, are still there)
  VARIANTS=("full:" "timing:-DSPLIT_TIMING" "full2:")
fi
mkdir -p "$R/build/ablate"
if [ "${1:-build}" = "build" ]; then
  for v in "${VARIANTS[@]}"; do
    name="${v%%:*}"; flags="${v#*:}"
    TDMPC2_OUT="$R/build/ablate/lib_${name}.so" TDMPC2_EXTRA_FLAGS="$flags" "$R/tdmpc2_amd/csrc/build.sh" > /dev/null 2>&1 &
  done
  wait
  ls -la "$R/build/ablate/"
else
  for v in "${VARIANTS[@]}"; do
    name="${v%%:*}"
    TDMPC2_TIMING=1 TDMPC2_PLAN_LIB="$R/build/ablate/lib_${name}.so" timeout 120 python "$R/bench.py" --steps 5 --warmup 2 --skip-cpu-baseline ${ABLATE_BENCH_ARGS:-} 2> >(grep "tdmpc2_plan timing" >&2) \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'plans/s', d['value'], 'rollout_ms', d['roofline']['avg_launch_ms'], 'lat1_ms', d['extra'].get('latency_ms_single_env'))"
  done
fi
