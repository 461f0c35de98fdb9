#!/bin/bash
# device-side assembly of one build variant -> VGPRs / scratch bytes of every kernel matching a pattern
# usage: tools/kstat.sh "<extra flags>" [kernel-name regex] [translation unit: k_fused.hip (default, -DTU_APAD=48) | k_cluster.hip | k_layered.hip | tdmpc2_plan.hip]
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="${KSTAT_OUT:-/tmp/kstat.s}"
TU="${3:-k_fused.hip}"
case "$TU" in k_fused.hip|k_cluster.hip) AP="-DTU_APAD=${KSTAT_APAD:-48}";; *) AP="";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -Wno-unused-function $AP $1 -o "$out" "$R/tdmpc2_amd/csrc/$TU" || exit 1
awk -v pat="${2:-ks_rollout}" '/\.amdhsa_kernel /{k=$2} /amdhsa_private_segment_fixed_size/{s=$2} /amdhsa_next_free_vgpr/{if (k ~ pat) print k, "vgpr", $2, "scratch", s}' "$out"
