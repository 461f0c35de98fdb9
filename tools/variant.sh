#!/bin/bash
# Build a VARIANT of the library for a same-call A/B (tools/gpu_ab.sh): the objects of the in-tree build are reused, only the
# translation unit(s) whose flags differ are recompiled (seconds for the layered unit, about a minute for a fused one).
# usage: tools/variant.sh <name> <unit>:"<flags>" [<unit>:"<flags>" ...]      units: main layered fused cluster
#   e.g. tools/variant.sh wr4 fused:"-DKLOOP_RING=4"   ->  build/ablate/lib_wr4.so
set -euo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME="$1"; shift
SRC="$R/tdmpc2_amd/csrc"
VB="$SRC/build/variants/$NAME"
mkdir -p "$VB" "$R/build/ablate"
for f in "$SRC"/build/*.o "$SRC"/build/*.flags; do [ -e "$f" ] && cp -u -p "$f" "$VB/"; done
ENVS=()
for spec in "$@"; do
  unit="${spec%%:*}"; flags="${spec#*:}"
  ENVS+=("TDMPC2_FLAGS_${unit}=${flags}")
done
env "${ENVS[@]}" TDMPC2_NO_HOOKS_LIB=1 TDMPC2_BUILD_DIR="$VB" TDMPC2_OUT="$R/build/ablate/lib_${NAME}.so" ${TDMPC2_ONLY_APAD:+TDMPC2_ONLY_APAD=$TDMPC2_ONLY_APAD} "$SRC/build.sh"
