#!/bin/bash
# Build the planner's C-ABI shared library for gfx950 (cross-compiles without a GPU).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${TDMPC2_OUT:-${HERE}/../libtdmpc2_plan.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"${HIPCC}" --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wall -Wno-unused-function \
    ${TDMPC2_EXTRA_FLAGS:-} -o "${OUT}" "${HERE}/tdmpc2_plan.hip"
echo "built ${OUT}"
