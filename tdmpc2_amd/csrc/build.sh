#!/bin/bash
# Build the planner's C-ABI shared library for gfx950 (cross-compiles without a GPU).
# One translation unit per kernel family (and, for the fused 512-wide family, per action padding), compiled in parallel and
# linked into ONE libtdmpc2_plan.so:
#   tdmpc2_plan.hip            C ABI, handle, bind / pack kernels, refit, encoder, host side of the fused family
#   k_fused.hip   x {16,32,48,64}   ks_setup / ks_pitraj / ks_rollout / ks_value
#   k_cluster.hip x {16,32,48,64}   ks_rollout_cl
#   k_layered.hip              layered GEMMs + row kernels + their host orchestration
# Objects are cached under build/ and rebuilt when a source they include is newer (make-style), so an experiment on one
# family recompiles one file.  TDMPC2_EXTRA_FLAGS (all units), TDMPC2_FLAGS_<unit> (one unit: main, fused, cluster, layered),
# TDMPC2_ONLY_APAD=48 (experiment builds: the fused / cluster units of one padding only), TDMPC2_OUT, TDMPC2_BUILD_DIR, JOBS.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${TDMPC2_OUT:-${HERE}/../libtdmpc2_plan.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
BUILD="${TDMPC2_BUILD_DIR:-${HERE}/build}"
JOBS="${JOBS:-$(nproc)}"
mkdir -p "${BUILD}"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${TDMPC2_EXTRA_FLAGS:-}"
APADS="${TDMPC2_ONLY_APAD:-16 32 48 64}"
ONLY=""
[ -n "${TDMPC2_ONLY_APAD:-}" ] && ONLY="-DTDMPC2_ONLY_APAD=${TDMPC2_ONLY_APAD}"

# unit name | source | extra flags
UNITS=()
UNITS+=("main|tdmpc2_plan.hip|${ONLY} ${TDMPC2_FLAGS_main:-}")
# the C-ABI unit once more with the test hooks compiled in (TDMPC2_CLUSTER_FAULT, TDMPC2_DEBUG_NO_TURN, TDMPC2_POISON): linked with the
# SAME kernel objects into libtdmpc2_plan_hooks.so, which only the GPU tests of the fault paths load (tdmpc2_amd/native.py)
[ -z "${TDMPC2_NO_HOOKS_LIB:-}" ] && UNITS+=("main_hooks|tdmpc2_plan.hip|${ONLY} -DTDMPC2_TEST_HOOKS ${TDMPC2_FLAGS_main:-}")
UNITS+=("layered|k_layered.hip|${TDMPC2_FLAGS_layered:-}")
for ap in ${APADS}; do
    UNITS+=("fused${ap}|k_fused.hip|-DTU_APAD=${ap} ${TDMPC2_FLAGS_fused:-}")
    UNITS+=("cluster${ap}|k_cluster.hip|-DTU_APAD=${ap} ${TDMPC2_FLAGS_cluster:-}")
done

compile_unit() {  # name src flags
    local name="$1" src="$2" flags="$3" obj="${BUILD}/$1.o" stamp="${BUILD}/$1.flags"
    local want="${COMMON} ${flags}"
    local need=0
    [ -f "${obj}" ] || need=1
    [ -f "${stamp}" ] && [ "$(cat "${stamp}")" == "${want}" ] || need=1
    if [ "${need}" == 0 ]; then
        for dep in "${HERE}"/*.hip "${HERE}"/*.cuh "${HERE}"/*.h "${HERE}/../../include/tdmpc2_plan.h"; do
            if [ "${dep}" -nt "${obj}" ] && grep -q "$(basename "${dep}")" <(unit_deps "${src}"); then need=1; break; fi
        done
    fi
    if [ "${need}" == 1 ]; then
        "${HIPCC}" ${want} -c -o "${obj}" "${HERE}/${src}"
        echo "${want}" > "${stamp}"
        echo "compiled ${name}"
    fi
}
unit_deps() {  # transitive quoted includes of a source (by name)
    local seen=" $1 " queue="$1" f inc
    while [ -n "${queue}" ]; do
        f="${queue%% *}"; queue="${queue#"${f}"}"; queue="${queue# }"
        for inc in $(grep -ho '#include "[^"]*"' "${HERE}/${f}" 2>/dev/null | sed 's/#include "\(.*\)"/\1/' | xargs -n1 basename 2>/dev/null); do
            case "${seen}" in *" ${inc} "*) ;; *) seen="${seen}${inc} "; queue="${queue} ${inc}";; esac
        done
    done
    echo "${seen}"
}
export -f compile_unit unit_deps
export HERE BUILD HIPCC COMMON

printf '%s\n' "${UNITS[@]}" | xargs -P "${JOBS}" -I{} bash -c 'IFS="|" read -r n s f <<< "{}"; compile_unit "$n" "$s" "$f"'
OBJS=(); HOBJS=()
for u in "${UNITS[@]}"; do
    n="${u%%|*}"
    [ "$n" == "main_hooks" ] && { HOBJS+=("${BUILD}/$n.o"); continue; }
    OBJS+=("${BUILD}/$n.o")
    [ "$n" != "main" ] && HOBJS+=("${BUILD}/$n.o")
done
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC -o "${OUT}" "${OBJS[@]}"
echo "built ${OUT}"
if [ -z "${TDMPC2_NO_HOOKS_LIB:-}" ]; then
    "${HIPCC}" --offload-arch=gfx950 -shared -fPIC -o "${OUT%.so}_hooks.so" "${HOBJS[@]}"
    echo "built ${OUT%.so}_hooks.so"
fi
