// Translation unit of the fused 512-wide kernel family for ONE action padding (-DTU_APAD=16|32|48|64): ks_setup, ks_pitraj,
// ks_rollout and ks_value of fused_kernels.cuh in both arithmetics, behind the FusedOps table of launch.h.  build.sh compiles
// the four paddings in parallel; an experiment on these kernels rebuilds one of them (TDMPC2_ONLY_APAD).
#include "launch.h"

#ifndef TU_APAD
#error "compile with -DTU_APAD=16|32|48|64"
#endif

namespace {
#include "fused_kernels.cuh"

constexpr int AP = TU_APAD;

void setup_(int ar, const SetupParamsT<NetS> &p, int E, size_t lds, hipStream_t st) {
    if (ar) hipLaunchKernelGGL((ks_setup<AP, 1>), dim3(E), dim3(NTHREADS), lds, st, p);
    else hipLaunchKernelGGL((ks_setup<AP, 0>), dim3(E), dim3(NTHREADS), lds, st, p);
}
void pitraj_(int ar, int nst, const PiTrajParamsT<NetS> &p, int E, size_t lds, hipStream_t st) {
    if (nst == 1) {
        if (ar) hipLaunchKernelGGL((ks_pitraj<AP, 1, 1>), dim3(E), dim3(NTHREADS), lds, st, p);
        else hipLaunchKernelGGL((ks_pitraj<AP, 1, 0>), dim3(E), dim3(NTHREADS), lds, st, p);
    } else {
        if (ar) hipLaunchKernelGGL((ks_pitraj<AP, 2, 1>), dim3(E), dim3(NTHREADS), lds, st, p);
        else hipLaunchKernelGGL((ks_pitraj<AP, 2, 0>), dim3(E), dim3(NTHREADS), lds, st, p);
    }
}
template <int AR>
void rollout_ar(int nst, int ep, int tracing, const RolloutParamsT<NetS> &p, int grid, size_t lds, hipStream_t st) {
    if (nst == 2) {
        if (ep) hipLaunchKernelGGL((ks_rollout<AP, 2, 8, AR, 1>), dim3(grid), dim3(NTHREADS), lds, st, p);
        else if (tracing) hipLaunchKernelGGL((ks_rollout<AP, 2, 8, AR, 0, 1>), dim3(grid), dim3(NTHREADS), lds, st, p);
        else hipLaunchKernelGGL((ks_rollout<AP, 2, 8, AR, 0>), dim3(grid), dim3(NTHREADS), lds, st, p);
    } else {
        if (ep) hipLaunchKernelGGL((ks_rollout<AP, 1, 8, AR, 1>), dim3(grid), dim3(NTHREADS), lds, st, p);
        else hipLaunchKernelGGL((ks_rollout<AP, 1, 8, AR, 0>), dim3(grid), dim3(NTHREADS), lds, st, p);
    }
}
void rollout_(int ar, int nst, int ep, int tracing, const RolloutParamsT<NetS> &p, int grid, size_t lds, hipStream_t st) {
    if (ar) rollout_ar<1>(nst, ep, tracing, p, grid, lds, st);
    else rollout_ar<0>(nst, ep, tracing, p, grid, lds, st);
}
void value_(int ar, const ValueParamsT<NetS> &p, int grid, size_t lds, hipStream_t st) {
    if (ar) hipLaunchKernelGGL((ks_value<AP, 1>), dim3(grid), dim3(NTHREADS), lds, st, p);
    else hipLaunchKernelGGL((ks_value<AP, 0>), dim3(grid), dim3(NTHREADS), lds, st, p);
}
template <int AR>
int set_lds_ar(int episodic, size_t b) {
    return set_lds(ks_setup<AP, AR>, b) || set_lds(ks_pitraj<AP, 2, AR>, b) || set_lds(ks_pitraj<AP, 1, AR>, b) ||
           set_lds(ks_value<AP, AR>, b) ||
           (episodic ? (set_lds(ks_rollout<AP, 2, 8, AR, 1>, b) || set_lds(ks_rollout<AP, 1, 8, AR, 1>, b))
                     : (set_lds(ks_rollout<AP, 2, 8, AR, 0>, b) || set_lds(ks_rollout<AP, 1, 8, AR, 0>, b) ||
                        set_lds(ks_rollout<AP, 2, 8, AR, 0, 1>, b)));
}
int set_lds_(int ar, int episodic, size_t b) { return ar ? set_lds_ar<1>(episodic, b) : set_lds_ar<0>(episodic, b); }
}  // namespace

#define TDK_CAT_(a, b) a##b
#define TDK_CAT(a, b) TDK_CAT_(a, b)
namespace tdk {
const FusedOps &TDK_CAT(fused_ops_ap, TU_APAD)() {
    static const FusedOps ops = {setup_, pitraj_, rollout_, value_, set_lds_};
    return ops;
}
}
