// Translation unit of the cluster path (cluster_kernels.cuh: ks_rollout_cl, single-plan latency of the fused family) for ONE
// action padding (-DTU_APAD=16|32|48|64), behind the ClusterOps table of launch.h.
#include "launch.h"

#ifndef TU_APAD
#error "compile with -DTU_APAD=16|32|48|64"
#endif

namespace {
#include "fused_kernels.cuh"
#include "cluster_kernels.cuh"
#include "cluster2_kernels.cuh"

constexpr int AP = TU_APAD;

void rollout_cl_(int ep, const RolloutParamsT<NetS> &p, int grid, size_t lds, hipStream_t st) {
    if (ep) hipLaunchKernelGGL((ks_rollout_cl<AP, 1>), dim3(grid), dim3(NTHREADS), lds, st, p);
    else hipLaunchKernelGGL((ks_rollout_cl<AP, 0>), dim3(grid), dim3(NTHREADS), lds, st, p);
}
void rollout_cl2_(const RolloutParamsT<NetS> &p, int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((ks_rollout_cl2<AP>), dim3(grid), dim3(NTHREADS), lds, st, p);
}
int set_lds_(int episodic, size_t b) {
    return episodic ? set_lds(ks_rollout_cl<AP, 1>, b) : (set_lds(ks_rollout_cl<AP, 0>, b) || set_lds(ks_rollout_cl2<AP>, b));
}
}  // namespace

#define TDK_CAT_(a, b) a##b
#define TDK_CAT(a, b) TDK_CAT_(a, b)
namespace tdk {
const ClusterOps &TDK_CAT(cluster_ops_ap, TU_APAD)() {
    static const ClusterOps ops = {rollout_cl_, set_lds_, rollout_cl2_};
    return ops;
}
}
