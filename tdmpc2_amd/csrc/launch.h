// Launcher interfaces between the translation units of libtdmpc2_plan.so (see common.cuh).  A kernel family is instantiated
// in exactly one translation unit; everything else reaches it through these functions / tables.
#pragma once
#include "handle.h"

namespace tdk {

// ---- fused 512-wide family: one table per action padding (k_fused.hip, compiled once per -DTU_APAD=16|32|48|64).
// ar: 0 = f16x2 split, 1 = exact fp32 MFMA; nst: 32-row sample tiles per workgroup (1 | 2).
struct FusedOps {
    void (*setup)(int ar, const SetupParamsT<NetS> &p, int E, size_t lds, hipStream_t st);
    void (*pitraj)(int ar, int nst, const PiTrajParamsT<NetS> &p, int E, size_t lds, hipStream_t st);
    void (*rollout)(int ar, int nst, int ep, int tracing, const RolloutParamsT<NetS> &p, int grid, size_t lds, hipStream_t st);
    void (*value)(int ar, const ValueParamsT<NetS> &p, int grid, size_t lds, hipStream_t st);
    int (*set_lds)(int ar, int episodic, size_t lds_bytes);  // hipFuncAttributeMaxDynamicSharedMemorySize of every instantiation
};
// ---- cluster path of the fused family (k_cluster.hip, per action padding; split arithmetic only)
struct ClusterOps {
    void (*rollout_cl)(int ep, const RolloutParamsT<NetS> &p, int grid, size_t lds, hipStream_t st);
    int (*set_lds)(int episodic, size_t lds_bytes);
    // two clusters per 32-row tile (reward chain beside the dynamics chain): single non-episodic plans, every launch (launch 0 with the policy-prior fold)
    void (*rollout_cl2)(const RolloutParamsT<NetS> &p, int grid, size_t lds, hipStream_t st);
};
// (accessor functions, not global tables: hipcc would emit a constant-initialised table on the device side as well)
const FusedOps &fused_ops_ap16(); const FusedOps &fused_ops_ap32(); const FusedOps &fused_ops_ap48(); const FusedOps &fused_ops_ap64();
const ClusterOps &cluster_ops_ap16(); const ClusterOps &cluster_ops_ap32(); const ClusterOps &cluster_ops_ap48(); const ClusterOps &cluster_ops_ap64();

template <typename K>
inline int set_lds(K kernel, size_t bytes) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

// ---- elite selection + refit, one workgroup per plan (k_refit, tdmpc2_plan.hip)
int launch_refit(const RefitParams &fp, int E, int N, size_t lds, hipStream_t st);

// ---- layer-at-a-time family (k_layered.hip: kernels + their host orchestration, layered_host.cuh)
int lay_setup(tdmpc2_plan *h, hipStream_t st, int E, const float *task_emb, const float *prev_mean, const unsigned char *t0,
              bool init_dist, float *beff_out = nullptr, const HostNet *qarr = nullptr);
int lay_cvec(tdmpc2_plan *h, hipStream_t st, int E, const float *z0);
int lay_pitraj(tdmpc2_plan *h, hipStream_t st, int E, const float *z0, const float *act_mask, const float *tape_eps,
               unsigned long long seed, unsigned call);
int lay_estimate_value(tdmpc2_plan *h, hipStream_t st, int E, const float *z0, const float *act_mask, const float *disc_pow,
                       const float *actions, const float *pi_eps, long pi_eps_estride, const int *qidx /* dense [E,2] */,
                       unsigned long long seed, unsigned call, int iter, float *value, float *trace, int n_off = 0, int n_sub = 0);
int lay_run(tdmpc2_plan *h, hipStream_t st, int E, const float *z0, const float *task_emb, const float *act_mask,
            const float *disc_pow, float *prev_mean, const uint8_t *t0, int eval_mode, const tdmpc2_noise *tape, uint64_t seed,
            float *action, const tdmpc2_debug *dbg);
int lay_value(tdmpc2_plan *h, hipStream_t st, int rows, const float *z, bool target, bool reduce_min, const float *pi_eps,
              const int *qidx_dev /* [2] */, unsigned long long seed, unsigned call, const float *reward, const float *terminated,
              float discount, const int *row_task /* padded [rows_p] or null */, float *action, float *out);
// one CEM iteration's sampled actions (rows n >= P of h->actions, every step) and its two Q heads per plan -> qbuf [E, 2]
// (used by both families when a plan is sharded: tdmpc2_plan_shard_values)
int lay_sample_iteration(tdmpc2_plan *h, hipStream_t st, int E, int iter, const float *act_mask, const tdmpc2_noise *tape,
                         uint64_t seed, unsigned call, int *qbuf);
// the two Q heads of a single evaluation (td_target / estimate_value entry points): copied from `qidx` ([E, 2], row stride
// `stride`) or drawn (Philox) when it is null
int lay_set_qidx(tdmpc2_plan *h, hipStream_t st, int E, const int *qidx, long stride, int nq, int iter, uint64_t seed, unsigned call, int *dst);

}  // namespace tdk
