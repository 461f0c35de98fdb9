// Host-side declarations shared by the translation units of libtdmpc2_plan.so: the handle, the packed-weight records, and
// the launcher interfaces through which tdmpc2_plan.hip (C ABI, bind, fused-family host code) reaches kernels that are
// instantiated in other translation units (k_fused.hip / k_cluster.hip per action padding, k_layered.hip).
#pragma once
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "common.cuh"

namespace tdk {

// error reporting (tdmpc2_last_error): defined in tdmpc2_plan.hip
int fail(int code, const char *fmt, ...);
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) return fail(TDMPC2_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
#define LAUNCH_CHECK()                                                                       \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess) return fail(TDMPC2_ERR_HIP, "launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

struct HostLayer {
    float *wp = nullptr, *bias = nullptr, *g = nullptr, *b = nullptr, *wemb = nullptr;
    // f16x2-split form (fused_kernels.cuh): hi/lo packed weights, per-matrix power-of-two scales (device scalars)
    _Float16 *wps = nullptr;
    LayerScal *scal = nullptr;   // split arithmetic: this layer's record inside the net's [heads][3] table
    float *oscale = nullptr, *ascale = nullptr;  // -> scal->oscale / ascale (split) or the unit scalar (exact fp32)
    int KB = 0, CT = 0, out = 0;  // KB: k-blocks of 8 (fp32 MFMA) or of 16 (split)
    bool bound = false, alloc = false;
    size_t wbytes = 0;            // bytes of this layer's packed weights (tdmpc2_plan_export_packed)
};
struct HostNet {
    HostLayer l[3];
    LayerScal *scal = nullptr;   // [heads][3] (heads of an ensemble share one allocation: stride 3 records)
};

}  // namespace tdk

// Measurement knobs of the layered family's tile choice (layered_host.cuh).  They used to be environment variables read inside the
// shipped library (28 names, an A/B surface that could silently change a deployment); now they are fields of the
// handle, reachable ONLY through tdmpc2_plan_set_tuning(h, TDMPC2_TUNE_EXPERT + id, value) -- the Python binding maps TDMPC2_X_<NAME>
// of its own environment onto that call for the A/B tools (tdmpc2_amd/native.py), the library itself reads none of them.
// Order = enum tdmpc2_expert_knob of include/tdmpc2_plan.h.
enum LayKnob { LK_W256_MIN, LK_W_SPLIT_MIN, LK_W_SPLIT_MAX, LK_W_SPLIT_OVH, LK_KSPLIT_AUTO_LO, LK_KSPLIT_AUTO_MIN, LK_W_XCD_ROWS, LK_NCT1,
               LK_WIDE_MIN, LK_RT4, LK_FILL_PERMILLE, LK_FILL_HEAD_PERMILLE, LK_SD1, LK_XCD_ROWS, LK_COL_PAD, LK_TWOHOT_UNFUSED,
               LK_Z0_SHARED_OFF, LK_MID_PARTS_MAX, LK_MID_FUSE_LN, LK_MID_SPLIT_XCD, LK_MID_PIFOLD, LK_COUNT };
constexpr int LAY_KNOB_DEFAULTS[LK_COUNT] = {192, 192, 4, 12000, 16, -1 /* cus / 4 */, -1, 0, 128, 0, 750, 750, 0, -1, 1, 0, 0, 16, 1, 1, 1};

// Workspace of the layer-at-a-time path (layered_kernels.cuh): activations of all E*N sample rows in HBM.
struct Layered {
    int knob[LK_COUNT] = {192, 192, 4, 12000, 16, -1, -1, 0, 128, 0, 750, 750, 0, -1, 1, 0, 0, 16, 1, 1, 1};
    bool on = false;
    int Kin = 0;    // row stride of X = first-layer K: round_up(L + A, 32)
    int Mp = 0;     // mlp_dim (multiple of 32)
    int ldl = 0;    // row stride of the head-logit buffer
    int Ppad = 0;   // rows per plan in the policy-prior pass: round_up(P, 32)
    float *X = nullptr, *HA = nullptr, *HB = nullptr, *LG = nullptr, *G = nullptr, *QT = nullptr, *TERM = nullptr;
    int *qidx = nullptr;  // [E, 2] heads of the current iteration
    // what the GEMM / row helpers of layered_host.cuh read besides their arguments (lay_value re-points them for a call)
    const HostNet *qarr = nullptr;   // the Q ensemble in use: online (planning) or target (td_target)
    const float *bias_tab = nullptr; // effective first-layer biases: per plan (h->beff) or per task (h->beff_tab)
    const int *row_env = nullptr;    // per-row env of the bias / mask lookups, or null: row / rows_per_env
    // LayerNorm + activation inside the GEMM epilogue (split arithmetic, g_gemm_s<.., EPI>): the exchange of per-row
    // (mean, M2) partials between the column blocks of a row block, and the arrival counters of a stage's fused launches
    // a second stream + buffer set for a second chain of GEMMs in flight (lay_estimate_value)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_side = nullptr, ev_xread = nullptr;
    float *HA2 = nullptr, *HB2 = nullptr, *LG2 = nullptr, *stats2 = nullptr;
    // split arithmetic: fp32 pre-activations of a NormedLinear whose epilogue is not fused ([rows, ldpre], one per chain)
    float *PRE = nullptr, *PRE2 = nullptr;
    int ldpre = 0;
    // t = 0 of every rollout: the z0 products of the reward / dynamics first layers, one vector per plan (lay_cvec)
    float *Z0X = nullptr, *cvec = nullptr;
    size_t cvec_rows = 0;
    bool cvec_ready = false;
    bool fuse_ln = false;
    float *stats = nullptr;
    size_t stats_cap = 0;            // floats
    unsigned int *arrive = nullptr;
    size_t arrive_cap = 0, arrive_off = 0;  // counters; the next free one (zeroed at the start of every stage)
    unsigned long long *gw_timing = nullptr;  // TDMPC2_GW_TIMING=1 with a -DGW_TIMING build: phase clocks of g_gemm_w, 4 classes x 8 words
    size_t arrive_high = 0;                 // most counters any stage of this handle has used (the extent of that memset)
    // lay_run -> lay_estimate_value_m, iteration 0 of a single few-row plan: the policy-prior rows' actions a_t = pi(z_t) are computed at
    // the top of step t of the stage itself (their latents ARE rows of the stage), instead of a pass of their own in front (lay_pitraj)
    bool pifold = false;
    const float *pifold_eps = nullptr;      // the tape's pi_traj_eps [E, H, P, A], or null (Philox)
    bool arrive_clean = false;              // l_sample has just zeroed the counters (lay_arrive_reset then has nothing to do)
    bool arrive_pending = false;            // few-row stage: the counters have not been zeroed yet (done in front of the first launch that waits)
    // g_gemm_w's K-split tail (layered_wide.cuh): partial accumulators of the split tiles, one workspace per chain
    float *ksws = nullptr, *ksws2 = nullptr;
    size_t ksws_slots = 0;                  // 256 KiB slots (split tile x part) per workspace
    size_t ks_tiles = 0;                    // 256 x 256 tiles of the handle's largest call (0: the wide tile never applies)
    int ksplit = 2;                         // TDMPC2_TUNE_KSPLIT: 0 never, 1 whenever the rule says so, 2 few-tile launches only (layered_host.cuh)
    // the few-row path (layered_mid.cuh): K-part tiles + row kernels for single plans; partial-sum workspaces of a launch's two problems
    bool mid = true;                        // TDMPC2_TUNE_FEWROW (and TDMPC2_TUNE_KSPLIT != 0)
    float *mws[2] = {nullptr, nullptr};
    size_t mws_cap = 0;                     // floats per workspace
};

struct tdmpc2_plan {
    tdmpc2_plan_cfg cfg;
    Layered lay;
    std::atomic<int> busy{0};  // handles are not reentrant: a second concurrent call is refused (Busy), not raced
    int fold_refit = 2;        // fused family: the last workgroup of a plan refits it inside the rollout launch (2 = auto)
    unsigned int *ticket = nullptr;  // [max_envs] arrival counters of that hand-over
    int *qidx_buf = nullptr;         // [max_envs, 2] the two Q heads of the current iteration (shard_values)
    unsigned int shard_call = 0;     // call counter captured by shard_begin (Philox stream of the sharded plan)
    bool in_shard = false;           // between shard_begin and the last shard_refit: no re-arm in between; word 0 of the error line is not cleared in between
    // per-task tables of policy_value / td_target on multitask batches (grown on demand)
    float *beff_tab = nullptr, *mask_tab = nullptr, *disc_tab = nullptr;
    int *task_rows = nullptr;  // [rows] copy of the row -> task map, padded to whole GEMM tiles (layered family)
    int tab_tasks = 0;
    size_t task_rows_cap = 0;
    // cluster path of the fused family (cluster_kernels.cuh): single-plan latency
    int cluster_mode = 2;            // TDMPC2_TUNE_CLUSTER: 0 never, 1 whenever the call fits, 2 auto (= 1 today)
    int cl_max_clusters = 0;         // clusters the buffers below were sized for (0: path not available on this handle)
    float *cl_xbuf = nullptr, *cl_zs = nullptr;
    unsigned int *cl_flags = nullptr;
    unsigned int *cl_err_host = nullptr, *cl_err_dev = nullptr;  // host-mapped error word of the bounded waits
    size_t cl_lds = 0;
    // ks_rollout_cl2 (cluster2_kernels.cuh): two clusters per tile for ONE plan -- buffers for 2 x tiles-of-32 clusters, or null
    float *cl2_xbuf = nullptr, *cl2_zs = nullptr, *cl2_mail = nullptr;
    unsigned int *cl2_flags = nullptr;
    int cl2_mode = 1;                // TDMPC2_CLUSTER2=0: off
    int cl_fault = 0;                // TDMPC2_CLUSTER_FAULT=1 at create: test hook of the bounded waits
    int faults = 0;                  // cluster plans that gave up since the last tdmpc2_plan_take_fault
    // Recovery from a reported wait (fault_note / fault_clean, tdmpc2_plan.hip): the paths with inter-workgroup waits are switched
    // off when a wait gives up and switched back on after `rearm_after` consecutive clean calls (doubling, up to 4096, every time
    // a fault follows a re-arm: a box that keeps faulting -- co-tenancy -- settles on the paths without waits)
    bool degraded = false;
    bool safe_once = false;          // TDMPC2_TUNE_SAFE_ONCE: the next whole plan runs without inter-workgroup waits (apply_modes)
    int user_cluster_mode = 2;       // what the caller / environment asked for (apply_modes derives cluster_mode / lay.fuse_ln from it)
    bool user_fuse_ln = false;
    int rearm_after = 8, rearm_base = 8;     // 0: never re-arm (the round-3 behaviour)
    int clean_calls = 0;             // consecutive calls without a fault since the downgrade
    int faults_total = 0, rearms = 0;
    std::chrono::steady_clock::time_point last_fault{};
    // stream hand-over (StreamTurn, tdmpc2_plan.hip): the event every call leaves on its stream; a call on another stream waits for it
    hipEvent_t turn_ev = nullptr;
    hipStream_t turn_stream = nullptr;
    bool turn_valid = false;
    bool split = false;  // fused kernels on the f16 matrix pipe with hi/lo operand split (fused_kernels.cuh)
    int force_rows = 0;  // TDMPC2_TUNE_ROWS_PER_WORKGROUP: 0 auto, 32, 64
    size_t row_bytes = 0;  // bytes of one sample row of the fused kernels' LDS tile
    float *one = nullptr;  // device scalar 1.0f: the output scale of the exact-fp32 arithmetic
    int Apad = 0, stride = 0, tiles = 0, nnets = 0;
    int num_cus = 0;  // compute units of cfg.device
    size_t lds_bytes = 0;
    HostNet dyn, rew, pi, term;
    HostNet q[MAXQ];
    HostNet tq[MAXQ];  // target ensemble (optional: tdmpc2_plan_td_target)
    std::vector<void *> allocs;
    uint64_t bytes = 0;
    // workspace
    float *bins = nullptr, *actions = nullptr, *value = nullptr, *mean = nullptr, *std = nullptr, *cvec = nullptr,
          *beff = nullptr, *zscratch = nullptr;
    // state-observation encoder (optional: bound with tdmpc2_plan_bind_encoder)
    struct Enc {
        float *wt = nullptr, *bias = nullptr, *g = nullptr, *b = nullptr;
        int in = 0, out = 0;
        bool bound = false;
    } enc[6];
    int enc_layers = 0;
    float *zenc = nullptr;  // [max_envs, L]: latents of tdmpc2_plan_run_obs
    float *enc_y = nullptr, *enc_x = nullptr;  // wide encoders: [max_envs, widest layer] pre-activations / activations
    int enc_ws_width = 0;
    unsigned int call = 0;
    unsigned long long *timing = nullptr;  // TDMPC2_TIMING=1 with a -DSPLIT_TIMING build: in-kernel phase cycle counters
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> ev;
    int ev_used = 0;
};
