// Shared by every translation unit of libtdmpc2_plan.so: kernel parameter blocks, constants and the small device
// helpers (math, Philox, the elite refit) that more than one kernel family uses.  Types live in namespace tdk so that
// launcher functions defined in one translation unit can be called from another; the kernels themselves stay in the
// anonymous namespace of the ONE translation unit that instantiates them (k_fused.hip, k_cluster.hip, k_layered.hip,
// tdmpc2_plan.hip), which is what lets build.sh compile the families in parallel and an experiment rebuild one of them.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/tdmpc2_plan.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace tdk {

constexpr int ROWS = 64;        // sample rows per rollout workgroup
constexpr int NTHREADS = 512;   // 8 wavefronts
constexpr int WIDTH = 512;      // latent_dim == mlp_dim of the fused size class
constexpr int MAXQ = 8;
constexpr int MAXH = 8;
constexpr float LN_EPS = 1e-5f;

// layered family (layered_kernels.cuh / layered_split.cuh): GEMM tile constants the host sizes buffers with
constexpr int GBM = 128;          // rows per GEMM workgroup (the narrow tiles; the wide tile takes 256)
constexpr int GBN = 128;          // output columns per GEMM workgroup (4 column tiles of 32)
constexpr int GBK = 32;           // k-chunk staged through LDS
constexpr int GLD = GBK + 4;      // LDS row stride in floats (stride/4 = 9, odd -> conflict-free ds_read_b128)
constexpr int GTHREADS = 256;     // 4 wavefronts: 2 (rows) x 2 (cols), each 64 x 64 = 2x2 MFMA tiles
constexpr int RW_THREADS = 256;   // row kernels: 4 rows (wavefronts) per workgroup
// cluster path (cluster_kernels.cuh)
constexpr int CL = 8;                  // workgroups per cluster
constexpr int CL_SLOTS = 6;            // exchange tiles per cluster (S4 holds a head's logits, [32][128] fp32; S5: episodic models)
constexpr int CL_TILE = 32 * WIDTH;    // floats per exchange tile
constexpr int CL_FLAG_STRIDE = 16;     // arrival words reserved per cluster (64 B)


// Scale of bounded operands (SimNorm latents in [0, 1], actions in [-1, 1]) and the LARGEST scale of a hidden activation.
// A hidden layer's own scale is chosen at bind time from its LayerNorm affine parameters (k_ascale: the largest power
// of two <= 2^5 that keeps |Mish(LayerNorm(.))| * scale below the f16 maximum for ANY input), so that a checkpoint with a
// huge LayerNorm gain cannot overflow the hi piece into Inf -> NaN -> nan_to_num(0).  The consuming layer's output
// scale (LayerS::oscale) carries the matching 2^-(kw + log2 scale_in).
constexpr float ACT_SCALE = 32.f;  // 2^5
constexpr int ACT_SCALE_LOG2 = 5;

struct LayerS {
    const _Float16 *wp;  // split: packed [CT][KB16][2 planes][64 lanes][8] f16; exact fp32: [CT][KB8][64 lanes][4] fp32
    const float *bias;   // [CT*32] zero padded
    const float *g, *b;  // LayerNorm affine (null for plain output layers)
    const float *oscale; // device scalar: 2^-(kw + log2 of the input's scale) (split) or 1 (exact fp32)
    const float *ascale; // device scalar: scale of THIS layer's output in operand form (split: 2^ka <= 32; exact fp32: 1)
    int KB;              // k-blocks: of 16 (split) or of 8 (exact fp32)
    int CT;
};
struct NetS {
    LayerS l[3];
};

// ---------------------------------------------------------------- kernel parameter blocks (NET = NetS, fused_kernels.cuh)
// elite select + refit (tdmpc2/tdmpc2.py:184-206): refit_plan() below
// A bounded inter-workgroup wait gave up.  The handle's host-mapped line: word 0 = "the call in flight is invalid" -- read by the
// call's own last kernel (refit_plan's final pick, l_value_head), cleared IN STREAM ORDER at the start of the next call, never by
// the host; word 8 = sticky "a wait gave up since the host last looked" -- read and cleared by the host (validate_envs,
// take_fault).  (Round 3 had one word that the host cleared when it enqueued the next call: pipelined calls could lose the verdict
// of the call still running -- ADVICE r3.)
// The bound of those waits is wall-clock time (round 5; it was a poll count worth about a third of a second): 5 ms of the
// constant 100 MHz clock (s_memrealtime).  A healthy wait is microseconds, or -- a row block that straddles its XCD's residency
// limit, a tile whose last K-part is still queued -- one tile's run time (< 0.5 ms for the 317M model's K = 4096).  The clock is
// only read from the 256th poll on (and then every 64th), so that a hand-over on the latency path never pays for it.
constexpr unsigned long long WAIT_TICKS = 500000ull;
// The bound is the handle's to choose (TDMPC2_TUNE_WAIT_US, ABI 9): word ERR_WAIT_TICKS of its host-mapped error line holds it in
// clock ticks (0: the default above).  Read ONCE per slow wait, when the clock is first consulted.
constexpr int ERR_WAIT_TICKS = 12;
struct WaitClock {
    int spin = 0;
    unsigned long long t0 = 0, limit = WAIT_TICKS;
    __device__ __forceinline__ bool expired(const unsigned int *err = nullptr) {
        if (++spin < 256 || (spin & 63) != 0) return false;
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (t0 == 0) {
            t0 = now;
            if (err) {
                const unsigned int w = __hip_atomic_load(err + ERR_WAIT_TICKS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (w) limit = w;
            }
            return false;
        }
        return now - t0 > limit;
    }
};
__device__ __forceinline__ void raise_fault(unsigned int *err, unsigned int code) {
    __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(err + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct RefitParams {
    int E, N, H, A, K, iter, last, eval_mode;
    int Nvalid;            // rows >= Nvalid (> 0) are padding of num_samples to the row tile: never elites (tdmpc2_plan_cfg::num_valid_samples)
    int stage;             // elite actions are staged in LDS ([K][H*A] floats after the other arrays): see refit_lds_bytes
    float temperature, min_std, max_std;
    float *value;          // [E,N] in/out (nan_to_num)
    const float *actions;  // [E,H,N,A]
    const float *act_mask; // [E,A] or null
    float *mean, *std;     // [E,H,A] out
    float *score;          // [E,K] out (may be null)
    int *elite_idx;        // [E,K] out (may be null)
    // last iteration only
    const float *gumbel_exp;  // [E,K] or null -> Philox
    const float *final_eps;   // [E,A] or null -> Philox
    unsigned long long seed;
    unsigned int call;
    float *prev_mean;  // [E,H,A]
    float *action;     // [E,A]
    // cluster path: the handle's host-mapped error word (a bounded hand-over wait gave up somewhere in this plan).  When it is
    // set the plan's numbers are garbage: the final pick then returns NaN actions and leaves prev_mean untouched, so that the
    // caller can re-plan the same step (tdmpc2_plan_take_fault); null on every other path.
    const unsigned int *err;
    // sharded plans (one API call per CEM iteration): the sticky word the host raises when it consumes `err` between two
    // iterations of the plan in flight -- checked like `err` by the final pick; null on every other path.
    const unsigned int *err2;
    // In-launch refit (fused family, ks_rollout's last-arriver epilogue): the elite actions are RE-DERIVED from the
    // iteration's sampling distribution and noise instead of being read back from `actions` -- the workgroups that sampled
    // them sit on other XCDs, and shipping 64 x H x A floats per workgroup through write-through stores cost 13 % of the
    // launch (profiles/README.md r02c).  regen = 1: rows n >= P: clamp(old_mean + old_std * eps) * mask with eps from the tape
    // slice or Philox (the rollout kernel's own formula and indices); rows n < P: the policy-prior actions written by an
    // earlier launch.
    int regen, P, Apad;
    const float *sample_eps;   // tape slice of this iteration (or null: Philox)
    long sample_eps_estride;
    // debug copies (per iteration slices already offset by the host; env stride given)
    float *dbg_value; long dbg_value_es;
    int *dbg_idx; long dbg_idx_es;
    float *dbg_score; long dbg_score_es;
    float *dbg_mean; long dbg_mean_es;
    float *dbg_std; long dbg_std_es;
};

template <class NET>
struct RolloutParamsT {
    int E, N, H, A, Apad, P, stride, tiles, nq, num_bins, multitask, given_actions, iter, iters_total;
    int tile_off;  // first row tile of the range this launch covers (tiles = tiles in the range)
    int nnets;  // vectors per plan in `beff`
    float log_std_min, log_std_dif;
    NET dyn, rew, pi, term;
    NET q[MAXQ];
    const float *bins;
    const float *z0;        // [E,L]
    const float *beff;      // [E,nnets,WIDTH] effective first-layer biases (multitask) or null
    const float *cvec;      // [E,2,WIDTH]: z0-part (+bias) of reward / dynamics layer 1
    const float *act_mask;  // [E,A] or null
    const float *disc_pow;  // [E,H+1]
    const float *mean;      // [E,H,A]
    const float *std;       // [E,H,A]
    const float *sample_eps;  // tape slice for this iteration: env stride given below
    long sample_eps_estride;
    const float *pi_eps;
    long pi_eps_estride;
    const int *qidx;
    long qidx_estride;
    unsigned long long seed;
    unsigned int call;
    float *actions;   // [E,H,N,A]
    float *value;     // [E,N]
    float *zscratch;  // [E*tiles,64,WIDTH]
    float *trace_tiles;    // optional [E*tiles, 5H+7, 64, WIDTH] activations after each phase
    float *trace_scalars;  // optional [E, N, H+2+A]: r_0..r_{H-1}, Q_a, Q_b, a_H[A]
    unsigned long long *timing;  // profiling builds (-DSPLIT_TIMING): 16 cycle counters summed over workgroups, else null
    // elite selection + refit by the last workgroup of each plan to finish (one launch per CEM iteration)
    int fold_refit;
    unsigned int *ticket;   // [E] arrival counters, zero between launches
    RefitParams rf;
    // cluster path (cluster_kernels.cuh): exchange tiles, arrival words, the handle's error word, per-member z_H scratch
    float *cl_xbuf;
    unsigned int *cl_flags;
    unsigned int *cl_err;
    float *cl_zs;
    int cl_fault;              // test hook: member 7 of cluster 0 never signals (the bounded waits must report it)
    int pi_fold;               // the policy-prior trajectories (tdmpc2.py:154-160) are computed by cluster 0 of each plan in launch 0
    const float *pi_traj_eps;  // [E,H,P,A] or null (Philox)
    // two clusters per tile (cluster2_kernels.cuh, ks_rollout_cl2): their exchange tiles, arrival words, z_H scratch, (G, Qb) mailbox
    float *cl2_xbuf;
    unsigned int *cl2_flags;
    float *cl2_zs;
    float *cl2_mail;
};

// pi + two Q heads on a batch of latent rows (fused_kernels.cuh: ks_value)
template <class NET>
struct ValueParamsT {
    int rows, A, Apad, nq, num_bins, reduce_min;
    float log_std_min, log_std_dif, discount;
    NET pi;
    NET q[MAXQ];
    const float *bins;
    const float *z;        // [rows, L]
    const float *pi_eps;   // [rows, A] or null (Philox)
    const int *qidx;       // [2] or null (Philox)
    unsigned long long seed;
    unsigned int call;
    const float *reward, *terminated;  // [rows] or null
    float *action;         // [rows, A] or null
    float *out;            // [rows]
    // multitask batches: one task per row
    int nnets;
    const int *task_ids;     // [rows] or null (single task)
    const float *beff_tab;   // [n_tasks, nnets, WIDTH] effective first-layer biases (ks_task_bias)
    const float *mask_tab;   // [n_tasks, A]
    const float *disc_tab;   // [n_tasks] or null (scalar `discount`)
};

// net slots inside `beff`
enum { BE_DYN = 0, BE_REW = 1, BE_PI = 2, BE_Q0 = 3 };

// ---------------------------------------------------------------- small math
__device__ __forceinline__ float mish_f(float x) {
    // x * tanh(softplus(x)) == x * n / (n + 2),  n = e^x (e^x + 2); no cancellation for x << 0.
    // reference: nn.Mish in NormedLinear, tdmpc2/common/layers.py:103
    if (x > 20.f) return x;
    const float e = expf(x);
    const float n = e * (e + 2.f);
    return x * (n / (n + 2.f));
}

__device__ __forceinline__ float symexp_f(float x) {
    // tdmpc2/common/math.py:50-55: sign(x) * (exp(|x|) - 1)
    const float m = expf(fabsf(x)) - 1.f;
    return x > 0.f ? m : (x < 0.f ? -m : 0.f);
}

// One sampled action (tdmpc2/tdmpc2.py:176-178): (mean + std * r).clamp(-1, 1) with the reference's two roundings (torch
// does not fuse the multiply-add); used by every kernel that draws or re-derives a sample, so that they agree bit for bit.
__device__ __forceinline__ float sample_action(float mean, float std, float r) {
    return fminf(fmaxf(__fadd_rn(mean, __fmul_rn(std, r)), -1.f), 1.f);
}

template <int W>
__device__ __forceinline__ float group_sum(float v) {  // sum over aligned groups of W lanes
#pragma unroll
    for (int m = 1; m < W; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
template <int W>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int m = 1; m < W; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// ---------------------------------------------------------------- Philox4x32-10 (fast mode RNG)
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
enum { SITE_PITRAJ = 1, SITE_SAMPLE = 2, SITE_PI = 3, SITE_QIDX = 4, SITE_GUMBEL = 5, SITE_FINAL = 6 };

__device__ __forceinline__ uint4 rng_raw(unsigned long long seed, unsigned call, int site, int iter, int env,
                                         unsigned idx) {
    return philox4x32_10(make_uint4(idx, (unsigned)(site | (iter << 8)), (unsigned)env, call),
                         make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
}
__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float rng_normal(unsigned long long seed, unsigned call, int site, int iter, int env,
                                            unsigned idx) {
    const uint4 r = rng_raw(seed, call, site, iter, env, idx);
    const float u1 = u01(r.x), u2 = u01(r.y);
    return sqrtf(-2.f * logf(u1)) * cospif(2.f * u2);
}
__device__ __forceinline__ float rng_exponential(unsigned long long seed, unsigned call, int site, int iter,
                                                 int env, unsigned idx) {
    return -logf(u01(rng_raw(seed, call, site, iter, env, idx).x));
}

// Two standard normals from ONE Philox4x32-10 call (Box-Muller, both branches) with the hardware log / sin / cos:
// the sampled distribution only has to be N(0,1) to sampling accuracy (fast mode; parity runs replay a noise tape).
__device__ __forceinline__ void rng_normal2(unsigned long long seed, unsigned call, int site, int iter, int env, unsigned pair,
                                            float &n0, float &n1) {
    const uint4 r = rng_raw(seed, call, site, iter, env, pair);
    const float u1 = u01(r.x), u2 = u01(r.y);
    const float rad = __builtin_amdgcn_sqrtf(-2.f * __logf(u1));
    n0 = rad * __builtin_amdgcn_cosf(u2);  // v_cos_f32 / v_sin_f32 take the angle in revolutions
    n1 = rad * __builtin_amdgcn_sinf(u2);
}

// per-plan setup: (1) effective first-layer biases b + W[:, L:L+T] . task_emb (multitask); (2) cvec = z0-part (+ bias) of
// the reward / dynamics first layers (all rows share z0 at t = 0, tdmpc2/tdmpc2.py:163); (3) mean / std initialisation
// and warm start (tdmpc2.py:164-167)
template <class NET>
struct SetupParamsT {
    int E, H, A, T, multitask, nq, nnets, stride;
    float max_std;
    NET dyn, rew, pi;
    NET q[MAXQ];
    const float *wemb[3 + MAXQ];  // [out=WIDTH][T] task-embedding columns of each first layer
    const float *z0, *task_emb, *prev_mean;
    const unsigned char *t0;
    float *beff, *cvec, *mean, *std;
    unsigned int *cl_flags;  // cluster path: arrival words, zeroed at the start of every plan ([E][cl_flag_words]) or null
    int cl_flag_words;
    unsigned int *cl2_flags; // ks_rollout_cl2's arrival words (single plans), zeroed by plan 0's workgroup, or null
    int cl2_flag_words;
    int skip_cvec;           // cluster path: no z0 products (cvec unused)
    unsigned int *err_clear; // word 0 of the handle's error line: zeroed here, i.e. in stream order at the start of the plan (raise_fault)
};

// policy-prior trajectories (tdmpc2/tdmpc2.py:154-160): rows < P of one tile per plan
template <class NET>
struct PiTrajParamsT {
    int E, N, H, A, Apad, P, stride, multitask, nnets;
    float log_std_min, log_std_dif;
    NET dyn, pi;
    const float *z0, *beff, *act_mask;
    const float *pi_traj_eps;  // [E,H,P,A] or null
    unsigned long long seed;
    unsigned int call;
    float *actions;   // [E,H,N,A]
    float *zscratch;  // [E,64,WIDTH] (tile 0 of each plan)
    long zscratch_estride;
};

// ================================================================ kernel: elite select + refit (struct RefitParams: above)
// dynamic LDS of the refit; `stage` out: whether the K x H x A elite actions fit next to the rest (they are then gathered
// by the whole workgroup in one round of loads instead of 2 K dependent global loads per (t, a) thread: 35 -> 12 us)
inline size_t refit_lds_bytes(int N, int K, int H, int A, int *stage, size_t budget = 48 * 1024) {
    size_t M = 64;
    while (M < (size_t)N) M <<= 1;  // sort keys: 8 bytes per padded sample
    const size_t base = (2 * M + 3 * (size_t)K + 4 * H * A + 48) * 4 + 64;
    const size_t elite = (size_t)K * H * A * 4;
    *stage = base + elite <= budget;
    return *stage ? base + elite : base;
}


// block-wide sum / max over `n` floats in LDS: strided thread-local partials, wavefront shuffle reduction, one LDS slot per
// wave, every thread reads the slots back (fixed order -> deterministic, the same value in every thread)
__device__ __forceinline__ float block_sum_lds(const float *x, int n, float *slots, int tid, int nthr) {
    float s = 0.f;
    for (int i = tid; i < n; i += nthr) s += x[i];
    s = group_sum<64>(s);
    __syncthreads();  // slots may still be read from a previous reduction
    if ((tid & 63) == 0) slots[tid >> 6] = s;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (nthr >> 6); ++w) t += slots[w];
    return t;
}
__device__ __forceinline__ float block_max_lds(const float *x, int n, float *slots, int tid, int nthr) {
    float s = -INFINITY;
    for (int i = tid; i < n; i += nthr) s = fmaxf(s, x[i]);
    s = group_max<64>(s);
    __syncthreads();
    if ((tid & 63) == 0) slots[tid >> 6] = s;
    __syncthreads();
    float t = -INFINITY;
    for (int w = 0; w < (nthr >> 6); ++w) t = fmaxf(t, slots[w]);
    return t;
}

// Elite selection + refit (+ final pick) of plan `e` by one workgroup of `nthr` threads (a multiple of 64; any N <= 1024).
// tdmpc2/tdmpc2.py:184-206.  Called by k_refit (one workgroup per plan) and by the last workgroup of a plan to finish
// its rollouts (ks_rollout, fused family): the elite statistics are wavefront-shuffle reductions.
// monotone map float -> uint (larger float <-> larger uint) and back; -0.0 has been folded into +0.0 by the caller
__device__ __forceinline__ unsigned ordered_of(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_of_ordered(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
}
#ifdef REFIT_TIMING  // probe builds: thread 0 leaves the cycle count of every phase in score[e][phase] (tools/probes)
#define RT_MARK(i) { if (tid == 0 && p.score) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); p.score[(size_t)e * p.K + (i)] = (float)(t_ - rt_last); rt_last = t_; } }
#define RT_INIT unsigned long long rt_last = __builtin_amdgcn_s_memtime();
#else
#define RT_MARK(i)
#define RT_INIT
#endif

// forceinline: as a real call it takes the ADDRESS of the caller's kernel-argument member (`p.rf`), which makes the compiler
// copy the caller's whole 2.5 KB argument struct to scratch and read every parameter from there (seen when the inliner's
// budget ran out in ks_rollout_cl: 1.47 -> 2.1 ms per plan).
__device__ __forceinline__ void refit_plan(const RefitParams &p, int e, float *smem, int tid, int nthr) {
    int M = 64;  // sort width: the power of two >= N
    while (M < p.N) M <<= 1;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);  // [M] sort keys (or: [N] floats, counting path)
    float *sv = smem;
    float *ev = smem + 2 * M;                   // [K]
    float *sc = ev + p.K;                       // [K]
    int *ei = reinterpret_cast<int *>(sc + p.K);  // [K]
    float *smean = reinterpret_cast<float *>(ei + p.K);  // [H*A]
    float *sstd = smean + p.H * p.A;                     // [H*A]
    float *slots = sstd + p.H * p.A;                     // [16] per-wave partials + [16] scratch scalars
    int *s_pick = reinterpret_cast<int *>(slots + 32);
    float *omean = slots + 48;                           // [H*A] the distribution this iteration sampled from (regen)
    float *ostd = omean + p.H * p.A;                     // [H*A]
    float *ea = ostd + p.H * p.A;                        // [K][H*A] elite_actions (tdmpc2.py:186) when staged
    RT_INIT
    const bool sorted_path = M <= nthr;  // one key per thread
    // value.nan_to_num(0): nan -> 0, +-inf -> +-FLT_MAX (tdmpc2.py:184)
    unsigned long long key = 0ull;  // padding keys sort last
    const int NV = p.Nvalid > 0 ? p.Nvalid : p.N;
    for (int i = tid; i < p.N; i += nthr) {
        if (i >= NV) {  // padding of num_samples to the row tile: behind every real row, whatever it evaluated to
            if (!sorted_path) sv[i] = -INFINITY;  // (real rows are finite after nan_to_num; ties go by index: lower first)
            continue;                             // sorted path: key stays 0, below every real key
        }
        // in-launch refit: the values of the plan's other workgroups arrive as write-through stores from other XCDs; read
        // them with agent-scope (sc1) loads -- past the L1, from lines this XCD's L2 cannot hold yet -- and do NOT
        // invalidate caches (an agent-scope acquire here, buffer_inv sc1, drops the XCD's L2-resident weights: +13 %)
        float v = p.regen ? __hip_atomic_load(p.value + (size_t)e * p.N + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                          : p.value[(size_t)e * p.N + i];
        if (v != v) v = 0.f;
        else if (v == INFINITY) v = 3.402823466e+38f;
        else if (v == -INFINITY) v = -3.402823466e+38f;
        p.value[(size_t)e * p.N + i] = v;
        if (p.dbg_value) p.dbg_value[(size_t)e * p.dbg_value_es + i] = v;
        if (sorted_path) key = ((unsigned long long)ordered_of(v + 0.f) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        else sv[i] = v;
    }
    RT_MARK(0)
    if (sorted_path) {
        // torch.topk(..., sorted=True) order (value desc, index asc on ties; tdmpc2.py:185) = descending order of the 64-bit
        // keys (ordered value | ~index): a bitonic sort with one key per thread -- the 39 of 45 stages (N = 512) whose
        // partner sits in the same wavefront are register shuffles, the rest go through LDS.
        for (int k = 2; k <= M; k <<= 1) {
            const bool desc = (tid & k) == 0;  // this k-block ends up descending (the last level: everyone)
            for (int j = k >> 1; j > 0; j >>= 1) {
                unsigned long long other;
                if (j >= 64) {
                    if (tid < M) keys[tid] = key;
                    __syncthreads();
                    other = tid < M ? keys[tid ^ j] : 0ull;
                    __syncthreads();
                } else {
                    const unsigned lo = __shfl_xor((unsigned)key, j), hi = __shfl_xor((unsigned)(key >> 32), j);
                    other = ((unsigned long long)hi << 32) | lo;
                }
                const bool lower = (tid & j) == 0;
                const bool keep_max = lower == desc;
                key = keep_max ? (key > other ? key : other) : (key < other ? key : other);
            }
        }
        if (tid < p.K) {
            ei[tid] = (int)(0xFFFFFFFFu - (unsigned)key);
            ev[tid] = float_of_ordered((unsigned)(key >> 32));
        }
        __syncthreads();
    } else {
        __syncthreads();
        // rank = position in (value desc, index asc) order by counting: every thread compares its value with the whole LDS
        // array (broadcast reads, no bank conflicts).  Only when N exceeds the workgroup (N = 1024 inside the 512-thread
        // rollout kernel).
        for (int i = tid; i < p.N; i += nthr) {
            const float v = sv[i];
            int rank = 0;
            for (int j = 0; j < p.N; j += 4) {
                const f32x4 u = *reinterpret_cast<const f32x4 *>(sv + j);  // N is a multiple of 64
#pragma unroll
                for (int q = 0; q < 4; ++q) rank += (u[q] > v) || (u[q] == v && j + q < i);
            }
            if (rank < p.K) {
                ei[rank] = i;
                ev[rank] = v;
            }
        }
        __syncthreads();
    }
    RT_MARK(1)
    const float vmax = ev[0];  // max(elite_value)
    for (int k = tid; k < p.K; k += nthr) sc[k] = expf(p.temperature * (ev[k] - vmax));
    const float s1 = block_sum_lds(sc, p.K, slots, tid, nthr);
    for (int k = tid; k < p.K; k += nthr) sc[k] = sc[k] / s1;  // score / score.sum(0)  (tdmpc2.py:191)
    const float s_ssum = block_sum_lds(sc, p.K, slots, tid, nthr) + 1e-9f;  // score.sum(0) + 1e-9 (tdmpc2.py:192-193)
    RT_MARK(2)
    const float *acts = p.actions + (size_t)e * p.H * p.N * p.A;
    const int HA = p.H * p.A;
    if (p.regen) {  // (always staged) elite actions re-derived from (old mean, old std, noise): see RefitParams
        for (int idx = tid; idx < HA; idx += nthr) {
            omean[idx] = p.mean[(size_t)e * HA + idx];
            ostd[idx] = p.std[(size_t)e * HA + idx];
        }
        __syncthreads();
        // one work item per PAIR of action columns (a, a + 1): one Philox call yields both normals, as in the rollout
        const int hp = p.Apad / 2, hpa = (p.A + 1) / 2, per_k = p.H * hpa;
        for (int idx = tid; idx < p.K * per_k; idx += nthr) {
            const int k = idx / per_k, rem = idx - k * per_k;
            const int t = rem / hpa, a0 = 2 * (rem - t * hpa);
            const int n = ei[k];
            float v[2] = {0.f, 0.f};
            if (n < p.P) {  // policy-prior rows: written by ks_pitraj (an earlier launch) or, on the cluster path, by another
                            // workgroup of THIS launch (agent-scope stores there, agent-scope loads here)
                v[0] = __hip_atomic_load(acts + ((size_t)t * p.N + n) * p.A + a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a0 + 1 < p.A)
                    v[1] = __hip_atomic_load(acts + ((size_t)t * p.N + n) * p.A + a0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                float r[2] = {0.f, 0.f};
                if (p.sample_eps) {
                    const float *ep = p.sample_eps + (size_t)e * p.sample_eps_estride + (unsigned)(((size_t)t * (p.N - p.P) + (n - p.P)) * p.A + a0);
                    r[0] = ep[0];
                    if (a0 + 1 < p.A) r[1] = ep[1];
                } else {
                    const unsigned pair = (unsigned)(((size_t)t * (p.N - p.P) + (n - p.P)) * hp + a0 / 2);
                    rng_normal2(p.seed, p.call, SITE_SAMPLE, p.iter, e, pair, r[0], r[1]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (a0 + u < p.A) {
                        v[u] = sample_action(omean[t * p.A + a0 + u], ostd[t * p.A + a0 + u], r[u]);
                        if (p.act_mask) v[u] *= p.act_mask[(size_t)e * p.A + a0 + u];
                    }
                }
            }
            ea[k * HA + t * p.A + a0] = v[0];
            if (a0 + 1 < p.A) ea[k * HA + t * p.A + a0 + 1] = v[1];
        }
        __syncthreads();
    } else if (p.stage) {
        // gather K x H x A elite actions: a wave takes every (nthr / 64)-th elite, its lanes the (t, a) columns -- one
        // integer division per column instead of two per element, and the loads of a wave's elites are independent
        const int wv = tid >> 6, ln = tid & 63, nwv = nthr >> 6;
        for (int ha = ln; ha < HA; ha += 64) {
            const int t = ha / p.A, a = ha - t * p.A;
            const float *col = acts + (size_t)t * p.N * p.A + a;
#pragma unroll 8
            for (int k = wv; k < p.K; k += nwv) ea[k * HA + ha] = col[(size_t)ei[k] * p.A];
        }
        __syncthreads();
    }
    RT_MARK(3)
    if (p.stage) {
        // four lanes per (t, a) output, each over a quarter of the elites, combined with two quad shuffles
        const int sub = tid & 3;
        for (int idx = tid >> 2; idx < HA; idx += nthr >> 2) {  // the four lanes of a quad share idx: uniform trip count
            float m = 0.f;
#pragma unroll 4
            for (int k = sub; k < p.K; k += 4) m += sc[k] * ea[k * HA + idx];
            m += __shfl_xor(m, 1);
            m += __shfl_xor(m, 2);
            m = m / s_ssum;
            float s2 = 0.f;
#pragma unroll 4
            for (int k = sub; k < p.K; k += 4) {
                const float d = ea[k * HA + idx] - m;
                s2 += sc[k] * (d * d);
            }
            s2 += __shfl_xor(s2, 1);
            s2 += __shfl_xor(s2, 2);
            if (sub != 0) continue;
            float sd = sqrtf(s2 / s_ssum);
            sd = fminf(fmaxf(sd, p.min_std), p.max_std);
            if (p.act_mask) {
                const float mk = p.act_mask[(size_t)e * p.A + idx % p.A];
                m *= mk;
                sd *= mk;
            }
            smean[idx] = m;
            sstd[idx] = sd;
            p.mean[(size_t)e * p.H * p.A + idx] = m;
            p.std[(size_t)e * p.H * p.A + idx] = sd;
            if (p.dbg_mean) p.dbg_mean[(size_t)e * p.dbg_mean_es + idx] = m;
            if (p.dbg_std) p.dbg_std[(size_t)e * p.dbg_std_es + idx] = sd;
        }
    } else
    for (int idx = tid; idx < HA; idx += nthr) {
        const int t = idx / p.A, a = idx % p.A;
        const float *at = acts + (size_t)t * p.N * p.A + a;
        float m = 0.f;
        for (int k = 0; k < p.K; ++k) m += sc[k] * at[(size_t)ei[k] * p.A];
        m = m / s_ssum;
        float s2 = 0.f;
        for (int k = 0; k < p.K; ++k) {
            const float d = at[(size_t)ei[k] * p.A] - m;
            s2 += sc[k] * (d * d);
        }
        float sd = sqrtf(s2 / s_ssum);
        sd = fminf(fmaxf(sd, p.min_std), p.max_std);
        if (p.act_mask) {
            const float mk = p.act_mask[(size_t)e * p.A + a];
            m *= mk;
            sd *= mk;
        }
        smean[idx] = m;
        sstd[idx] = sd;
        p.mean[(size_t)e * p.H * p.A + idx] = m;
        p.std[(size_t)e * p.H * p.A + idx] = sd;
        if (p.dbg_mean) p.dbg_mean[(size_t)e * p.dbg_mean_es + idx] = m;
        if (p.dbg_std) p.dbg_std[(size_t)e * p.dbg_std_es + idx] = sd;
    }
    RT_MARK(4)
#ifndef REFIT_TIMING
    for (int k = tid; k < p.K; k += nthr) {
        if (p.score) p.score[(size_t)e * p.K + k] = sc[k];
        if (p.elite_idx) p.elite_idx[(size_t)e * p.K + k] = ei[k];
        if (p.dbg_score) p.dbg_score[(size_t)e * p.dbg_score_es + k] = sc[k];
        if (p.dbg_idx) p.dbg_idx[(size_t)e * p.dbg_idx_es + k] = ei[k];
    }
#endif
    RT_MARK(5)
    if (!p.last) return;
    __syncthreads();
    // gumbel_softmax_sample(score) (tdmpc2/common/math.py:86-94): argmax softmax(log p - log Exp(1)); first index on ties
    for (int k = tid; k < p.K; k += nthr) {
        const float ex = p.gumbel_exp ? p.gumbel_exp[(size_t)e * p.K + k]
                                      : rng_exponential(p.seed, p.call, SITE_GUMBEL, 0, e, (unsigned)k);
        ev[k] = logf(sc[k]) + (-logf(ex));
    }
    const float gmax = block_max_lds(ev, p.K, slots, tid, nthr);
    for (int k = tid; k < p.K; k += nthr) ev[k] = expf(ev[k] - gmax);
    const float gs = block_sum_lds(ev, p.K, slots, tid, nthr);
    for (int k = tid; k < p.K; k += nthr) ev[k] = ev[k] / gs;
    const float ymax = block_max_lds(ev, p.K, slots, tid, nthr);
    if (tid == 0) *s_pick = p.K;
    __syncthreads();
    for (int k = tid; k < p.K; k += nthr)
        if (ev[k] == ymax) atomicMin(s_pick, k);
    __syncthreads();
    const int pick = ei[*s_pick];
    bool bad = false;
    if (p.err || p.err2) {  // uniform: one system-scope load of the host-mapped word(s), broadcast through LDS
        if (tid == 0) {
            unsigned int w = 0;
            if (p.err) w |= __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (p.err2) w |= __hip_atomic_load(p.err2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            s_pick[1] = w != 0;
        }
        __syncthreads();
        bad = s_pick[1] != 0;
    }
    if (bad) {
        for (int a = tid; a < p.A; a += nthr) p.action[(size_t)e * p.A + a] = __uint_as_float(0x7fc00000u);
        return;
    }
    for (int a = tid; a < p.A; a += nthr) {
        float x = p.stage ? ea[(size_t)*s_pick * HA + a] : acts[(size_t)pick * p.A + a];  // elite_actions[0, rand_idx]
        if (!p.eval_mode) {
            const float n = p.final_eps ? p.final_eps[(size_t)e * p.A + a]
                                        : rng_normal(p.seed, p.call, SITE_FINAL, 0, e, (unsigned)a);
            x = x + sstd[a] * n;  // a + std[0] * randn (tdmpc2.py:203-204)
        }
        p.action[(size_t)e * p.A + a] = fminf(fmaxf(x, -1.f), 1.f);
    }
    for (int idx = tid; idx < p.H * p.A; idx += nthr)
        p.prev_mean[(size_t)e * p.H * p.A + idx] = smean[idx];  // _prev_mean.copy_(mean) (tdmpc2.py:205)
    RT_MARK(6)
}

// threads of a k_refit workgroup: the sort width (one key per thread)
inline int refit_threads(int N) {
    int M = 64;
    while (M < N) M <<= 1;
    return M;
}


// beff_tab[task][net][WIDTH] = b + W[:, L:L+T] . task_emb[task] for the policy and the Q heads (online or target):
// the per-task effective first-layer biases ks_value indexes per row.  grid = n_tasks, block = WIDTH threads.
struct TaskBiasParams {
    int T, nq, nnets;
    const float *task_emb;            // [n_tasks, T] (max_norm renorm applied by the caller, world_model.py:21)
    const float *wemb[3 + MAXQ];      // [WIDTH][T] per net slot (null: skipped)
    const float *bias[3 + MAXQ];      // [WIDTH]
    float *beff_tab;
};
// Per-layer scalars of the split arithmetic, device resident (one record per layer and ensemble member).
struct LayerScal {
    float wscale;          // 2^kw: applied to the weights when they are packed
    float oscale;          // 2^-(kw + log2 of the INPUT's operand scale): applied to the fp32 accumulator
    unsigned int maxbits;  // max |W| as bits (k_absmax)
    int kw;
    float ascale;          // 2^ka: operand scale of this layer's OUTPUT (hidden layers; <= ACT_SCALE)
    int ka;
    unsigned int gmax, bmax;  // max |LayerNorm weight|, max |LayerNorm bias| as bits
};

}  // namespace tdk
using namespace tdk;
