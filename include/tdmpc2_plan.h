/*
 * tdmpc2_plan.h — C ABI of the MI355X-native TD-MPC2 MPPI/CEM planner.
 *
 * The reference (nicklashansen/tdmpc2) is pure Python and has no FFI layer; its
 * boundary for this path is the method pair
 *     TDMPC2.act(obs, t0, eval_mode, task)        tdmpc2/tdmpc2.py:97-120
 *     TDMPC2.plan -> _plan(obs, t0, eval_mode, task)  tdmpc2/tdmpc2.py:45-55,138-206
 * The entry points below are what a ctypes binding of that path needs (the
 * reference-side stub is shown in INTEGRATION.md).  They replace, one for one:
 *
 *   tdmpc2_plan_create        <- TDMPC2.__init__ planner state         tdmpc2/tdmpc2.py:17-43
 *   tdmpc2_plan_bind_weights  <- WorldModel parameters / load_state_dict tdmpc2/common/world_model.py:20-36,
 *                                                                       tdmpc2/tdmpc2.py:81-95
 *   tdmpc2_plan_run           <- TDMPC2._plan after encode()            tdmpc2/tdmpc2.py:154-206
 *   tdmpc2_plan_estimate_value<- TDMPC2._estimate_value                 tdmpc2/tdmpc2.py:122-136
 *   tdmpc2_plan_refit         <- the elite select + refit block         tdmpc2/tdmpc2.py:184-197
 *   tdmpc2_plan_bind_encoder  <- the state encoder's parameters         tdmpc2/common/layers.py:153-164
 *   tdmpc2_plan_encode        <- WorldModel.encode (state observations) tdmpc2/common/world_model.py:103-112
 *   tdmpc2_plan_run_obs       <- TDMPC2._plan including encode()        tdmpc2/tdmpc2.py:152-206
 *   tdmpc2_plan_td_target[_mt]    <- TDMPC2._td_target                  tdmpc2/tdmpc2.py:239-254
 *   tdmpc2_plan_policy_value[_mt] <- forward half of TDMPC2.update_pi   tdmpc2/tdmpc2.py:208-225
 *   tdmpc2_plan_export_packed / import_packed <- TDMPC2.save / load of the planner's weights  tdmpc2/tdmpc2.py:72-95
 *   tdmpc2_plan_export_noise  <- the six RNG draw sites of one plan    tdmpc2/tdmpc2.py:176,204, tdmpc2/common/world_model.py:156,212,
 *                                (what torch.manual_seed pins there)   tdmpc2/common/math.py:90
 *
 * Conventions
 *   - plain C types only; every tensor is a DEVICE pointer to fp32 (or int32 /
 *     uint8 where stated), densely packed, last index fastest.
 *   - the caller owns every buffer it passes; the library owns only what it
 *     allocates in create/bind and frees in destroy.  `run` allocates nothing.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *     a handle is not re-entrant: a call entered while another thread is inside
 *     the same handle returns TDMPC2_ERR_STATE (use one handle per thread; handles
 *     share nothing).  A handle is one workspace, so its calls must not overlap on
 *     the device either: calls on ONE stream are ordered by the stream; when a handle
 *     moves to another stream its first call there waits (hipStreamWaitEvent) for the
 *     handle's last call on the previous stream -- the library does that itself.
 *     The exception is stream capture: a captured call neither waits nor leaves an
 *     event (the graph replays wherever it is launched); ordering a graph launch
 *     against eager calls of the same handle on other streams is the caller's.
 *     Every call runs on the handle's device (cfg.device) and
 *     restores the caller's current device.  Return value 0 = ok, otherwise an error code
 *     (no C++ exception crosses the ABI); `tdmpc2_last_error()` has the text.
 *   - E = number of independent environments planned in one call (the reference
 *     is E = 1: tdmpc2/tdmpc2.py:111,163).  H horizon, N num_samples,
 *     K num_elites, P num_pi_trajs, I iterations, A action_dim, L latent_dim,
 *     M mlp_dim, T task_dim, B num_bins, nq num_q.
 */
#ifndef TDMPC2_PLAN_H
#define TDMPC2_PLAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDMPC2_PLAN_ABI_VERSION 9

typedef struct tdmpc2_plan tdmpc2_plan_t;

/* Planner constants; field names follow the reference config
 * (tdmpc2/config.yaml:33-64).  `iterations` is the value AFTER the reference's
 * "+2 if action_dim >= 20" adjustment (tdmpc2/tdmpc2.py:34) — the host applies it. */
typedef struct tdmpc2_plan_cfg {
    int32_t horizon, num_samples, num_elites, num_pi_trajs, iterations;
    int32_t action_dim, latent_dim, mlp_dim, task_dim, num_bins, num_q, simnorm_dim;
    /* num_bins: 2 .. 128 two-hot bins, or 0 / 1 = the reference's regression heads (one output column; two_hot_inv is the
     * identity / symexp, common/math.py:76-79).  simnorm_dim: 8 (layers.py:84-88; every released model). */
    float vmin, vmax, min_std, max_std, temperature;
    float log_std_min, log_std_dif;   /* WorldModel buffers, world_model.py:34-35 */
    int32_t multitask, episodic;
    int32_t max_envs;                 /* workspace is sized for this many concurrent plans */
    int32_t device;                   /* HIP device ordinal */
    int32_t path;                     /* enum tdmpc2_path: which kernel family runs the rollout */
    int32_t precision;                /* enum tdmpc2_precision: how the fp32 contractions are carried out */
    int32_t num_valid_samples;        /* ABI 8.  0 = num_samples.  Otherwise (num_elites <= . <= num_samples): the reference's
                                       * cfg.num_samples when that is not a multiple of the kernels' row tile (config.yaml:36 allows
                                       * any value; the kernels want 64 / 128): create the handle with num_samples rounded UP, pass the
                                       * true count here, and pad the noise tapes' sample axis to the rounded count.  The padding rows
                                       * are rolled out like the others but can never be elites: they sort behind every real row
                                       * in the top-k of tdmpc2.py:185 (as the -inf rows of a masked topk would), so mean / std / action
                                       * are those of a plan over the true count.  tdmpc2_amd.NativePlanner does all of this itself. */
} tdmpc2_plan_cfg;

/* Two kernel families implement the same math (results agree to fp32 round-off):
 *   FUSED   one persistent workgroup per 64 sample rows keeps activations in LDS for a whole CEM
 *           iteration; built for latent_dim == mlp_dim == 512 (every 5M model), episodic or not.
 *   LAYERED one MFMA GEMM launch per nn.Linear over all E*N rows, activations in HBM; any
 *           latent_dim / mlp_dim that are multiples of 32 (1M ... 317M models), episodic or not.
 * AUTO picks FUSED when the configuration fits it, else LAYERED. */
enum tdmpc2_path { TDMPC2_PATH_AUTO = 0, TDMPC2_PATH_FUSED = 1, TDMPC2_PATH_LAYERED = 2 };

/* Arithmetic of the nn.Linear contractions (everything else is fp32 in both modes):
 *   FP32       v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation (bitwise an fmaf chain).
 *   SPLIT_F16  every fp32 operand is carried as hi + lo f16 pieces (22 significand bits) and a product is
 *              a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation: fp32-class
 *              error (checked against fp64 in the tests) at up to 16/3 of the fp32 matrix rate.
 * AUTO = SPLIT_F16. */
enum tdmpc2_precision { TDMPC2_PREC_AUTO = 0, TDMPC2_PREC_FP32 = 1, TDMPC2_PREC_SPLIT_F16 = 2 };

enum tdmpc2_net {
    TDMPC2_NET_DYNAMICS = 0,    /* WorldModel._dynamics     world_model.py:26 */
    TDMPC2_NET_REWARD = 1,      /* WorldModel._reward       world_model.py:27 */
    TDMPC2_NET_PI = 2,          /* WorldModel._pi           world_model.py:29 */
    TDMPC2_NET_Q = 3,           /* WorldModel._Qs (stacked) world_model.py:30 */
    TDMPC2_NET_TERMINATION = 4, /* WorldModel._termination  world_model.py:28 */
    TDMPC2_NET_TARGET_Q = 5     /* WorldModel._target_Qs (stacked, optional)  world_model.py:38-53 */
};

enum tdmpc2_status {
    TDMPC2_OK = 0,
    TDMPC2_ERR_INVALID = 1,      /* bad argument */
    TDMPC2_ERR_UNSUPPORTED = 2,  /* configuration outside what the kernels are built for */
    TDMPC2_ERR_HIP = 3,          /* a HIP runtime call failed */
    TDMPC2_ERR_STATE = 4         /* e.g. run before all weights are bound */
};

/* The reference's six RNG draw sites as input tensors.
 * Passing a tape makes a plan a pure function of its inputs (parity runs);
 * passing NULL selects the in-kernel Philox4x32-10 generator (fast mode). */
typedef struct tdmpc2_noise {
    const float *pi_traj_eps;  /* [E,H,P,A]      randn_like, world_model.py:156 via tdmpc2.py:158,160 */
    const float *sample_eps;   /* [E,I,H,N-P,A]  randn,      tdmpc2.py:176 */
    const float *pi_eps;       /* [E,I,N,A]      randn_like, world_model.py:156 via tdmpc2.py:135 */
    const int32_t *qidx;       /* [E,I,2]        randperm(nq)[:2], world_model.py:212 */
    const float *gumbel_exp;   /* [E,K]          exponential_(), math.py:90 */
    const float *final_eps;    /* [E,A]          randn, tdmpc2.py:204 (unused when eval_mode) */
} tdmpc2_noise;

/* The same six tensors as OUTPUTS (tdmpc2_plan_export_noise): device pointers the library writes; any may be NULL. */
typedef struct tdmpc2_noise_out {
    float *pi_traj_eps;  /* [n,H,P,A] */
    float *sample_eps;   /* [n,I,H,N-P,A] */
    float *pi_eps;       /* [n,I,N,A] */
    int32_t *qidx;       /* [n,I,2] */
    float *gumbel_exp;   /* [n,K] */
    float *final_eps;    /* [n,A] */
} tdmpc2_noise_out;

/* Optional stage-wise outputs (any pointer may be NULL). */
typedef struct tdmpc2_debug {
    float *value;       /* [E,I,N]   value after nan_to_num, tdmpc2.py:184 */
    int32_t *elite_idx; /* [E,I,K]   topk indices (value desc, index asc on ties), tdmpc2.py:185 */
    float *score;       /* [E,I,K]   normalised elite scores, tdmpc2.py:190-191 */
    float *mean;        /* [E,I,H,A] tdmpc2.py:192,196 */
    float *std;         /* [E,I,H,A] tdmpc2.py:193-194,197 */
    float *actions;     /* [E,I,H,N,A] sampled actions of every iteration, tdmpc2.py:176-181 */
} tdmpc2_debug;

int tdmpc2_plan_abi_version(void);
const char *tdmpc2_last_error(void);

/* Allocate a planner for `cfg` on cfg->device.  TDMPC2_ERR_UNSUPPORTED if the
 * configuration is outside the compiled kernels' envelope. */
int tdmpc2_plan_create(const tdmpc2_plan_cfg *cfg, tdmpc2_plan_t **out);
void tdmpc2_plan_destroy(tdmpc2_plan_t *h);

/* Bytes of device memory held by the handle (packed weights + workspace). */
uint64_t tdmpc2_plan_device_bytes(const tdmpc2_plan_t *h);

/* The kernel family / arithmetic the handle resolved to (never the AUTO values). */
int tdmpc2_plan_path(const tdmpc2_plan_t *h);
int tdmpc2_plan_precision(const tdmpc2_plan_t *h);

/* Hand one layer of one network to the planner.  Pointers are device fp32 in
 * the checkpoint's own layout (nn.Linear: W[out,in] row-major, b[out];
 * LayerNorm ln_g/ln_b[out], NULL for the plain output layers).  For
 * TDMPC2_NET_Q every tensor carries the leading ensemble dim num_q
 * ("_Qs.params.<layer>.<name>", tdmpc2/common/layers.py:167-199).  The first
 * layer's input columns are ordered [z | task_emb | action]
 * (world_model.py:118-120).  The library re-packs into its own MFMA fragment
 * layout; the source buffers may be freed after the stream reaches this call. */
int tdmpc2_plan_bind_weights(tdmpc2_plan_t *h, int net, int layer, const float *W, const float *b,
                             const float *ln_g, const float *ln_b, int out_features, int in_features,
                             void *stream);

/* One plan per environment: everything of TDMPC2._plan after encode().
 *   z0         [E,L]     latent from WorldModel.encode (tdmpc2.py:153)
 *   task_emb   [E,T]     looked-up (max_norm-renormalised) task embedding rows, NULL if !multitask
 *   act_mask   [E,A]     WorldModel._action_masks[task], NULL if !multitask
 *   disc_pow   [E,H+1]   discount^0..discount^H exactly as tdmpc2.py:126,130-132 accumulates them
 *   prev_mean  [E,H,A]   in: TDMPC2._prev_mean, out: new mean (tdmpc2.py:166-167,205)
 *   t0         [E] u8    first step of an episode (no warm start)
 *   eval_mode            0: add std*noise to the chosen action (tdmpc2.py:203-204)
 *   tape                 NULL -> Philox(seed); `seed` is ignored when a tape is given
 *   action     [E,A]     out: the clamped action (tdmpc2.py:206) */
int tdmpc2_plan_run(tdmpc2_plan_t *h, int n_envs, const float *z0, const float *task_emb,
                    const float *act_mask, const float *disc_pow, float *prev_mean, const uint8_t *t0,
                    int eval_mode, const tdmpc2_noise *tape, uint64_t seed, float *action,
                    const tdmpc2_debug *dbg, void *stream);

/* The in-kernel generator, made visible (parity of the fast mode).  A plan run with tape = NULL draws its noise from
 * Philox4x32-10 keyed by (seed, the handle's call counter at entry, draw site, CEM iteration, environment, element) --
 * the six sites of the reference (tdmpc2/tdmpc2.py:176,204; tdmpc2/common/world_model.py:156,212; tdmpc2/common/math.py:90).
 * export_noise writes exactly those draws, for environments [env_first, env_first + n_envs) of such a call, as the tensors
 * of a noise tape: tdmpc2_plan_run(..., tape = the export, ...) then reproduces the tape = NULL plan bit for bit, and the
 * same tape replays through a CPU restatement of the reference.  Every kernel family draws the same numbers.
 *   call  = the value tdmpc2_plan_call_counter returned BEFORE the run being reproduced (run / run_obs / shard_begin /
 *           policy_value / td_target each consume one count). */
int tdmpc2_plan_export_noise(tdmpc2_plan_t *h, int env_first, int n_envs, uint64_t seed, uint32_t call,
                             const tdmpc2_noise_out *out, void *stream);
/* The per-handle call counter mixed into the Philox key (so that consecutive plans under one seed draw fresh noise).
 * Ranks that shard ONE plan (tdmpc2_plan_shard_*) must agree on it: set it from one rank's value before shard_begin. */
int tdmpc2_plan_call_counter(const tdmpc2_plan_t *h, uint32_t *next_call);
int tdmpc2_plan_set_call_counter(tdmpc2_plan_t *h, uint32_t next_call);

/* State-observation encoder: WorldModel.encode for cfg.obs == 'state'
 * (tdmpc2/common/world_model.py:103-112) with the network of layers.enc (tdmpc2/common/layers.py:153-164):
 * n_layers NormedLinear blocks, Mish after all but the last, SimNorm after the last.  Pixel observations are encoded
 * by the host framework's conv module (tdmpc2_amd/layers.py: conv) and enter through tdmpc2_plan_run as latents.  bind_encoder takes one nn.Linear (`W` [out, in] row-major, `b` [out]) and its LayerNorm (`ln_g`,
 * `ln_b` [out]) per call, DEVICE pointers, copied (weights transposed) into library memory; layer 0 takes
 * obs_dim + task_dim inputs, the last layer has latent_dim outputs; widths up to 4096. */
int tdmpc2_plan_bind_encoder(tdmpc2_plan_t *h, int layer, int n_layers, const float *W, const float *b,
                             const float *ln_g, const float *ln_b, int out_features, int in_features,
                             void *stream);
/*   obs [E, obs_dim], task_emb [E, T] (NULL if !multitask; the rows the reference concatenates in
 *   WorldModel.task_emb, world_model.py:88-101) -> z_out [E, L].  n_envs is not limited by max_envs unless a layer
 *   is wider than 1024 (those encoders run layer by layer through a workspace sized for max_envs rows). */
int tdmpc2_plan_encode(tdmpc2_plan_t *h, int n_envs, const float *obs, int obs_dim, const float *task_emb,
                       float *z_out, void *stream);
/* TDMPC2._plan from the observation on (tdmpc2/tdmpc2.py:152-206): encode into library memory, then exactly
 * tdmpc2_plan_run.  One call per environment step, no framework kernel in between. */
int tdmpc2_plan_run_obs(tdmpc2_plan_t *h, int n_envs, const float *obs, int obs_dim, const float *task_emb,
                        const float *act_mask, const float *disc_pow, float *prev_mean, const uint8_t *t0,
                        int eval_mode, const tdmpc2_noise *tape, uint64_t seed, float *action, void *stream);

/* Training-side consumers of the planner's layer code, forward only, no gradients.  Both kernel
 * families, single-task and multitask models.  The fused family takes any number of rows; the layered family at most
 * max_envs * num_samples rows per call (its activation workspace).  Multitask tables are (re)built inside the call and
 * their storage grows on demand (the first such call allocates: keep it outside a hipGraph capture).
 *
 * policy_value: a = pi(z) (world_model.py:144-184, sampled with pi_eps [n_rows, A] or Philox(seed) when NULL), then the
 * two heads qidx[0..1] (device int32[2]; NULL: drawn like randperm(num_q)[:2], world_model.py:212) of the online ensemble
 * (use_target = 0: what update_pi evaluates, tdmpc2.py:220-221) or of the target ensemble (use_target = 1: bind net
 * TDMPC2_NET_TARGET_Q first), reduced 'avg' (reduce_min = 0) or 'min' (1), world_model.py:213-216.
 *   z [n_rows, L] -> action [n_rows, A] (may be NULL), q [n_rows]. */
int tdmpc2_plan_policy_value(tdmpc2_plan_t *h, int n_rows, const float *z, int use_target, int reduce_min,
                             const float *pi_eps, const int32_t *qidx, uint64_t seed, float *action, float *q,
                             void *stream);
/* TDMPC2._td_target (tdmpc2.py:239-254): td = reward + discount (1 - terminated) min_{2 target heads} Q(next_z, pi(next_z)).
 *   next_z [n_rows, L] (the reference's [H, B, L] flattened), reward, terminated [n_rows] -> td [n_rows]. */
int tdmpc2_plan_td_target(tdmpc2_plan_t *h, int n_rows, const float *next_z, const float *reward,
                          const float *terminated, float discount, const float *pi_eps, const int32_t *qidx,
                          uint64_t seed, float *td, void *stream);

/* The same on multitask models, where a training batch carries ONE TASK PER ROW (WorldModel.task_emb with a task
 * vector, world_model.py:88-101; the reference repeats task [B] over the H leading rows of next_z [H, B, L]):
 * the row -> task map plus the per-task tables the reference indexes with it.  All DEVICE pointers. */
typedef struct tdmpc2_task_tables {
    const int32_t *task_ids;  /* [n_rows]            task of each row */
    const float *task_emb;    /* [n_tasks, T]        WorldModel._task_emb rows, max_norm renorm applied (world_model.py:21) */
    const float *act_mask;    /* [n_tasks, A]        WorldModel._action_masks (world_model.py:22-24) */
    const float *discount;    /* [n_tasks] or NULL   TDMPC2.discount (tdmpc2.py:35-37); td_target only */
    int32_t n_tasks;
} tdmpc2_task_tables;
/* `tasks` must be NULL for single-task handles and non-NULL for multitask ones. */
int tdmpc2_plan_policy_value_mt(tdmpc2_plan_t *h, int n_rows, const float *z, const tdmpc2_task_tables *tasks,
                                int use_target, int reduce_min, const float *pi_eps, const int32_t *qidx, uint64_t seed,
                                float *action, float *q, void *stream);
/* `discount` is used by single-task handles; multitask handles take tasks->discount[task of the row]. */
int tdmpc2_plan_td_target_mt(tdmpc2_plan_t *h, int n_rows, const float *next_z, const float *reward,
                             const float *terminated, float discount, const tdmpc2_task_tables *tasks,
                             const float *pi_eps, const int32_t *qidx, uint64_t seed, float *td, void *stream);

/* Packed weight file (the native counterpart of TDMPC2.save / load, tdmpc2.py:72-95).
 * export_packed copies everything the binds produced -- weights in MFMA fragment order (hi / lo split and scaled for the
 * SPLIT_F16 arithmetic), padded biases, LayerNorm parameters, task-embedding columns, per-layer scale records, the
 * transposed encoder, the target ensemble when bound -- into ONE host buffer of packed_size bytes; import_packed
 * restores a handle created with the same model dimensions, kernel family and arithmetic from such a buffer with plain
 * host-to-device copies (no packing kernels, no fp32 checkpoint on the device).  The blob is specific to (path, precision);
 * TDMPC2_ERR_INVALID on any mismatch.  Both synchronise `stream`. */
int tdmpc2_plan_packed_size(tdmpc2_plan_t *h, uint64_t *bytes);
int tdmpc2_plan_export_packed(tdmpc2_plan_t *h, void *host_buf, uint64_t bytes, void *stream);
int tdmpc2_plan_import_packed(tdmpc2_plan_t *h, const void *host_buf, uint64_t bytes, void *stream);

/* TDMPC2._estimate_value on given action sequences (stage-wise parity).
 *   actions [E,H,N,A], pi_eps [E,N,A], qidx [E,2] -> value [E,N] (before nan_to_num). */
int tdmpc2_plan_estimate_value(tdmpc2_plan_t *h, int n_envs, const float *z0, const float *task_emb,
                               const float *act_mask, const float *disc_pow, const float *actions,
                               const float *pi_eps, const int32_t *qidx, float *value, void *stream);

/* The same, additionally dumping the activation tile after every fused phase (layer-level parity:
 * each fused stage is checked against its unfused counterpart at its own scale).
 *   trace_tiles   [E*N/64, 5H+7, 64, L]  per 64-row tile, in execution order: for t < H {reward h1, reward h2,
 *                 dynamics h1, dynamics h2, z_{t+1}}, then {pi h1, pi h2, z_H, Q_a h1, Q_a h2, Q_b h1, Q_b h2}
 *   trace_scalars [E, N, H+2+A]          r_0..r_{H-1}, Q_a, Q_b, a_H[A]
 * Either may be NULL.  LAYERED handles dump scalars only (trace_tiles must be NULL there). */
int tdmpc2_plan_estimate_value_trace(tdmpc2_plan_t *h, int n_envs, const float *z0, const float *task_emb,
                                     const float *act_mask, const float *disc_pow, const float *actions,
                                     const float *pi_eps, const int32_t *qidx, float *value,
                                     float *trace_tiles, float *trace_scalars, void *stream);

/* Elite select + refit on given values (stage-wise parity).
 *   value [E,N] in/out (nan_to_num applied), actions [E,H,N,A], act_mask [E,A]|NULL
 *   -> mean,std [E,H,A]; score [E,K]; elite_idx [E,K] (outputs may be NULL). */
int tdmpc2_plan_refit(tdmpc2_plan_t *h, int n_envs, float *value, const float *actions,
                      const float *act_mask, float *mean, float *std, float *score,
                      int32_t *elite_idx, void *stream);

/* ONE plan sharded over G GPUs (last row: worth it for 317M-class models at E = 1).  The N sample rows of
 * every plan are split over the ranks; weights, set-up, action sampling (same tape or same Philox seed on every rank) and
 * the elite selection + refit are replicated.  Per plan: shard_begin once (= the prologue of tdmpc2_plan_run: warm start,
 * policy-prior trajectories, tdmpc2.py:154-170); per CEM iteration shard_values for this rank's rows
 * [row_begin, row_end) (a multiple of 64 rows, FUSED, or 128, LAYERED) writing value[E, N] at those rows only, then the
 * HOST all-gathers the value slices (RCCL, N * 4 bytes per plan), then shard_refit on the complete value[E, N]
 * (tdmpc2.py:184-197; at iter == iterations - 1 also the final pick, tdmpc2.py:199-206).  With one rank and the full row
 * range the three calls compute what tdmpc2_plan_run computes.  tdmpc2_amd/dist.py: sharded_plan.
 * Faults: a bounded inter-workgroup wait that gave up in ANY of the plan's calls makes the final pick return NaN actions and
 * keep prev_mean (every call consumes the handle's error word, so the library keeps a second, sticky word per plan in flight);
 * tdmpc2_plan_take_fault after a sync reports it, and every rank has to plan the step again (dist.sharded_plan all-reduces the
 * verdict).  The sticky word is written by the host: synchronise with the final shard_refit (the caller needs its action
 * anyway) before the next shard_begin on the same handle. */
int tdmpc2_plan_shard_begin(tdmpc2_plan_t *h, int n_envs, const float *z0, const float *task_emb, const float *act_mask,
                            const float *prev_mean, const uint8_t *t0, const tdmpc2_noise *tape, uint64_t seed, void *stream);
int tdmpc2_plan_shard_values(tdmpc2_plan_t *h, int n_envs, int iter, int row_begin, int row_end, const float *z0,
                             const float *act_mask, const float *disc_pow, const tdmpc2_noise *tape, uint64_t seed,
                             float *value, void *stream);
int tdmpc2_plan_shard_refit(tdmpc2_plan_t *h, int n_envs, int iter, float *value, const float *act_mask, float *prev_mean,
                            int eval_mode, const tdmpc2_noise *tape, uint64_t seed, float *action, const tdmpc2_debug *dbg,
                            void *stream);

/* Tuning knobs that never change results beyond fp32 round-off.  key TDMPC2_TUNE_ROWS_PER_WORKGROUP: sample rows a
 * fused split-arithmetic rollout workgroup owns -- 0 = automatic (32 when a call brings too few plans to occupy the
 * chip, i.e. single-environment latency; 64 otherwise), or 32 / 64 to force one.  key TDMPC2_TUNE_FOLD_REFIT (fused
 * family): 1 = the last workgroup of a plan to finish its rollouts does the elite selection + refit (tdmpc2.py:184-206)
 * inside the rollout launch, one launch per CEM iteration; 0 = always a launch of its own (k_refit); 2 (default) = inside
 * the rollout launch when the call's workgroups fit the chip in one round (few plans: single-environment latency), a
 * launch of its own otherwise (many plans: the in-launch refits would delay the next round of workgroups).
 * key TDMPC2_TUNE_CLUSTER (fused family, f16x2-split arithmetic): single-plan latency path -- every 512-wide
 * layer of a 32-row sample tile is split over a cluster of 8 workgroups on 8 CUs that exchange the layer's raw sums through
 * L2 (tdmpc2_amd/csrc/cluster_kernels.cuh); used when all of a call's clusters fit the chip at once (one or two plans of
 * 512 samples on 256 CUs).  0 = never, 1 = whenever the call fits, 2 (default) = 1, and for a SINGLE non-episodic plan every
 * launch -- the first included, which also folds the policy-prior trajectories in -- gives each tile a second cluster that runs
 * the reward chain (and the second Q head) beside the dynamics chain (cluster2_kernels.cuh: all 256 CUs, 16 instead of 23 hand-overs on the critical path; identical values).
 * key TDMPC2_TUNE_FUSE_LN (layered family, f16x2-split arithmetic): 1 (default) = the LayerNorm + Mish / SimNorm + operand split
 * of every NormedLinear (tdmpc2/common/layers.py:94-118) runs in the epilogue of its GEMM -- the column blocks of a row block
 * exchange per-row (mean, M2) partials through L2, a bounded wait like the cluster path's (tdmpc2_plan_take_fault) --;
 * 0 = fp32 pre-activations to HBM and a row kernel per layer.
 * key TDMPC2_TUNE_REARM_AFTER: consecutive clean calls after which a handle that was downgraded by a reported wait goes back to
 * the CLUSTER / FUSE_LN paths (default 8 -- 64 before ABI 8; 0 = never: the downgrade is for good, as in ABI <= 6).  See
 * tdmpc2_plan_take_fault.
 * key TDMPC2_TUNE_SAFE_ONCE (ABI 8): 1 = the NEXT whole plan on this handle (tdmpc2_plan_run / run_obs, or shard_begin .. the last
 * shard_refit) runs on the paths without inter-workgroup waits, whatever CLUSTER / FUSE_LN say; the caller's settings, the
 * downgrade state and the re-arm counter are not touched and the flag clears itself when that plan has been enqueued (the
 * re-plan of a sharded plan after a reported wait: dist.sharded_plan).
 * key TDMPC2_TUNE_KSPLIT (ABI 8; layered family, f16x2-split arithmetic): the 256 x 256 output tiles of a GEMM's last, partly
 * filled round of the chip can each be computed by 2-4 workgroups over disjoint K ranges whose partial sums meet in a workspace
 * and are added in a fixed order (tdmpc2_amd/csrc/layered_wide.cuh, tile_order.h: gemm_w_order).  0 = never: every tile whole -- a
 * plan then computes the same bits alone, in any batch and with its rows split over ranks; 1 = whenever the round arithmetic
 * says so (measured: slower on launches that fill the chip -- the partial sums' traffic); 2 (default) = only for launches of
 * 16 .. 128 tiles, which leave most of the chip idle (one or two plans of the 317M model: single-plan latency -9 %).  Same values
 * to fp32 round-off (1e-5 of the trajectory values); the bits of a plan then depend on how many plans share the call.
 * key TDMPC2_TUNE_FEWROW (ABI 9; layered family, f16x2-split arithmetic): 1 (default) = calls with so few sample rows that one round
 * of 64 x 256 output tiles does not fill the chip -- single plans, the reference's own call pattern (evaluate.py:80): the 48M
 * model up to 4 plans, the 317M model 1 -- run every nn.Linear as K-PARTS of such tiles (tdmpc2_amd/csrc/layered_mid.cuh: an 8-wave
 * LDS-DMA ring GEMM writing raw partial sums) followed by a row kernel that adds the parts in a fixed order and applies the
 * NormedLinear / two-hot / policy-head math; two chains per launch, one stream, no workgroup ever waits for another one (no fault
 * path).  Same values to fp32 round-off; needs TDMPC2_TUNE_KSPLIT != 0 (with KSPLIT = 0 a plan keeps computing the same bits alone and
 * in any batch).  0 = the per-layer tiles of the batch path for every call size.
 * key TDMPC2_TUNE_WAIT_US (ABI 9): the wall-clock bound of the inter-workgroup waits in microseconds (default 5000; 100 .. 10 000 000).
 * A handle that shares its GPU with another process's multi-millisecond kernels may want more; the hot path never reads it (the
 * clock is only consulted from the 256th poll of a wait on).
 * keys TDMPC2_TUNE_EXPERT + tdmpc2_expert_knob (ABI 9): the measurement knobs of the layered family's tile choice -- thresholds between
 * kernels that compute the same values to fp32 round-off.  They were environment variables of the library until ABI 8; the library
 * now reads exactly the environment variables listed in INTEGRATION.md section C and nothing else.  value INT32_MIN = the default. */
enum tdmpc2_tuning { TDMPC2_TUNE_ROWS_PER_WORKGROUP = 0, TDMPC2_TUNE_FOLD_REFIT = 1, TDMPC2_TUNE_CLUSTER = 2, TDMPC2_TUNE_FUSE_LN = 3,
                     TDMPC2_TUNE_REARM_AFTER = 4, TDMPC2_TUNE_SAFE_ONCE = 5, TDMPC2_TUNE_KSPLIT = 6, TDMPC2_TUNE_FEWROW = 7,
                     TDMPC2_TUNE_WAIT_US = 8, TDMPC2_TUNE_EXPERT = 100 };
/* (what each knob decides: tdmpc2_amd/csrc/layered_host.cuh; defaults in tdmpc2_amd/csrc/handle.h) */
enum tdmpc2_expert_knob { TDMPC2_X_GEMM_W256_MIN = 0, TDMPC2_X_GEMM_W_SPLIT_MIN, TDMPC2_X_GEMM_W_SPLIT_MAX, TDMPC2_X_GEMM_W_SPLIT_OVH,
                          TDMPC2_X_KSPLIT_AUTO_LO, TDMPC2_X_KSPLIT_AUTO_MIN, TDMPC2_X_GEMM_W_XCD_ROWS, TDMPC2_X_GEMM_NCT1,
                          TDMPC2_X_GEMM_WIDE_MIN, TDMPC2_X_GEMM_RT4, TDMPC2_X_GEMM_FILL_PERMILLE, TDMPC2_X_GEMM_FILL_HEAD_PERMILLE,
                          TDMPC2_X_GEMM_SD1, TDMPC2_X_GEMM_XCD_ROWS, TDMPC2_X_GEMM_COL_PAD, TDMPC2_X_TWOHOT_UNFUSED, TDMPC2_X_Z0_SHARED_OFF,
                          TDMPC2_X_MID_PARTS_MAX, TDMPC2_X_MID_FUSE_LN, TDMPC2_X_MID_SPLIT_XCD, TDMPC2_X_MID_PIFOLD, TDMPC2_X_COUNT };
int tdmpc2_plan_set_tuning(tdmpc2_plan_t *h, int key, int value);

/* Fault report of the paths whose workgroups wait for each other: the cluster path (TDMPC2_TUNE_CLUSTER) and the NormedLinear
 * epilogue inside the layered family's GEMMs (TDMPC2_TUNE_FUSE_LN).  Those waits are bounded by the wall clock (5 ms of the constant
 * 100 MHz clock -- a poll count worth a third of a second before ABI 8; a healthy wait is microseconds to one tile's run time); when one gives up -- another process or a foreign kernel held
 * the compute units -- the call in flight is invalid AND SAYS SO: a plan's action[E, A] comes back as NaN with prev_mean left as
 * it was (the step can simply be planned again); tdmpc2_plan_td_target / policy_value (LAYERED family) return NaN in every
 * element of out[] (and action[]).  The reference has no analogue (its only guard is the nan_to_num of tdmpc2.py:184).
 * Call this after synchronising the stream of such a call: *faults = number of invalid calls since the last take_fault (0 = none).
 * After a fault the handle runs the paths without inter-workgroup waits; it switches back to the fast ones after
 * TDMPC2_TUNE_REARM_AFTER (default 8) consecutive clean calls, doubling that number (up to 4096) every time a fault follows a
 * re-arm and forgetting the back-off after a long clean run; an explicit tdmpc2_plan_set_tuning(CLUSTER / FUSE_LN) re-arms at
 * once.  Calls enqueued back to back without a synchronisation in between each carry their own verdict: the word a call's last
 * kernel reads is raised on the device and cleared by the NEXT call in stream order, never by the host (which looks at a
 * separate sticky word with one atomic exchange: the count reported here is "looks that found it set", at most one per API
 * call -- a lower bound on the waits that gave up).  The later calls of a
 * sharded plan (shard_values / shard_refit) do not clear it: a wait that gave up in any iteration invalidates the final pick. */
int tdmpc2_plan_take_fault(tdmpc2_plan_t *h, int *faults);

/* The verdict of the call(s) in flight WITHOUT a host synchronisation (ABI 8): enqueues a 4-byte copy of the device-visible
 * verdict word (0 = no bounded wait has given up since the last call that cleared it: tdmpc2_plan_run*, shard_begin, ...) into
 * dst_dev[0] on `stream`, behind the kernels enqueued so far.  A rank of a sharded plan appends the word to the value slice it
 * all-gathers anyway, so that every rank learns every rank's verdict with the one synchronisation the caller needs for the
 * action (dist.sharded_plan); the host-side bookkeeping (downgrade, re-arm, take_fault) is untouched.  No reference analogue. */
int tdmpc2_plan_fault_word(tdmpc2_plan_t *h, uint32_t *dst_dev, void *stream);

/* The fault history of a handle (no synchronisation, nothing consumed): how often a bounded wait has given up, how long ago the
 * last one was, whether the handle is currently on the downgraded paths and how far the re-arm counter has got. */
typedef struct tdmpc2_fault_info {
    int32_t faults_total;        /* bounded waits that gave up since tdmpc2_plan_create */
    int32_t rearms;              /* times the fast paths were switched back on */
    int32_t degraded;            /* 1: running the paths without inter-workgroup waits right now */
    int32_t clean_calls;         /* consecutive clean calls since the downgrade (re-arm at rearm_after) */
    int32_t rearm_after;         /* current threshold (doubles after a re-arm, TDMPC2_TUNE_REARM_AFTER resets it; 0: never) */
    int32_t reserved;
    double seconds_since_fault;  /* wall-clock seconds since the last fault was noted; -1: never */
} tdmpc2_fault_info;
int tdmpc2_plan_fault_info(tdmpc2_plan_t *h, tdmpc2_fault_info *info);

/* Live timing of the dominant (rollout) stage: after set_profiling(h, n > 0) every rollout launch
 * (FUSED: one ks_rollout kernel; LAYERED: the GEMM / row-kernel sequence of one CEM iteration's
 * _estimate_value) is bracketed by HIP events recorded on the caller's stream (up to n are kept;
 * n = 0 turns it off).  profile_read synchronises those events, returns their summed duration and the number of
 * launches measured, and rewinds the buffer. */
int tdmpc2_plan_set_profiling(tdmpc2_plan_t *h, int max_launches);
int tdmpc2_plan_profile_read(tdmpc2_plan_t *h, float *rollout_ms_total, int *rollout_launches);

#ifdef __cplusplus
}
#endif
#endif /* TDMPC2_PLAN_H */
