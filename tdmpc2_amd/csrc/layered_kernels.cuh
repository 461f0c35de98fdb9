// Layer-at-a-time planner kernels for ANY (latent_dim, mlp_dim): the 19M / 48M / 317M world models
// (c3, c4, c5) and episodic (termination-head) planning at every size.
//
// The fused kernel (ks_rollout) keeps a 64-row activation tile in LDS across a whole CEM iteration, which only
// fits for 512-wide layers.  Here activations live in HBM (288 GB: [E*N, mlp_dim] fp32 per buffer) and every
// nn.Linear of the reference becomes one LDS-tiled fp32-MFMA GEMM launch over all E*N sample rows of all
// plans, followed by one row-wise kernel for the LayerNorm / Mish / SimNorm / two-hot / policy-head math
// (reference: tdmpc2/common/layers.py:94-133, tdmpc2/common/math.py:12-83, tdmpc2/common/world_model.py:114-216).
// Included by tdmpc2_plan.hip inside its anonymous namespace (device helpers mish_f, symexp_f, group_*, rng_* are
// defined there).
#pragma once


// out[r, c] = sum_k A[r, k] * W[c, k] + bias(r)[c]          (pre-activation of one nn.Linear)
#ifndef TDMPC2_GEMM_ROWMAJOR_TILES
#define TDMPC2_GEMM_ROWMAJOR_TILES 0
#endif
// Output tile of workgroup b.  Hardware places block b on XCD b % 8; every XCD gets a contiguous run of tiles, and inside
// the run the tiles that are in flight together (about 64 per XCD) form a patch of 8 column blocks x several row blocks
// instead of whole rows of column blocks, so that they share both the A row panels and the W column panels streaming
// through that XCD's 4 MB L2 (c4: 16 column blocks of 4 MB weights each; fabric-side reads per GEMM 1.9 GB before).
__device__ __forceinline__ void gemm_tile_of_block(int b, int nb, int ncolblk, int &rb, int &cb) {
    if (nb % 8 != 0) {
        rb = b / ncolblk;
        cb = b % ncolblk;
        return;
    }
    const int nloc = nb / 8, x = b % 8, t = b / 8;  // XCD x, its t-th tile
    constexpr int CS = 8;
    if (ncolblk % CS == 0 && nloc % ncolblk == 0 && !TDMPC2_GEMM_ROWMAJOR_TILES) {
        const int rows_loc = nloc / ncolblk;  // row blocks owned by this XCD
        const int per_strip = rows_loc * CS;  // tiles of one strip of CS column blocks
        const int strip = t / per_strip, u = t % per_strip;
        rb = x * rows_loc + u / CS;
        cb = strip * CS + u % CS;
        return;
    }
    const int tile = x * nloc + t;
    rb = tile / ncolblk;
    cb = tile % ncolblk;
}

struct GemmParams {
    const float *A;     // [Rp, lda] fp32 row-major, Rp a multiple of GBM
    int lda;
    int K;              // contraction length, multiple of GBK (weights zero-padded)
    const float *wp;    // packed [CT][K/8][64][4] (k_pack_weight), + sel * w_sel_stride for ensembles
    long w_sel_stride;
    int CT;             // output column tiles of 32
    int ncolblk;        // ceil(CT / 4)
    const float *bias;  // bias(r) = bias + env(r) * bias_env_stride + sel * bias_sel_stride, CT*32 valid floats
    long bias_env_stride, bias_sel_stride;
    const int *sel;     // per-plan ensemble member (Q heads) or null; requires rows_per_env % GBM == 0
    long sel_stride;
    int rows_per_env;   // env(r) = r / rows_per_env
    const int *row_env; // or: env(r) = row_env[r] for the per-row bias (training batches: one task per row); sel keeps r / rows_per_env
    float *out;         // [Rp, ldo]
    int ldo;
};

__global__ __launch_bounds__(GTHREADS, 2) void g_gemm(GemmParams p) {
    __shared__ __attribute__((aligned(16))) float As[2][GBM * GLD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    int rb, cb;
    gemm_tile_of_block(blockIdx.x, gridDim.x, p.ncolblk, rb, cb);
    const int row0 = rb * GBM;
    const int sel = p.sel ? p.sel[(size_t)(row0 / p.rows_per_env) * p.sel_stride] : 0;
    const int KB = p.K / 8;
    const int ct0 = cb * 4 + wc * 2;
    const bool v0 = ct0 < p.CT, v1 = ct0 + 1 < p.CT;
    const f32x4 *wbase = reinterpret_cast<const f32x4 *>(p.wp + (size_t)sel * p.w_sel_stride);
    const f32x4 *w0 = wbase + (size_t)(v0 ? ct0 : p.CT - 1) * KB * 64 + lane;
    const f32x4 *w1 = wbase + (size_t)(v1 ? ct0 + 1 : p.CT - 1) * KB * 64 + lane;

    // A staging: thread -> rows (tid >> 3) + 32 i, float4 column tid & 7 of the 32-wide chunk (128 B per row, coalesced)
    const int srow = tid >> 3, sc4 = tid & 7;
    const float *ag = p.A + (size_t)(row0 + srow) * p.lda + 4 * sc4;
    f32x4 stage[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) stage[i] = *reinterpret_cast<const f32x4 *>(ag + (size_t)(32 * i) * p.lda);
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(&As[0][(srow + 32 * i) * GLD + 4 * sc4]) = stage[i];
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;

    const int i32 = lane & 31, h = lane >> 5;
    const int nchunks = p.K / GBK;
    f32x4 bn0 = w0[0], bn1 = w1[0];
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                stage[i] = *reinterpret_cast<const f32x4 *>(ag + (size_t)(32 * i) * p.lda + (size_t)(c + 1) * GBK);
        }
        const float *as = &As[c & 1][0];
        const float *a0p = as + (wr * 64 + i32) * GLD + 4 * h;
        const float *a1p = a0p + 32 * GLD;
#pragma unroll
        for (int kb = 0; kb < GBK / 8; ++kb) {
            const f32x4 b0 = bn0, b1 = bn1;
            const int kn = c * (GBK / 8) + kb + 1;
            const int knc = kn < KB ? kn : KB - 1;
            bn0 = w0[(size_t)knc * 64];
            bn1 = w1[(size_t)knc * 64];
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(a0p + kb * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(a1p + kb * 8);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], b0[r], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], b1[r], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], b0[r], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], b1[r], acc[1][1], 0, 0, 0);
            }
        }
        if (more) {
            float *dst = &As[(c + 1) & 1][0];
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(dst + (srow + 32 * i) * GLD + 4 * sc4) = stage[i];
        }
        __syncthreads();
    }

    // epilogue: + bias, store.  C/D fragment: lane holds column (lane & 31), rows (reg&3) + 8 (reg>>2) + 4 (lane>>5).
    const float *bsel = p.bias + (size_t)sel * p.bias_sel_stride;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (!(c == 0 ? v0 : v1)) continue;
        const int col = (ct0 + c) * 32 + i32;
        const float bshared = p.bias_env_stride == 0 ? bsel[col] : 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = row0 + wr * 64 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                float bv = bshared;
                if (p.bias_env_stride != 0)
                    bv = bsel[(size_t)(p.row_env ? p.row_env[row] : row / p.rows_per_env) * p.bias_env_stride + col];
                p.out[(size_t)row * p.ldo + col] = acc[rt][c][reg] + bv;
            }
    }
}

// ---------------------------------------------------------------- row-wise kernels: one wavefront per row

// In place: x <- ACT(LayerNorm(x)) over `width` columns of each row (width % 4 == 0).  ACT 0 Mish, 1 SimNorm(8).
// Per-plan ensemble member selection for the LayerNorm affine parameters like g_gemm.
struct LnActParams {
    float *x;
    int ld, width, rows, rows_per_env;
    const float *g, *b;
    long gb_sel_stride;
    const int *sel;
    long sel_stride;
    char *out;   // split arithmetic: the fragment-packed operand buffer the activated row goes to (x is then read-only) ...
    int KBo;     // ... and its k16-blocks per row
    const float *ascale;   // split arithmetic: operand scale of this layer's output (LayerScal::ascale), + sel * asc_sel_stride
    long asc_sel_stride;
};

template <int ACT>
__global__ __launch_bounds__(RW_THREADS) void l_ln_act(LnActParams p) {
    const int row = blockIdx.x * (RW_THREADS / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= p.rows) return;
    const int sel = p.sel ? p.sel[(size_t)(row / p.rows_per_env) * p.sel_stride] : 0;
    const float *g = p.g + (size_t)sel * p.gb_sel_stride, *bb = p.b + (size_t)sel * p.gb_sel_stride;
    float *xr = p.x + (size_t)row * p.ld;
    const int n4 = p.width / 4;
    float s = 0.f;
    for (int q = lane; q < n4; q += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(xr + 4 * q);
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float mean = group_sum<64>(s) / (float)p.width;
    float ss = 0.f;
    for (int q = lane; q < n4; q += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(xr + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[e] - mean;
            ss += d * d;
        }
    }
    const float var = group_sum<64>(ss) / (float)p.width;
    const float rstd = 1.0f / sqrtf(var + LN_EPS);
    // SimNorm groups of 8 columns = the float4 of this lane and of lane ^ 1; every lane of a pair must take part in
    // the shuffles, so the loop bound is rounded up to a whole wave and out-of-range lanes carry -inf / 0.
    const int n4r = (n4 + 63) / 64 * 64;
    for (int q = lane; q < n4r; q += 64) {
        const bool ok = q < n4;
        f32x4 y;
        if (ok) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xr + 4 * q);
            const f32x4 gg = *reinterpret_cast<const f32x4 *>(g + 4 * q);
            const f32x4 be = *reinterpret_cast<const f32x4 *>(bb + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[e] - mean) * rstd * gg[e] + be[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = -INFINITY;
        }
        if (ACT == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = mish_f(y[e]);
        } else {
            float m = fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3]));
            m = fmaxf(m, __shfl_xor(m, 1));
            float es = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = ok ? expf(y[e] - m) : 0.f;
                es += y[e];
            }
            es += __shfl_xor(es, 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = y[e] / es;
        }
        if (ok) *reinterpret_cast<f32x4 *>(xr + 4 * q) = y;
    }
}

// two_hot_inv of one row of logits (tdmpc2/common/math.py:74-83); result in every lane.
__device__ __forceinline__ float twohot_wave(const float *lg, const float *bins, int num_bins, int lane) {
    if (num_bins <= 1) return num_bins == 0 ? lg[0] : symexp_f(lg[0]);  // regression head: identity / symexp (math.py:76-79)
    float v0 = lane < num_bins ? lg[lane] : -INFINITY;
    float v1 = lane + 64 < num_bins ? lg[lane + 64] : -INFINITY;
    const float m = group_max<64>(fmaxf(v0, v1));
    v0 = lane < num_bins ? expf(v0 - m) : 0.f;
    v1 = lane + 64 < num_bins ? expf(v1 - m) : 0.f;
    const float es = group_sum<64>(v0 + v1);
    float x = 0.f;
    if (lane < num_bins) x += (v0 / es) * bins[lane];
    if (lane + 64 < num_bins) x += (v1 / es) * bins[lane + 64];
    return symexp_f(group_sum<64>(x));
}

// Two-hot heads.  mode 0: reward of step t  -> G += disc[t] * (1 - term) * r          (tdmpc2/tdmpc2.py:128-130)
//                 mode 1: first Q head       -> qtmp = q
//                 mode 2: second Q head      -> value = G + disc[H] * (1 - term) * (qtmp + q) / 2   (tdmpc2.py:136)
struct TwoHotParams {
    const float *lg;
    int ld, rows, rows_per_env, num_bins, mode, t, H;
    const float *bins, *disc_pow;  // disc_pow [E, H+1]
    float *G, *qtmp, *value;       // [rows]
    const float *term;             // [rows] or null (non-episodic)
    float *trace;                  // optional [rows, trace_ld]: r_0..r_{H-1}, Q_a, Q_b, (a_H[A] written by l_pi_head)
    int trace_ld;
    int n_full, n_off;             // value[env * n_full + n_off + row % rows_per_env]: a row range of every plan (n_full = 0: value[row])
};

// what a two-hot head's value of one row goes into (lane 0 of the row's wavefront): shared by l_twohot and the head GEMM's
// two-hot epilogue (g_gemm_s<.., EPI = 3>)
__device__ __forceinline__ void twohot_apply(const TwoHotParams &p, int row, float r) {
    const float *disc = p.disc_pow + (size_t)(row / p.rows_per_env) * (p.H + 1);
    const float live = p.term ? 1.f - p.term[row] : 1.f;
    if (p.mode == 0) {
        const float g0 = p.t == 0 ? 0.f : p.G[row];
        p.G[row] = g0 + disc[p.t] * live * r;
        if (p.trace) p.trace[(size_t)row * p.trace_ld + p.t] = r;
    } else if (p.mode == 1) {
        p.qtmp[row] = r;
        if (p.trace) p.trace[(size_t)row * p.trace_ld + p.H] = r;
    } else {
        const size_t vi = p.n_full ? (size_t)(row / p.rows_per_env) * p.n_full + p.n_off + row % p.rows_per_env : (size_t)row;
        p.value[vi] = p.G[row] + disc[p.H] * live * ((p.qtmp[row] + r) / 2.f);
        if (p.trace) p.trace[(size_t)row * p.trace_ld + p.H + 1] = r;
    }
}

__global__ __launch_bounds__(RW_THREADS) void l_twohot(TwoHotParams p) {
    const int row = blockIdx.x * (RW_THREADS / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= p.rows) return;
    const float r = twohot_wave(p.lg + (size_t)row * p.ld, p.bins, p.num_bins, lane);
    if (lane != 0) return;
    twohot_apply(p, row, r);
}

// Termination head (tdmpc2/common/world_model.py:132-141, tdmpc2/tdmpc2.py:133-134):
// term <- clip(term + (sigmoid(logit) > 0.5), max = 1).  One thread per row.
__global__ void l_term(const float *lg, int ld, int rows, float *term) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const float pr = 1.f / (1.f + expf(-lg[(size_t)row * ld]));
    const float tnew = term[row] + (pr > 0.5f ? 1.f : 0.f);
    term[row] = fminf(tnew, 1.f);
}

// Policy head (world_model.py:152-174): action = tanh(mu + eps * exp(log_std)) -> X[row, L + a]; optionally also
// into actions[e, t, n, a] (policy-prior trajectories).  One thread per (row, a).
struct PiHeadParams {
    const float *lg;  // [rows, ld]: mu[0..A) | log_std[A..2A)
    int ld, rows, rows_per_env, nvalid /* rows per env that are real */, A, L, ldx;
    float lsmin, lsdif;
    const float *mask;      // [E, A] or null
    const float *eps;       // tape: eps[env * eps_estride + n * A + a], or null -> Philox
    long eps_estride;
    unsigned long long seed;
    unsigned int call;
    int site, iter;
    float *X;               // [rows, ldx]
    float *actions;         // [E, H, N, A] or null
    int t, H, N;
    float *trace;           // optional [rows, H+2+A]: a_H into columns H+2..
    const int *row_env;     // optional: the mask row of each sample row (training batches); eps / actions keep e = row / rows_per_env
    int n_off;              // sample index of the first row of every plan (row ranges: shard_values)
};

__global__ void l_pi_head(PiHeadParams p) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.rows * p.A) return;
    const int row = idx / p.A, a = idx % p.A;
    const int e = row / p.rows_per_env, n = row % p.rows_per_env;
    const float *lr = p.lg + (size_t)row * p.ld;
    float mu = lr[a];
    float ls = p.lsmin + 0.5f * p.lsdif * (tanhf(lr[p.A + a]) + 1.f);
    float eps = 0.f;
    if (n < p.nvalid) {
        const unsigned ridx = (unsigned)((size_t)(n + p.n_off) * p.A + a);
        eps = p.eps ? p.eps[(size_t)e * p.eps_estride + ridx] : rng_normal(p.seed, p.call, p.site, p.iter, e, ridx);
    }
    if (p.mask) {
        const float mk = p.mask[(size_t)(p.row_env ? p.row_env[row] : e) * p.A + a];
        mu *= mk;
        ls *= mk;
        eps *= mk;
    }
    const float act = tanhf(mu + eps * expf(ls));
    p.X[(size_t)row * p.ldx + p.L + a] = act;
    if (p.actions && n < p.nvalid) p.actions[(((size_t)e * p.H + p.t) * p.N + n) * p.A + a] = act;
    if (p.trace) p.trace[(size_t)row * (p.H + 2 + p.A) + p.H + 2 + a] = act;
}

// X[row, 0:L) <- z0[env]; X[row, L:ldx) <- 0; G, term <- 0.  One workgroup per row.
__global__ void l_init_x(float *X, int ldx, int L, int rows_per_env, const float *z0, float *G, float *term) {
    const int row = blockIdx.x;
    const float *z = z0 + (size_t)(row / rows_per_env) * L;
    float *xr = X + (size_t)row * ldx;
    for (int c = threadIdx.x; c < ldx; c += blockDim.x) xr[c] = c < L ? z[c] : 0.f;
    if (threadIdx.x == 0) {
        if (G) G[row] = 0.f;
        if (term) term[row] = 0.f;
    }
}

// Sampling of one CEM iteration (tdmpc2/tdmpc2.py:176-181) for all steps: rows n >= P of actions[E, H, N, A].
struct SampleParams {
    int E, H, N, A, P, iter;
    const float *mean, *std;  // [E, H, A]
    const float *mask;        // [E, A] or null
    const float *eps;         // tape slice of this iteration [.., H, N-P, A], env stride below; null -> Philox
    long eps_estride;
    unsigned long long seed;
    unsigned int call;
    float *actions;
    int *qidx;                // optional [E, 2]: this iteration's two Q heads drawn here as well (Philox; what l_qidx would launch for)
    int nq;
    unsigned int *arrive;     // optional: the stage's arrival counters [arrive_n], zeroed here (what lay_arrive_reset's memset would launch for)
    int arrive_n;
};

__global__ void l_sample(SampleParams p) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p.arrive_n; i += gridDim.x * blockDim.x) p.arrive[i] = 0u;
    if (p.qidx && blockIdx.x == 0) {  // two distinct heads, uniform over ordered pairs (l_qidx's draw)
        for (int e = threadIdx.x; e < p.E; e += blockDim.x) {
            const uint4 r = rng_raw(p.seed, p.call, SITE_QIDX, p.iter, e, 0);
            int q0 = (int)(r.x % (unsigned)p.nq), q1 = (int)(r.y % (unsigned)(p.nq - 1));
            if (q1 >= q0) ++q1;
            p.qidx[2 * e] = q0;
            p.qidx[2 * e + 1] = q1;
        }
    }
    // one work item per PAIR of action columns: the same Philox pairs (index, Box-Muller branches) as the fused family's
    // rollout kernels draw, so that a seed means the same plan on every kernel family (tdmpc2_plan_export_noise)
    const int hp = (p.A + 15) / 16 * 8, hpa = (p.A + 1) / 2;
    const size_t total = (size_t)p.E * p.H * (p.N - p.P) * hpa;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int a0 = 2 * (int)(idx % hpa);
        size_t r = idx / hpa;
        const int n = r % (p.N - p.P);
        r /= (p.N - p.P);
        const int t = r % p.H, e = r / p.H;
        float z[2] = {0.f, 0.f};
        if (p.eps) {
            const float *ep = p.eps + (size_t)e * p.eps_estride + (unsigned)(((size_t)t * (p.N - p.P) + n) * p.A + a0);
            z[0] = ep[0];
            if (a0 + 1 < p.A) z[1] = ep[1];
        } else {
            rng_normal2(p.seed, p.call, SITE_SAMPLE, p.iter, e, (unsigned)(((size_t)t * (p.N - p.P) + n) * hp + a0 / 2), z[0], z[1]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int a = a0 + u;
            if (a >= p.A) break;
            float v = sample_action(p.mean[((size_t)e * p.H + t) * p.A + a], p.std[((size_t)e * p.H + t) * p.A + a], z[u]);
            if (p.mask) v *= p.mask[(size_t)e * p.A + a];
            p.actions[(((size_t)e * p.H + t) * p.N + p.P + n) * p.A + a] = v;
        }
    }
}

// X[row, L + a] <- actions[e, t, n_off + n, a]; rows are (e, n) with n < nsub
__global__ void l_set_action(float *X, int ldx, int L, int A, int N, int H, int t, int rows, const float *actions, int nsub, int n_off) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * A) return;
    const int row = idx / A, a = idx % A;
    const int e = row / nsub, n = n_off + row % nsub;
    X[(size_t)row * ldx + L + a] = actions[(((size_t)e * H + t) * N + n) * A + a];
}

// Per-plan set-up of the layered path (grid E): effective first-layer biases b + W[:, L:L+T] . task_emb for every net
// (multitask), mean / std initialisation and warm start (tdmpc2/tdmpc2.py:164-167).
struct LSetupParams {
    int E, H, A, T, M, Mp, nnets, multitask;
    float max_std;
    const float *bias[4 + MAXQ];  // first-layer biases [Mp]
    const float *wemb[4 + MAXQ];  // [M, T] task-embedding columns of each first layer (null when the net is absent)
    const float *task_emb, *prev_mean;
    const unsigned char *t0;
    float *beff, *mean, *std;
};

// grid (E, nnets, column chunks): block (e, net, z) folds columns z, z + gridDim.z, ... x 256 of one net's task-embedding columns
// (one chunk per block when the grid is made that way: a single plan of the 317M model spent 240 us here with 13 blocks of 16 columns
// per thread, r6zw); block (e, 0, 0) also initialises mean / std.  One fmaf chain over the task dimension per column, whatever the grid.
__global__ void l_setup(LSetupParams p) {
    const int e = blockIdx.x, tid = threadIdx.x;
    if (p.multitask) {
        const float *emb = p.task_emb + (size_t)e * p.T;
        {
            const int net = blockIdx.y;
            if (p.wemb[net])
            for (int c = blockIdx.z * blockDim.x + tid; c < p.Mp; c += blockDim.x * gridDim.z) {
                float s = 0.f;
                if (c < p.M) {
                    const float *w = p.wemb[net] + (size_t)c * p.T;
                    for (int k = 0; k < p.T; ++k) s = fmaf(w[k], emb[k], s);
                }
                p.beff[((size_t)e * p.nnets + net) * p.Mp + c] = p.bias[net][c] + s;
            }
        }
    }
    if (p.mean && blockIdx.y == 0 && blockIdx.z == 0) {
        for (int idx = tid; idx < p.H * p.A; idx += blockDim.x) {
            const int t = idx / p.A;
            float m = 0.f;
            if (!p.t0[e] && t < p.H - 1) m = p.prev_mean[(size_t)e * p.H * p.A + idx + p.A];
            p.mean[(size_t)e * p.H * p.A + idx] = m;
            p.std[(size_t)e * p.H * p.A + idx] = p.max_std;
        }
    }
}

// qidx of this iteration from Philox when no tape is given: two distinct heads, uniform over ordered pairs.
__global__ void l_qidx(int E, int nq, int iter, unsigned long long seed, unsigned int call, int *qidx) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const uint4 r = rng_raw(seed, call, SITE_QIDX, iter, e, 0);
    int q0 = (int)(r.x % (unsigned)nq), q1 = (int)(r.y % (unsigned)(nq - 1));
    if (q1 >= q0) ++q1;
    qidx[2 * e] = q0;
    qidx[2 * e + 1] = q1;
}

// value[e, n] <- vrow[e * N + n]  is the identity layout; copy of the tape's qidx slice into the dense [E, 2] buffer.
__global__ void l_copy_qidx(int E, const int *src, long estride, int *dst) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    dst[2 * e] = src[(size_t)e * estride];
    dst[2 * e + 1] = src[(size_t)e * estride + 1];
}

// ---------------------------------------------------------------- pi + two Q heads on a batch of latent rows (layered family)
// X[row, 0:L) <- z[row] (rows >= nvalid: zeros); X[row, L:ldx) <- 0.  One workgroup per row.
__global__ void l_init_rows(float *X, int ldx, int L, const float *z, int nvalid) {
    const int row = blockIdx.x;
    float *xr = X + (size_t)row * ldx;
    for (int c = threadIdx.x; c < ldx; c += blockDim.x) xr[c] = (c < L && row < nvalid) ? z[(size_t)row * L + c] : 0.f;
}

// The value heads of TDMPC2._td_target / update_pi (tdmpc2/tdmpc2.py:208-254) on two-hot logits, one wavefront per row:
// mode 0: qtmp <- two_hot_inv(first head); mode 1: out <- min | mean of (qtmp, second head), then optionally
// reward + discount (1 - terminated) out with the row's task discount.
struct ValueHeadParams {
    const float *lg;
    int ld, rows, num_bins, mode, reduce_min;
    const float *bins;
    float *qtmp, *out;
    const float *reward, *terminated;  // [rows] or null
    float discount;
    const float *disc_tab;             // [n_tasks] or null
    const int *row_env;                // [rows] task of each row (with disc_tab)
    // the handle's host-mapped error word: a bounded inter-workgroup wait (fused NormedLinear epilogue) gave up somewhere in this
    // call -> the outputs are NaN, never finite garbage (tdmpc2_plan_take_fault; null: no such wait on this path)
    const unsigned int *err;
    float *action;                     // [rows, A] (policy_value's second output) or null: NaN-filled with `out` on a fault
    int A;
};
__global__ __launch_bounds__(RW_THREADS) void l_value_head(ValueHeadParams p) {
    const int row = blockIdx.x * (RW_THREADS / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= p.rows) return;
    const float q = twohot_wave(p.lg + (size_t)row * p.ld, p.bins, p.num_bins, lane);
    if (lane != 0) return;
    if (p.mode == 0) {
        p.qtmp[row] = q;
        return;
    }
    float v = p.reduce_min ? fminf(p.qtmp[row], q) : (p.qtmp[row] + q) / 2.f;
    if (p.reward) {
        const float disc = p.disc_tab ? p.disc_tab[p.row_env[row]] : p.discount;
        v = p.reward[row] + disc * (1.f - p.terminated[row]) * v;
    }
    if (p.err && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
        v = __builtin_nanf("");
        if (p.action)
            for (int a = 0; a < p.A; ++a) p.action[(size_t)row * p.A + a] = __builtin_nanf("");
    }
    p.out[row] = v;
}
