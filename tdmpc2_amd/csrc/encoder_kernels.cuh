// State-observation encoder (tdmpc2/common/world_model.py:103-112, layers.enc: tdmpc2/common/layers.py:153-164):
//   z = SimNorm(LN(W_n ... Mish(LN(W_1 [obs, task_emb] + b_1)) ... + b_n))
// one workgroup per environment, the activation row in LDS, every layer in the same launch.  This is 0.2 % of a plan's
// FLOPs; it is here so that a planning step is obs -> action inside the library (no framework launches in between),
// not for its arithmetic: plain fp32 FMAs, thread = output feature, weights stored transposed ([in][out]) at bind
// time so that a wave reads 256 contiguous bytes per k.
// Included by tdmpc2_plan.hip inside its anonymous namespace.
#pragma once

constexpr int ENC_MAX_LAYERS = 6;    // num_enc_layers is 2..5 in the reference's model table (common/__init__.py:1-24)
constexpr int ENC_THREADS = 512;
constexpr int ENC_MAX_PER_THREAD = 8;  // output features per thread: widths up to 4096 (the 317M model's enc_dim)

struct EncLayerDev {
    const float *wt;    // [in][out], transposed nn.Linear weight
    const float *bias;  // [out]
    const float *g, *b; // LayerNorm affine [out]
    int in, out;
};
struct EncodeParams {
    EncLayerDev l[ENC_MAX_LAYERS];
    int nl;
    int obs_dim, T, maxw, simnorm_dim;
    const float *obs;       // [E, obs_dim]
    const float *task_emb;  // [E, T] or null
    float *z;               // [E, latent_dim]
};

__device__ __forceinline__ float enc_block_sum(float v, float *red) {
    // all threads get the sum over the workgroup
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // `red` may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < ENC_THREADS / 64; ++w) s += red[w];
    return s;
}

__global__ __launch_bounds__(ENC_THREADS) void k_encode(EncodeParams p) {
    extern __shared__ float enc_lds[];
    float *xa = enc_lds, *xb = enc_lds + p.maxw, *red = enc_lds + 2 * p.maxw;
    const int e = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < p.obs_dim; i += ENC_THREADS) xa[i] = p.obs[(size_t)e * p.obs_dim + i];
    for (int i = tid; i < p.T; i += ENC_THREADS) xa[p.obs_dim + i] = p.task_emb[(size_t)e * p.T + i];
    __syncthreads();
    for (int l = 0; l < p.nl; ++l) {
        const EncLayerDev ly = p.l[l];
        float y[ENC_MAX_PER_THREAD];
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < ENC_MAX_PER_THREAD; ++u) {
            const int f = tid + u * ENC_THREADS;
            y[u] = 0.f;
            if (f < ly.out) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // four chains: the loads of different k are independent
                const float *w = ly.wt + f;
                int k = 0;
                for (; k + 4 <= ly.in; k += 4) {
                    a0 = fmaf(xa[k], w[(size_t)k * ly.out], a0);
                    a1 = fmaf(xa[k + 1], w[(size_t)(k + 1) * ly.out], a1);
                    a2 = fmaf(xa[k + 2], w[(size_t)(k + 2) * ly.out], a2);
                    a3 = fmaf(xa[k + 3], w[(size_t)(k + 3) * ly.out], a3);
                }
                for (; k < ly.in; ++k) a0 = fmaf(xa[k], w[(size_t)k * ly.out], a0);
                y[u] = ((a0 + a1) + (a2 + a3)) + ly.bias[f];
                part += y[u];
            }
        }
        const float mean = enc_block_sum(part, red) / (float)ly.out;
        part = 0.f;
#pragma unroll
        for (int u = 0; u < ENC_MAX_PER_THREAD; ++u) {
            const int f = tid + u * ENC_THREADS;
            if (f < ly.out) {
                const float d = y[u] - mean;
                part = fmaf(d, d, part);
            }
        }
        const float rstd = 1.0f / sqrtf(enc_block_sum(part, red) / (float)ly.out + LN_EPS);
        const bool last = l == p.nl - 1;
#pragma unroll
        for (int u = 0; u < ENC_MAX_PER_THREAD; ++u) {
            const int f = tid + u * ENC_THREADS;
            const bool on = f < ly.out;
            float v = on ? (y[u] - mean) * rstd * ly.g[f] + ly.b[f] : -INFINITY;
            if (!last) {
                if (on) {  // Mish (layers.py:103), libm-accurate like the exact arithmetic of the planner
                    const float ex = expf(fminf(v, 20.f));
                    const float n = ex * (ex + 2.f);
                    xb[f] = v * (n / (n + 2.f));
                }
            } else {
                // SimNorm (layers.py:74-91): softmax over groups of simnorm_dim (8) consecutive features = 8 adjacent lanes
                float mx = v;
                for (int o = 1; o < p.simnorm_dim; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                const float ex = on ? expf(v - mx) : 0.f;
                float s = ex;
                for (int o = 1; o < p.simnorm_dim; o <<= 1) s += __shfl_xor(s, o);
                if (on) p.z[(size_t)e * ly.out + f] = ex / s;
            }
        }
        __syncthreads();
        float *t = xa; xa = xb; xb = t;
    }
}

// ---- wide encoders (a layer beyond ENC_WIDE columns: the 19M / 48M / 317M models): one workgroup per environment would
// stream up to 225 MB of weights through a single CU (measured 9.9 ms for the 317M encoder), so each layer becomes two
// launches: k_enc_gemv (grid: out / 64 x E; the four waves of a workgroup split the contraction, lane = output feature,
// 256 contiguous bytes of the transposed weights per wave and k) and k_enc_norm (grid E: LayerNorm + Mish / SimNorm of
// the row).
constexpr int ENC_WIDE = 1024;

struct EncGemvParams {
    const float *wt, *bias;  // [in][out], [out]
    const float *x;          // [E, ldx] activations (layer 0: obs | task_emb assembled by the caller kernel)
    const float *obs, *emb;  // layer 0 only: obs [E, obs_dim], emb [E, T] (x = null)
    int obs_dim, T;
    int in, out, ldx;
    float *y;                // [E, out] pre-activations
};

__global__ __launch_bounds__(256) void k_enc_gemv(EncGemvParams p) {
    extern __shared__ float enc_lds[];  // x row [in] + partials [4][64]
    float *xs = enc_lds, *part = enc_lds + p.in;
    const int e = blockIdx.y, f0 = blockIdx.x * 64, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p.x) {
        for (int i = tid; i < p.in; i += 256) xs[i] = p.x[(size_t)e * p.ldx + i];
    } else {
        for (int i = tid; i < p.obs_dim; i += 256) xs[i] = p.obs[(size_t)e * p.obs_dim + i];
        for (int i = tid; i < p.T; i += 256) xs[p.obs_dim + i] = p.emb[(size_t)e * p.T + i];
    }
    __syncthreads();
    const int f = f0 + lane;
    const int kq = (p.in + 3) / 4, k0 = wave * kq, k1 = min(p.in, k0 + kq);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (f < p.out) {
        const float *w = p.wt + f;
        int k = k0;
        for (; k + 4 <= k1; k += 4) {
            a0 = fmaf(xs[k], w[(size_t)k * p.out], a0);
            a1 = fmaf(xs[k + 1], w[(size_t)(k + 1) * p.out], a1);
            a2 = fmaf(xs[k + 2], w[(size_t)(k + 2) * p.out], a2);
            a3 = fmaf(xs[k + 3], w[(size_t)(k + 3) * p.out], a3);
        }
        for (; k < k1; ++k) a0 = fmaf(xs[k], w[(size_t)k * p.out], a0);
    }
    part[wave * 64 + lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0 && f < p.out)
        p.y[(size_t)e * p.out + f] = ((part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane])) + p.bias[f];
}

struct EncNormParams {
    const float *y;      // [E, width] pre-activations
    const float *g, *b;  // [width]
    float *out;          // [E, width]: Mish(LN(y)) (hidden layers) or SimNorm(LN(y)) (last layer = z)
    int width, last, simnorm_dim;
};

__global__ __launch_bounds__(ENC_THREADS) void k_enc_norm(EncNormParams p) {
    __shared__ float red[ENC_THREADS / 64];
    const int e = blockIdx.x, tid = threadIdx.x;
    float y[ENC_MAX_PER_THREAD];
    float part = 0.f;
#pragma unroll
    for (int u = 0; u < ENC_MAX_PER_THREAD; ++u) {
        const int f = tid + u * ENC_THREADS;
        y[u] = f < p.width ? p.y[(size_t)e * p.width + f] : 0.f;
        part += y[u];
    }
    const float mean = enc_block_sum(part, red) / (float)p.width;
    part = 0.f;
#pragma unroll
    for (int u = 0; u < ENC_MAX_PER_THREAD; ++u) {
        if (tid + u * ENC_THREADS < p.width) {
            const float d = y[u] - mean;
            part = fmaf(d, d, part);
        }
    }
    const float rstd = 1.0f / sqrtf(enc_block_sum(part, red) / (float)p.width + LN_EPS);
#pragma unroll
    for (int u = 0; u < ENC_MAX_PER_THREAD; ++u) {
        const int f = tid + u * ENC_THREADS;
        const bool on = f < p.width;
        const float v = on ? (y[u] - mean) * rstd * p.g[f] + p.b[f] : -INFINITY;
        if (!p.last) {
            if (on) {
                const float ex = expf(fminf(v, 20.f));
                const float n = ex * (ex + 2.f);
                p.out[(size_t)e * p.width + f] = v * (n / (n + 2.f));
            }
        } else {
            float mx = v;
            for (int o = 1; o < p.simnorm_dim; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            const float ex = on ? expf(v - mx) : 0.f;
            float s = ex;
            for (int o = 1; o < p.simnorm_dim; o <<= 1) s += __shfl_xor(s, o);
            if (on) p.out[(size_t)e * p.width + f] = ex / s;
        }
    }
}

// nn.Linear weight [out][in] -> [in][out]
__global__ void k_transpose(const float *__restrict__ w, float *__restrict__ wt, int out, int in) {
    const size_t n = (size_t)out * in;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / out), f = (int)(i % out);
        wt[i] = w[(size_t)f * in + k];
    }
}
