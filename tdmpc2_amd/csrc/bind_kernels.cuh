// Bind-time kernels (weight scaling / packing, per-task bias tables): launched by the host side of tdmpc2_plan.hip only.
// Included by tdmpc2_plan.hip inside its anonymous namespace.
#pragma once

// beff_tab[task][net][WIDTH] = b + W[:, L:L+T] . task_emb[task] for the policy and the Q heads (online or target):
// the per-task effective first-layer biases ks_value indexes per row.  grid = n_tasks, block = WIDTH threads.
__global__ void ks_task_bias(TaskBiasParams p) {
    const int task = blockIdx.x, f = threadIdx.x;
    const float *emb = p.task_emb + (size_t)task * p.T;
    for (int net = 0; net < p.nnets; ++net) {
        if (!p.wemb[net]) continue;
        const float *w = p.wemb[net] + (size_t)f * p.T;
        float sacc = 0.f;
        for (int k = 0; k < p.T; ++k) sacc = fmaf(w[k], emb[k], sacc);
        p.beff_tab[((size_t)task * p.nnets + net) * WIDTH + f] = p.bias[net][f] + sacc;
    }
}

// ================================================================ weight scaling + packing
// max |W| of one matrix -> bits (atomicMax on the uint pattern of a non-negative float is order preserving)
__global__ void k_absmax(const float *W, size_t n, unsigned int *out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = fabsf(W[i]);
        if (a == a && a < INFINITY) m = fmaxf(m, a);
    }
    m = group_max<64>(m);
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}
// kw such that max|W| 2^kw in [2^13, 2^14); wscale = 2^kw (for packing)
__global__ void k_wscale(LayerScal *s) {
    const float m = __uint_as_float(s->maxbits);
    int ex = 0;
    if (m > 0.f) frexpf(m, &ex);  // m = f 2^ex, f in [0.5, 1)
    int kw = 14 - ex;
    kw = kw > 40 ? 40 : (kw < -40 ? -40 : kw);
    s->kw = kw;
    s->wscale = ldexpf(1.f, kw);
}
// Output scale of a LayerNorm + Mish layer of `width` features: |LayerNorm(x)_i| <= sqrt(width - 1) for any x, so
// |Mish(g x + b)| <= sqrt(width - 1) max|g| + max|b| =: B.  ka = the largest exponent <= 5 with B 2^ka < 2^15 (half of
// the f16 maximum: rounding of the hi piece cannot reach Inf).  Trained checkpoints (g ~ 1) keep ka = 5.
__global__ void k_ascale(LayerScal *s, int width, int has_ln) {
    int ka = ACT_SCALE_LOG2;
    if (has_ln) {
        const float bound = sqrtf((float)(width > 1 ? width - 1 : 1)) * __uint_as_float(s->gmax) + __uint_as_float(s->bmax);
        if (bound > 0.f) {
            int ex = 0;
            frexpf(bound, &ex);  // bound < 2^ex
            ka = 15 - ex < ka ? 15 - ex : ka;
        }
        ka = ka < -24 ? -24 : ka;
    }
    s->ka = ka;
    s->ascale = ldexpf(1.f, ka);
}
// oscale of the three layers of one net: layer 0 reads [z | a] (scale 2^5), layer l > 0 reads layer l - 1's output
__global__ void k_net_scales(LayerScal *s3) {
    for (int l = 0; l < 3; ++l) {
        const int kin = l == 0 ? ACT_SCALE_LOG2 : s3[l - 1].ka;
        s3[l].oscale = ldexpf(1.f, -(s3[l].kw + kin));
    }
}
// dst[ct][kb][plane][lane][e]: W[row = ct*32 + (lane & 31)][k = kb*16 + 8 (lane >> 5) + e] * wscale, hi / lo pieces;
// packed k axis [z columns (nz) | action columns (na, zero padded)], source columns [z | task_emb (nt) | action].
__global__ void k_pack_split(const float *W, int out, int in, int nz, int nt, int na, int CT, int KB, const float *wscale,
                             _Float16 *dst) {
    const size_t total = (size_t)CT * KB * 512;  // (lane, e) pairs per (ct, kb)
    const float sc = *wscale;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63;
        const size_t blk = idx >> 9;
        const int kb = blk % KB, ct = blk / KB;
        const int row = ct * 32 + (lane & 31);
        const int k = kb * 16 + 8 * (lane >> 5) + e;
        float v = 0.f;
        if (row < out) {
            int src = -1;
            if (k < nz) src = k;
            else if (k - nz < na) src = nz + nt + (k - nz);
            if (src >= 0 && src < in) v = W[(size_t)row * in + src] * sc;
        }
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        _Float16 *base = dst + blk * 1024;  // 2 planes x 64 lanes x 8
        base[lane * 8 + e] = h;
        base[512 + lane * 8 + e] = l;
    }
}

