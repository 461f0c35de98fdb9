// Fused 512-wide planner kernels on the f16 matrix pipe with fp32-class accuracy ("f16x2 split").
//
// Every fp32 operand x is carried as two f16 pieces, x ~ hi + lo with hi = f16(x), lo = f16(x - hi): 22 significand
// bits.  A product a.b is formed from THREE v_mfma_f32_32x32x16_f16 instructions
//         a_hi.b_hi + a_hi.b_lo + a_lo.b_hi                       (the dropped a_lo.b_lo term is ~2^-22 relative)
// accumulated in fp32.  The matrix pipe issues f16 MFMAs at 16x the rate of v_mfma_f32_32x32x2_f32, so the contraction
// runs at up to 16/3 of the exact-fp32 kernels' roof while the error against an fp64 evaluation of the same network
// stays in the fp32 round-off class (measured: tests/test_gpu_split.py; CPU emulation in DESIGN.md).
// Operands are pre-scaled by powers of two so that the lo pieces stay in the f16 normal range: activations by 2^5,
// each weight matrix by 2^kw with max|W| 2^kw in [2^13, 2^14) (k_wscale); the fp32 accumulator is multiplied back by
// the exact 2^-(kw+5) in the epilogue.
//
// Same phase structure, work savings and reference citations as k_rollout / k_pitraj / k_setup in tdmpc2_plan.hip;
// what differs is the LDS tile (operand form: per row [hi plane | lo plane], f16) and a leaner epilogue: the
// pre-activation tile is staged once in fp32 (row-local alias of the same bytes), each thread then holds its 64-column
// row slice in registers for LayerNorm statistics, activation and the hi/lo split, and writes the operand form back.
// Included by tdmpc2_plan.hip inside its anonymous namespace.
#pragma once

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr float ACT_SCALE = 32.f;  // 2^5
constexpr int ACT_SCALE_LOG2 = 5;
constexpr int ZKB16 = WIDTH / 16;  // k16-blocks covering the latent columns

struct LayerS {
    const _Float16 *wp;  // packed [CT][KB][2 planes][64 lanes][8]
    const float *bias;   // [CT*32] zero padded
    const float *g, *b;  // LayerNorm affine (null for plain output layers)
    const float *oscale; // device scalar: 2^-(kw + ACT_SCALE_LOG2)
    int KB;              // k16-blocks
    int CT;
};
struct NetS {
    LayerS l[3];
};

// The action padding (16, 32, 48 or 64 columns) is a template parameter: with compile-time row strides every LDS
// address of the unrolled epilogues is base + immediate; with a run-time stride the compiler pre-computes hundreds of
// addresses, hoists them out of the step loop and spills them.
template <int APAD>
struct CtxT {
    static constexpr int SH = WIDTH + APAD;  // plane length in halfs
    static constexpr int RSH = 2 * SH + 8;   // row stride in halfs (row stride in dwords = SH + 4 = 4 x odd)
    _Float16 *act;  // LDS tile, operand form: row r at act + r * RSH: [hi: SH halfs | lo: SH halfs | 8 pad]
    float *stats;   // LDS [8 waves][64 rows][2]: per-wave LayerNorm partials
    int tid, wave, lane;
    __device__ __forceinline__ float *f32() const { return reinterpret_cast<float *>(act); }  // staging view [64][RSF]
    static constexpr __device__ __forceinline__ int RSF() { return RSH / 2; }
};

__device__ __forceinline__ float mish_fast(float x) {
    // x * tanh(softplus(x)) = x * n / (n + 2), n = e^x (e^x + 2); for x > 20 the ratio rounds to 1 in fp32, so the
    // exponent is clamped instead of branching (tdmpc2/common/layers.py:103)
    const float e = __expf(fminf(x, 20.f));
    const float n = e * (e + 2.f);
    return x * (n * __builtin_amdgcn_rcpf(n + 2.f));  // v_rcp_f32: 1 ulp; __fdividef expands to the full division
}

__device__ __forceinline__ void split4(const f32x4 y, f16x4 &hi, f16x4 &lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ys = y[e] * ACT_SCALE;
        const _Float16 h = (_Float16)ys;
        hi[e] = h;
        lo[e] = (_Float16)(ys - (float)h);
    }
}

// ---------------------------------------------------------------- contraction loops
// v_mfma_f32_32x32x16_f16: lane l supplies A[i = l & 31][k = 8 (l >> 5) .. +7] and B[k = 8 (l >> 5) .. +7][j = l & 31].
// Wave w of 8 owns output columns [64 w, 64 w + 64) for both 32-row tiles: acc[set][row tile][col tile].
// B fragments are prefetched PF k-blocks ahead (PF x 16 VGPRs): at the f16 matrix rate one k-block is only
// 12 MFMAs = 384 pipe cycles per wave, far shorter than an L2 round trip under load.
#ifndef SPLIT_PF
#define SPLIT_PF 4
#endif
constexpr int PF = SPLIT_PF;
struct BFrag {
    f16x8 h[2], l[2];
};
// B-fragment addressing: wave-uniform byte pointers (SGPR pairs, advanced by scalar arithmetic) + ONE 32-bit lane
// offset, made opaque so that the compiler cannot fully unroll the k-loop into per-k-block 64-bit VGPR addresses and
// hoist them out of the step loop (that cost ~1000 spilled registers).
// SPLIT_ABL_* macros exist for ablation timing builds only (tools/ablate.sh); the shipped library defines none.
__device__ __forceinline__ f16x8 ldw(const char *ubase, unsigned voff, int imm) {
#ifdef SPLIT_ABL_NO_BLOAD
    f16x8 r;
    const _Float16 v = (_Float16)(float)(voff + imm);
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = v;
    return r;
#else
    return *reinterpret_cast<const f16x8 *>(ubase + voff + imm);
#endif
}
#ifdef SPLIT_ABL_NO_MFMA
__device__ __forceinline__ f32x16 fake_mfma(f16x8 a, f16x8 b, f32x16 c) {
    c[0] += (float)a[0] * (float)b[0];  // keeps the dependence on the operand loads, costs one VALU op
    return c;
}
#define SPLIT_MFMA(a, b, c) fake_mfma(a, b, c)
#else
#define SPLIT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif
__device__ __forceinline__ void load_b(BFrag &b, const char *u0, const char *u1, unsigned voff) {
    b.h[0] = ldw(u0, voff, 0);
    b.l[0] = ldw(u0, voff, 1024);
    b.h[1] = ldw(u1, voff, 0);
    b.l[1] = ldw(u1, voff, 1024);
}
template <class CT>
__device__ __forceinline__ void kloop_s(const CT &c, const LayerS &ly, int kb0, int kb1, f32x16 (&acc)[2][2]) {
    const int i = c.lane & 31, hh = c.lane >> 5;
    const _Float16 *a0p = c.act + i * c.RSH + 8 * hh + kb0 * 16;
    const _Float16 *a1p = a0p + 32 * c.RSH;
    // one k-block of one column tile = 2 planes x 64 lanes x 16 B = 2048 B
    const char *u0 = reinterpret_cast<const char *>(ly.wp) + ((size_t)(2 * c.wave) * ly.KB + kb0) * 2048;
    const char *u1 = u0 + (size_t)ly.KB * 2048;
    unsigned voff = (unsigned)c.lane * 16u;
    asm volatile("" : "+v"(voff));
    const int nk = kb1 - kb0;
    BFrag ring[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int kd = d < nk ? d : nk - 1;
        load_b(ring[d], u0 + (size_t)kd * 2048, u1 + (size_t)kd * 2048, voff);
    }
#pragma unroll 1
    for (int k = 0; k < nk; k += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int kk = k + d;
            if (kk < nk) {  // wave-uniform
                const BFrag b = ring[d];
                const int kn = kk + PF < nk ? kk + PF : nk - 1;
                load_b(ring[d], u0 + (size_t)kn * 2048, u1 + (size_t)kn * 2048, voff);
                const f16x8 ah0 = *reinterpret_cast<const f16x8 *>(a0p + kk * 16);
                const f16x8 al0 = *reinterpret_cast<const f16x8 *>(a0p + c.SH + kk * 16);
                const f16x8 ah1 = *reinterpret_cast<const f16x8 *>(a1p + kk * 16);
                const f16x8 al1 = *reinterpret_cast<const f16x8 *>(a1p + c.SH + kk * 16);
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    acc[0][cc] = SPLIT_MFMA(ah0, b.h[cc], acc[0][cc]);
                    acc[1][cc] = SPLIT_MFMA(ah1, b.h[cc], acc[1][cc]);
                }
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    acc[0][cc] = SPLIT_MFMA(ah0, b.l[cc], acc[0][cc]);
                    acc[1][cc] = SPLIT_MFMA(ah1, b.l[cc], acc[1][cc]);
                }
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    acc[0][cc] = SPLIT_MFMA(al0, b.h[cc], acc[0][cc]);
                    acc[1][cc] = SPLIT_MFMA(al1, b.h[cc], acc[1][cc]);
                }
            }
        }
    }
}

// One 32x32 output tile (row tile rt, column tile ct): the narrow output layers.  Three independent accumulators
// (one per product kind) keep the matrix pipe issuing back to back; they are summed at the end.
template <class CT>
__device__ __forceinline__ void kloop_tile_s(const CT &c, const LayerS &ly, int ct, int rt, int kb0, int kb1, f32x16 &acc) {
#ifndef SPLIT_PFT
#define SPLIT_PFT 8
#endif
    constexpr int PFT = SPLIT_PFT;  // only 3 MFMAs (96 pipe cycles) per k-block here: prefetch deeper
    const int i = c.lane & 31, hh = c.lane >> 5;
    const _Float16 *ap = c.act + (rt * 32 + i) * c.RSH + 8 * hh + kb0 * 16;
    const char *u = reinterpret_cast<const char *>(ly.wp) + ((size_t)ct * ly.KB + kb0) * 2048;  // uniform
    unsigned voff = (unsigned)c.lane * 16u;
    asm volatile("" : "+v"(voff));
    const int nk = kb1 - kb0;
    f32x16 a1, a2;
#pragma unroll
    for (int e = 0; e < 16; ++e) a1[e] = a2[e] = 0.f;
    f16x8 rh[PFT], rl[PFT];
#pragma unroll
    for (int d = 0; d < PFT; ++d) {
        const int kd = d < nk ? d : nk - 1;
        rh[d] = ldw(u + (size_t)kd * 2048, voff, 0);
        rl[d] = ldw(u + (size_t)kd * 2048, voff, 1024);
    }
#pragma unroll 1
    for (int k = 0; k < nk; k += PFT) {
#pragma unroll
        for (int d = 0; d < PFT; ++d) {
            const int kk = k + d;
            if (kk < nk) {
                const f16x8 bh = rh[d], bl = rl[d];
                const int kn = kk + PFT < nk ? kk + PFT : nk - 1;
                rh[d] = ldw(u + (size_t)kn * 2048, voff, 0);
                rl[d] = ldw(u + (size_t)kn * 2048, voff, 1024);
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(ap + kk * 16);
                const f16x8 al = *reinterpret_cast<const f16x8 *>(ap + c.SH + kk * 16);
                acc = SPLIT_MFMA(ah, bh, acc);
                a1 = SPLIT_MFMA(ah, bl, a1);
                a2 = SPLIT_MFMA(al, bh, a2);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] += a1[e] + a2[e];
}

// acc * oscale + bias -> fp32 staging view.  C/D fragment: lane holds column (l & 31), rows (reg&3) + 8 (reg>>2) + 4 (l>>5).
template <class CT>
__device__ __forceinline__ void store_full_s(const CT &c, const f32x16 (&acc)[2][2], float osc, const float *bias) {
    float *f = c.f32();
    const int RSF = c.RSF();
    const int j = c.lane & 31, hh = c.lane >> 5;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int col = (2 * c.wave + cc) * 32 + j;
        const float bv = bias[col];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                f[row * RSF + col] = fmaf(acc[rt][cc][reg], osc, bv);
            }
    }
}
// the same into a dense global tile [64][WIDTH] (pre-activation parked in L2 instead of 64 held accumulators)
template <class CT>
__device__ __forceinline__ void store_full_global(const CT &c, const f32x16 (&acc)[2][2], float osc, const float *bias, float *dst) {
    const int j = c.lane & 31, hh = c.lane >> 5;
    // one 32-bit lane offset against the wave-uniform tile base; made opaque so that the 64 store addresses are formed
    // here (base + constant) instead of being hoisted out of the step loop as 64 live 64-bit pointers (spills)
    unsigned lane_off = (unsigned)(4 * hh * WIDTH + 2 * c.wave * 32 + j);
    asm volatile("" : "+v"(lane_off));
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const float bv = bias[(2 * c.wave + cc) * 32 + j];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const unsigned koff = (unsigned)((rt * 32 + (reg & 3) + 8 * (reg >> 2)) * WIDTH + cc * 32);
                dst[lane_off + koff] = fmaf(acc[rt][cc][reg], osc, bv);
            }
    }
}
template <class CT>
__device__ __forceinline__ void store_tile_s(const CT &c, const f32x16 &acc, float osc, const float *bias, int ct, int rt) {
    float *f = c.f32();
    const int RSF = c.RSF();
    const int j = c.lane & 31, hh = c.lane >> 5;
    const int col = ct * 32 + j;
    const float bv = bias[col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        f[row * RSF + col] = fmaf(acc[reg], osc, bv);
    }
}

// ---------------------------------------------------------------- row epilogue: staging (fp32) -> operand form (hi/lo f16)
// Lane = row, wave w = columns [64 w, 64 w + 64): the LayerNorm affine parameters and every column offset are
// wave-uniform (scalar loads, immediates), SimNorm groups of 8 are thread-local, and no cross-lane shuffles are needed.
// Row statistics are combined across the 8 waves through a small LDS exchange with Chan's parallel-variance formula
// (per-wave mean and sum of squared deviations: as robust as the two-pass form).  The barrier of that exchange also
// separates every wave's reads of the staging view from the operand-form writes that alias it.
// LayerNorm: biased variance, eps 1e-5 (layers.py:101).  ACT 0 Mish, 1 SimNorm(8) (layers.py:84-88).
template <int ACT, class CT>
__device__ __forceinline__ void ln_rows_s(const CT &c, const float *g, const float *b, float *gcopy /* optional [64][WIDTH] fp32 */,
                                          const float *gsrc = nullptr /* pre-activation tile in global instead of the staging view */) {
    const int row = c.lane, col0 = 64 * c.wave;
    const float *rp = gsrc ? gsrc + row * WIDTH + col0 : c.f32() + row * c.RSF() + col0;
    f32x4 v[16];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        v[q] = *reinterpret_cast<const f32x4 *>(rp + 4 * q);
        s += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
    }
    const float mw = s * (1.0f / 64.f);
    float m2 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[q][e] - mw;
            m2 = fmaf(d, d, m2);
        }
    float *stats = c.stats;  // [8 waves][64 rows][2]
    stats[(c.wave * 64 + row) * 2 + 0] = mw;
    stats[(c.wave * 64 + row) * 2 + 1] = m2;
    __syncthreads();
    float mean = 0.f, msum = 0.f;
    float pm[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        pm[w] = stats[(w * 64 + row) * 2 + 0];
        msum += stats[(w * 64 + row) * 2 + 1];
        mean += pm[w];
    }
    mean *= 0.125f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const float d = pm[w] - mean;
        msum = fmaf(64.f * d, d, msum);
    }
    const float rstd = 1.0f / sqrtf(msum * (1.0f / WIDTH) + LN_EPS);
    const float shift = -mean * rstd;
    _Float16 *hp = c.act + row * c.RSH + col0;
    const float *gw = g + col0, *bw = b + col0;  // wave-uniform
#ifdef SPLIT_ABL_NO_EPI
    if (mean == 12345.f)  // never true: the activation math below is skipped
#endif
#pragma unroll
    for (int q2 = 0; q2 < 8; ++q2) {  // two float4 chunks = one SimNorm group of 8 columns
        f32x4 y[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = 2 * q2 + u;
            const f32x4 gg = *reinterpret_cast<const f32x4 *>(gw + 4 * q);
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(bw + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[u][e] = fmaf(fmaf(v[q][e], rstd, shift), gg[e], bb[e]);
        }
        if (ACT == 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) y[u][e] = mish_fast(y[u][e]);
        } else {
            float m = fmaxf(fmaxf(fmaxf(y[0][0], y[0][1]), fmaxf(y[0][2], y[0][3])),
                            fmaxf(fmaxf(y[1][0], y[1][1]), fmaxf(y[1][2], y[1][3])));
            float es = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[u][e] = __expf(y[u][e] - m);
                    es += y[u][e];
                }
            const float inv = 1.0f / es;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) y[u][e] *= inv;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = 2 * q2 + u;
            f16x4 hi, lo;
            split4(y[u], hi, lo);
            *reinterpret_cast<f16x4 *>(hp + 4 * q) = hi;
            *reinterpret_cast<f16x4 *>(hp + c.SH + 4 * q) = lo;
            if (gcopy) *reinterpret_cast<f32x4 *>(gcopy + row * WIDTH + col0 + 4 * q) = y[u];
        }
    }
}

// two_hot_inv (math.py:74-83) on fp32 logits in the staging view.
template <class CT>
__device__ __forceinline__ float twohot_rows_s(const CT &c, const float *bins, int num_bins) {
    return twohot_rows(c.f32(), c.RSF(), bins, num_bins, c.tid);
}

template <int ACT, class CT>
__device__ __forceinline__ void layer_full_s(const CT &c, const LayerS &ly, const float *bias, int kb0, int kb1,
                                             float *gcopy = nullptr) {
    f32x16 acc[2][2];
    zero4(acc);
    kloop_s(c, ly, kb0, kb1, acc);
    const float osc = *ly.oscale;
    __syncthreads();
    store_full_s(c, acc, osc, bias);
    __syncthreads();
    ln_rows_s<ACT>(c, ly.g, ly.b, gcopy);
    __syncthreads();
}

template <class CT>
__device__ __forceinline__ float head_twohot_s(const CT &c, const LayerS &ly, const float *bins, int num_bins) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int rt = c.wave & 1, ct = c.wave >> 1;
    if (ct < ly.CT) kloop_tile_s(c, ly, ct, rt, 0, ZKB16, acc);
    const float osc = *ly.oscale;
    __syncthreads();
    if (ct < ly.CT) store_tile_s(c, acc, osc, ly.bias, ct, rt);
    __syncthreads();
    const float r = twohot_rows_s(c, bins, num_bins);
    __syncthreads();
    return r;
}

// write one action value into the operand-form action columns of a row
template <class CT>
__device__ __forceinline__ void put_action(const CT &c, int row, int a, float v) {
    const float vs = v * ACT_SCALE;
    const _Float16 h = (_Float16)vs;
    _Float16 *rp = c.act + row * c.RSH + WIDTH + a;
    rp[0] = h;
    rp[c.SH] = (_Float16)(vs - (float)h);
}

// Policy prior output layer + squashed Gaussian sample (world_model.py:152-173); action -> operand-form action columns
// (zero for padded columns) and optionally gdst[row * A + a] for rows < nvalid.
template <class CT, typename EpsFn>
__device__ __forceinline__ void head_pi_s(const CT &c, const LayerS &ly, int A, int Apad, float lsmin, float lsdif,
                                          const float *mask, EpsFn eps, float *gdst, int nvalid, float *tsc) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int rt = c.wave & 1, ct = c.wave >> 1;
    if (ct < ly.CT) kloop_tile_s(c, ly, ct, rt, 0, ZKB16, acc);
    const float osc = *ly.oscale;
    __syncthreads();
    if (ct < ly.CT) store_tile_s(c, acc, osc, ly.bias, ct, rt);
    __syncthreads();
    const int row = c.tid >> 3, part = c.tid & 7;
    const float *rp = c.f32() + row * c.RSF();
    // the logits (staging columns < 2A <= 128) do not alias the action columns (hi: floats 256.., lo: floats >= 512)
    for (int a = part; a < Apad; a += 8) {
        float out = 0.f;
        if (a < A) {
            float mu = rp[a];
            float ls = lsmin + 0.5f * lsdif * (tanhf(rp[A + a]) + 1.f);  // math.log_std, math.py:12-13
            float e = eps(row, a);
            if (mask) {
                const float mk = mask[a];
                mu *= mk;
                ls *= mk;
                e *= mk;
            }
            out = tanhf(mu + e * expf(ls));
            if (gdst && row < nvalid) gdst[row * A + a] = out;
            if (tsc) tsc[a] = out;
        }
        put_action(c, row, a, out);
    }
    __syncthreads();
}

// global fp32 [64][WIDTH] -> operand-form z columns
template <class CT>
__device__ __forceinline__ void tile_from_global_s(const CT &c, const float *src) {
    for (int idx = c.tid; idx < ROWS * (WIDTH / 4); idx += NTHREADS) {
        const int row = idx / (WIDTH / 4), c4 = idx % (WIDTH / 4);
        const f32x4 y = *reinterpret_cast<const f32x4 *>(src + row * WIDTH + 4 * c4);
        f16x4 hi, lo;
        split4(y, hi, lo);
        _Float16 *hp = c.act + row * c.RSH + 4 * c4;
        *reinterpret_cast<f16x4 *>(hp) = hi;
        *reinterpret_cast<f16x4 *>(hp + c.SH) = lo;
    }
}
template <class CT>
__device__ __forceinline__ void tile_broadcast_row_s(const CT &c, const float *src_row) {
    for (int idx = c.tid; idx < ROWS * (WIDTH / 4); idx += NTHREADS) {
        const int row = idx / (WIDTH / 4), c4 = idx % (WIDTH / 4);
        const f32x4 y = *reinterpret_cast<const f32x4 *>(src_row + 4 * c4);
        f16x4 hi, lo;
        split4(y, hi, lo);
        _Float16 *hp = c.act + row * c.RSH + 4 * c4;
        *reinterpret_cast<f16x4 *>(hp) = hi;
        *reinterpret_cast<f16x4 *>(hp + c.SH) = lo;
    }
}
// operand form -> fp32 trace dump (hi + lo, unscaled)
template <class CT>
__device__ __forceinline__ void dump_tile_s(const CT &c, float *trace, int nslot, int slot) {
    if (!trace) return;
    float *dst = trace + ((size_t)blockIdx.x * nslot + slot) * ROWS * WIDTH;
    for (int idx = c.tid; idx < ROWS * WIDTH; idx += NTHREADS) {
        const int row = idx / WIDTH, col = idx % WIDTH;
        const _Float16 *hp = c.act + row * c.RSH + col;
        dst[idx] = ((float)hp[0] + (float)hp[c.SH]) * (1.0f / ACT_SCALE);
    }
}

// ================================================================ kernel: per-plan setup (cf. k_setup)
template <int APAD>
__global__ __launch_bounds__(NTHREADS, 2) void ks_setup(SetupParamsT<NetS> p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int e = blockIdx.x, tid = threadIdx.x;
    CtxT<APAD> c{reinterpret_cast<_Float16 *>(smem), smem + ROWS * CtxT<APAD>::RSH / 2, tid,
                 __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63};
    if (p.multitask) {
        const float *emb = p.task_emb + (size_t)e * p.T;
        for (int net = 0; net < p.nnets; ++net) {
            const LayerS &l1 = net == BE_DYN ? p.dyn.l[0] : net == BE_REW ? p.rew.l[0] : net == BE_PI ? p.pi.l[0]
                                                                                                    : p.q[net - BE_Q0].l[0];
            const float *w = p.wemb[net] + (size_t)tid * p.T;
            float s = 0.f;
            for (int k = 0; k < p.T; ++k) s = fmaf(w[k], emb[k], s);
            p.beff[((size_t)e * p.nnets + net) * WIDTH + tid] = l1.bias[tid] + s;
        }
    }
    for (int idx = tid; idx < p.H * p.A; idx += NTHREADS) {
        const int t = idx / p.A;
        float m = 0.f;
        if (!p.t0[e] && t < p.H - 1) m = p.prev_mean[(size_t)e * p.H * p.A + idx + p.A];
        p.mean[(size_t)e * p.H * p.A + idx] = m;
        p.std[(size_t)e * p.H * p.A + idx] = p.max_std;
    }
    tile_broadcast_row_s(c, p.z0 + (size_t)e * WIDTH);
    __syncthreads();
    f32x16 acc[2][2][2];
    zero4(acc[0]);
    zero4(acc[1]);
    kloop_s(c, p.rew.l[0], 0, ZKB16, acc[0]);
    kloop_s(c, p.dyn.l[0], 0, ZKB16, acc[1]);
    const float *b_rew = p.multitask ? p.beff + ((size_t)e * p.nnets + BE_REW) * WIDTH : p.rew.l[0].bias;
    const float *b_dyn = p.multitask ? p.beff + ((size_t)e * p.nnets + BE_DYN) * WIDTH : p.dyn.l[0].bias;
    const float o_rew = *p.rew.l[0].oscale, o_dyn = *p.dyn.l[0].oscale;
    if ((c.lane >> 5) == 0) {  // row 0 of the tile: register 0 of row tile 0 in lanes 0..31
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int col = (2 * c.wave + ct) * 32 + (c.lane & 31);
            p.cvec[((size_t)e * 2 + 0) * WIDTH + col] = fmaf(acc[0][0][ct][0], o_rew, b_rew[col]);
            p.cvec[((size_t)e * 2 + 1) * WIDTH + col] = fmaf(acc[1][0][ct][0], o_dyn, b_dyn[col]);
        }
    }
}

// ================================================================ kernel: policy-prior trajectories (cf. k_pitraj)
template <int APAD>
__global__ __launch_bounds__(NTHREADS, 2) void ks_pitraj(PiTrajParamsT<NetS> p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int e = blockIdx.x, tid = threadIdx.x;
    CtxT<APAD> c{reinterpret_cast<_Float16 *>(smem), smem + ROWS * CtxT<APAD>::RSH / 2, tid,
                 __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63};
    const float *mask = p.act_mask ? p.act_mask + (size_t)e * p.A : nullptr;
    const float *b_pi = p.multitask ? p.beff + ((size_t)e * p.nnets + BE_PI) * WIDTH : p.pi.l[0].bias;
    const float *b_dyn = p.multitask ? p.beff + ((size_t)e * p.nnets + BE_DYN) * WIDTH : p.dyn.l[0].bias;
    float *zs = p.zscratch + (size_t)e * p.zscratch_estride;
    const int KBA = ZKB16 + p.Apad / 16;
    tile_broadcast_row_s(c, p.z0 + (size_t)e * WIDTH);
    for (int idx = tid; idx < ROWS * WIDTH / 4; idx += NTHREADS)
        *reinterpret_cast<f32x4 *>(zs + 4 * idx) =
            *reinterpret_cast<const f32x4 *>(p.z0 + (size_t)e * WIDTH + 4 * (idx % (WIDTH / 4)));
    __syncthreads();
    for (int t = 0; t < p.H; ++t) {
        layer_full_s<0>(c, p.pi.l[0], b_pi, 0, ZKB16);
        layer_full_s<0>(c, p.pi.l[1], p.pi.l[1].bias, 0, ZKB16);
        const float *tape = p.pi_traj_eps ? p.pi_traj_eps + ((size_t)e * p.H + t) * p.P * p.A : nullptr;
        auto eps = [&](int row, int a) -> float {
            if (row >= p.P) return 0.f;
            if (tape) return tape[row * p.A + a];
            return rng_normal(p.seed, p.call, SITE_PITRAJ, t, e, (unsigned)(row * p.A + a));
        };
        head_pi_s(c, p.pi.l[2], p.A, p.Apad, p.log_std_min, p.log_std_dif, mask, eps,
                  p.actions + ((size_t)e * p.H + t) * p.N * p.A, p.P, nullptr);
        if (t == p.H - 1) break;
        tile_from_global_s(c, zs);
        __syncthreads();
        layer_full_s<0>(c, p.dyn.l[0], b_dyn, 0, KBA);
        layer_full_s<0>(c, p.dyn.l[1], p.dyn.l[1].bias, 0, ZKB16);
        layer_full_s<1>(c, p.dyn.l[2], p.dyn.l[2].bias, 0, ZKB16, zs);
    }
}

// ================================================================ kernel: one CEM iteration's rollouts (cf. k_rollout)
template <int APAD>
__global__ __launch_bounds__(NTHREADS, 2) void ks_rollout(RolloutParamsT<NetS> p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int e = blockIdx.x / p.tiles, tile = blockIdx.x % p.tiles;
    const int tid = threadIdx.x;
    CtxT<APAD> c{reinterpret_cast<_Float16 *>(smem), smem + ROWS * CtxT<APAD>::RSH / 2, tid,
                 __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63};
    float *sm_mean = smem + ROWS * c.RSH / 2 + 1024;  // [H*A] after the tile and the LayerNorm partials
    float *sm_std = sm_mean + p.H * p.A;
    const int row0 = tile * ROWS;
    const float *mask = p.act_mask ? p.act_mask + (size_t)e * p.A : nullptr;
    const float *disc = p.disc_pow + (size_t)e * (p.H + 1);
    const int KBA = ZKB16 + p.Apad / 16;
    float *zs = p.zscratch + (size_t)blockIdx.x * ROWS * WIDTH;
    const int NSLOT = 5 * p.H + 7;
    float *tsc = p.trace_scalars ? p.trace_scalars + ((size_t)e * p.N + row0 + (tid >> 3)) * (p.H + 2 + p.A) : nullptr;

    for (int idx = tid; idx < p.H * p.A; idx += NTHREADS) {
        sm_mean[idx] = p.mean[(size_t)e * p.H * p.A + idx];
        sm_std[idx] = p.std[(size_t)e * p.H * p.A + idx];
    }
    int q0, q1;
    if (p.qidx) {
        q0 = p.qidx[(size_t)e * p.qidx_estride + 0];
        q1 = p.qidx[(size_t)e * p.qidx_estride + 1];
    } else {
        const uint4 r = rng_raw(p.seed, p.call, SITE_QIDX, p.iter, e, 0);
        q0 = (int)(r.x % (unsigned)p.nq);
        q1 = (int)(r.y % (unsigned)(p.nq - 1));
        if (q1 >= q0) ++q1;
    }
    const float *b_rew = p.multitask ? p.beff + ((size_t)e * p.nnets + BE_REW) * WIDTH : p.rew.l[0].bias;
    const float *b_dyn = p.multitask ? p.beff + ((size_t)e * p.nnets + BE_DYN) * WIDTH : p.dyn.l[0].bias;
    const float *b_pi = p.multitask ? p.beff + ((size_t)e * p.nnets + BE_PI) * WIDTH : p.pi.l[0].bias;
    const float *b_q0 = p.multitask ? p.beff + ((size_t)e * p.nnets + BE_Q0 + q0) * WIDTH : p.q[q0].l[0].bias;
    const float *b_q1 = p.multitask ? p.beff + ((size_t)e * p.nnets + BE_Q0 + q1) * WIDTH : p.q[q1].l[0].bias;
    __syncthreads();

    float G = 0.f;
    for (int t = 0; t < p.H; ++t) {
        // ---- actions of step t (tdmpc2.py:176-181) -> operand-form action columns
        {
            float *ag = p.actions + ((size_t)e * p.H + t) * p.N * p.A;
            for (int idx = tid; idx < ROWS * p.Apad; idx += NTHREADS) {
                const int row = idx / p.Apad, a = idx % p.Apad;
                const int n = row0 + row;
                float v = 0.f;
                if (a < p.A) {
                    if (p.given_actions || n < p.P) {
                        v = ag[(size_t)n * p.A + a];
                    } else {
                        float r;
                        const unsigned ridx = (unsigned)(((size_t)t * (p.N - p.P) + (n - p.P)) * p.A + a);
                        if (p.sample_eps)
                            r = p.sample_eps[(size_t)e * p.sample_eps_estride + ridx];
                        else
                            r = rng_normal(p.seed, p.call, SITE_SAMPLE, p.iter, e, ridx);
                        v = sm_mean[t * p.A + a] + sm_std[t * p.A + a] * r;
                        v = fminf(fmaxf(v, -1.f), 1.f);
                    }
                    if (mask && !p.given_actions) v *= mask[a];
                    if (!p.given_actions) ag[(size_t)n * p.A + a] = v;
                }
                put_action(c, row, a, v);
            }
        }
        __syncthreads();
        // ---- first layers of dynamics and reward over the same [z_t | a_t] tile; the dynamics pre-activation is parked
        // in the workgroup's L2-resident scratch tile until the reward chain is done (64 VGPRs not held)
        {
            f32x16 acc[2][2];
            zero4(acc);
            kloop_s(c, p.dyn.l[0], t == 0 ? ZKB16 : 0, KBA, acc);
            store_full_global(c, acc, *p.dyn.l[0].oscale, t == 0 ? p.cvec + ((size_t)e * 2 + 1) * WIDTH : b_dyn, zs);
            zero4(acc);
            kloop_s(c, p.rew.l[0], t == 0 ? ZKB16 : 0, KBA, acc);
            const float o_rew = *p.rew.l[0].oscale;
            __syncthreads();
            store_full_s(c, acc, o_rew, t == 0 ? p.cvec + ((size_t)e * 2 + 0) * WIDTH : b_rew);
        }
        __syncthreads();
        ln_rows_s<0>(c, p.rew.l[0].g, p.rew.l[0].b, nullptr);
        __syncthreads();
        dump_tile_s(c, p.trace_tiles, NSLOT, 5 * t + 0);
        // ---- reward: layer 2, two-hot head
        layer_full_s<0>(c, p.rew.l[1], p.rew.l[1].bias, 0, ZKB16);
        dump_tile_s(c, p.trace_tiles, NSLOT, 5 * t + 1);
        const float r = head_twohot_s(c, p.rew.l[2], p.bins, p.num_bins);
        if (tsc && (tid & 7) == 0) tsc[t] = r;
        G += disc[t] * r;
        // ---- dynamics: pick the parked first layer up from L2, layers 2 and 3 (SimNorm)
        ln_rows_s<0>(c, p.dyn.l[0].g, p.dyn.l[0].b, nullptr, zs);
        __syncthreads();
        dump_tile_s(c, p.trace_tiles, NSLOT, 5 * t + 2);
        layer_full_s<0>(c, p.dyn.l[1], p.dyn.l[1].bias, 0, ZKB16);
        dump_tile_s(c, p.trace_tiles, NSLOT, 5 * t + 3);
        layer_full_s<1>(c, p.dyn.l[2], p.dyn.l[2].bias, 0, ZKB16, t == p.H - 1 ? zs : nullptr);
        dump_tile_s(c, p.trace_tiles, NSLOT, 5 * t + 4);
    }
    // ---- a_H = pi(z_H) (tdmpc2.py:135); z_H was also saved to zs
    layer_full_s<0>(c, p.pi.l[0], b_pi, 0, ZKB16);
    dump_tile_s(c, p.trace_tiles, NSLOT, 5 * p.H + 0);
    layer_full_s<0>(c, p.pi.l[1], p.pi.l[1].bias, 0, ZKB16);
    dump_tile_s(c, p.trace_tiles, NSLOT, 5 * p.H + 1);
    {
        auto eps = [&](int row, int a) -> float {
            const unsigned ridx = (unsigned)((size_t)(row0 + row) * p.A + a);
            if (p.pi_eps) return p.pi_eps[(size_t)e * p.pi_eps_estride + ridx];
            return rng_normal(p.seed, p.call, SITE_PI, p.iter, e, ridx);
        };
        head_pi_s(c, p.pi.l[2], p.A, p.Apad, p.log_std_min, p.log_std_dif, mask, eps, nullptr, 0,
                  tsc ? tsc + p.H + 2 : nullptr);
    }
    tile_from_global_s(c, zs);
    __syncthreads();
    dump_tile_s(c, p.trace_tiles, NSLOT, 5 * p.H + 2);
    // ---- Q(z_H, a_H): the two selected heads, first layers in one pass (world_model.py:186-216)
    {
        // z_H has been read back from zs above (tile_from_global_s + barrier): the scratch tile is free to park the
        // second head's first-layer pre-activation
        f32x16 acc[2][2];
        zero4(acc);
        kloop_s(c, p.q[q1].l[0], 0, KBA, acc);
        store_full_global(c, acc, *p.q[q1].l[0].oscale, b_q1, zs);
        zero4(acc);
        kloop_s(c, p.q[q0].l[0], 0, KBA, acc);
        const float o_q0 = *p.q[q0].l[0].oscale;
        __syncthreads();
        store_full_s(c, acc, o_q0, b_q0);
    }
    __syncthreads();
    ln_rows_s<0>(c, p.q[q0].l[0].g, p.q[q0].l[0].b, nullptr);
    __syncthreads();
    dump_tile_s(c, p.trace_tiles, NSLOT, 5 * p.H + 3);
    layer_full_s<0>(c, p.q[q0].l[1], p.q[q0].l[1].bias, 0, ZKB16);
    dump_tile_s(c, p.trace_tiles, NSLOT, 5 * p.H + 4);
    const float qa = head_twohot_s(c, p.q[q0].l[2], p.bins, p.num_bins);
    ln_rows_s<0>(c, p.q[q1].l[0].g, p.q[q1].l[0].b, nullptr, zs);
    __syncthreads();
    dump_tile_s(c, p.trace_tiles, NSLOT, 5 * p.H + 5);
    layer_full_s<0>(c, p.q[q1].l[1], p.q[q1].l[1].bias, 0, ZKB16);
    dump_tile_s(c, p.trace_tiles, NSLOT, 5 * p.H + 6);
    const float qb = head_twohot_s(c, p.q[q1].l[2], p.bins, p.num_bins);
    if (tsc && (tid & 7) == 0) {
        tsc[p.H] = qa;
        tsc[p.H + 1] = qb;
    }
    if ((tid & 7) == 0) p.value[(size_t)e * p.N + row0 + (tid >> 3)] = G + disc[p.H] * ((qa + qb) / 2.f);
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
// kw such that max|W| 2^kw in [2^13, 2^14); wscale = 2^kw (for packing), oscale = 2^-(kw + ACT_SCALE_LOG2)
__global__ void k_wscale(const unsigned int *maxbits, float *wscale, float *oscale) {
    const float m = __uint_as_float(*maxbits);
    int ex = 0;
    if (m > 0.f) frexpf(m, &ex);  // m = f 2^ex, f in [0.5, 1)
    int kw = 14 - ex;
    kw = kw > 40 ? 40 : (kw < -40 ? -40 : kw);
    *wscale = ldexpf(1.f, kw);
    *oscale = ldexpf(1.f, -(kw + ACT_SCALE_LOG2));
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
