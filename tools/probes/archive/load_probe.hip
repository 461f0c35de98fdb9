// Probe (MI355X): do the k-loop's operand loads (16 B/lane weight fragments from L2/L1, 16 B/lane activation fragments
// from LDS) overlap with the wave's own f16 MFMAs, or do they serialise?  Same geometry as ks_rollout: 8 waves per
// workgroup, one workgroup per CU, 12 MFMAs + NG global + NL LDS 16-byte loads per step.
// build: hipcc --offload-arch=gfx950 -O3 -o load_probe load_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE bit 0: MFMAs, bit 1: global loads, bit 2: LDS loads; operands of the MFMAs come from the loaded data when loads are on
template <int MODE, int PF, int FT, int NW>
__global__ __launch_bounds__(64 * NW) void probe(const f32x4 *__restrict__ w, float *out, int steps, int wg_stride_vec) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 *l4 = reinterpret_cast<f32x4 *>(lds);
    for (int i = threadIdx.x; i < 140 * 1024 / 16; i += 64 * NW) l4[i] = f32x4{1e-3f * i, 0.f, 1.f, 2.f};
    __syncthreads();
    const f32x4 *wp = w + (size_t)(blockIdx.x % 8) * wg_stride_vec + wave * 64 * 2 * FT + lane;  // 2 FT KB per wave per step
    f32x16 acc[2 * FT];
    for (int t = 0; t < 2 * FT; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    f32x4 g[PF][2 * FT], a[4];
    for (int d = 0; d < PF; ++d) for (int q = 0; q < 2 * FT; ++q) g[d][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < 4; ++q) a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nstep_wrap = wg_stride_vec / (NW * 64 * 2 * FT);  // steps before the per-workgroup weight window wraps
    int sidx = 0;
#pragma unroll 1
    for (int s = 0; s < steps; s += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            f32x4 gn[2 * FT], an[4];
            if (MODE & 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) an[q] = l4[((s + d) * 37 + q * 64 * 9 + lane) % (140 * 64)];
            }
            if (MODE & 1) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int t = 0; t < 2 * FT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, g[d][(t + p) % (2 * FT)]), __builtin_bit_cast(f16x8, a[(t + p) & 3]), acc[t], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 2 * FT; ++t) acc[t][0] += g[d][t][0] + a[t & 3][1];
            }
            if (MODE & 2) {
#pragma unroll
                for (int q = 0; q < 2 * FT; ++q) gn[q] = wp[(size_t)sidx * (NW * 64 * 2 * FT) + q * 64];
                sidx = sidx + 1 == nstep_wrap ? 0 : sidx + 1;
#pragma unroll
                for (int q = 0; q < 2 * FT; ++q) g[d][q] = gn[q];
            }
            if (MODE & 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = an[q];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0.f;
    for (int t = 0; t < 2 * FT; ++t) for (int i = 0; i < 16; ++i) r += acc[t][i];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MODE, int PF, int FT = 2, int NW = 8>
float run(const f32x4 *w, float *out, int steps, int stride) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void *)probe<MODE, PF, FT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    probe<MODE, PF, FT, NW><<<256, 64 * NW, 140 * 1024>>>(w, out, 64, stride);
    (void)hipEventRecord(e0);
    probe<MODE, PF, FT, NW><<<256, 64 * NW, 140 * 1024>>>(w, out, steps, stride);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    // weight window: 8 distinct 2.5 MB regions (20 MB total, like the c2 planner weights): L2 hits + some MALL traffic
    const int stride = 2560 * 1024 / 16;
    f32x4 *w; float *out;
    (void)hipMalloc(&w, (size_t)8 * stride * 16); (void)hipMalloc(&out, 4096);
    (void)hipMemset(w, 0, (size_t)8 * stride * 16);
    const int steps = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        printf("PF=2: mfma %.3f  glob %.3f  lds %.3f  glob+lds %.3f | mfma+glob %.3f  mfma+lds %.3f  all %.3f ms\n",
               run<1, 2>(w, out, steps, stride), run<2, 2>(w, out, steps, stride), run<4, 2>(w, out, steps, stride),
               run<6, 2>(w, out, steps, stride), run<3, 2>(w, out, steps, stride), run<5, 2>(w, out, steps, stride),
               run<7, 2>(w, out, steps, stride));
        printf("PF=4: mfma %.3f  glob %.3f  lds %.3f  glob+lds %.3f | mfma+glob %.3f  mfma+lds %.3f  all %.3f ms\n",
               run<1, 4>(w, out, steps, stride), run<2, 4>(w, out, steps, stride), run<4, 4>(w, out, steps, stride),
               run<6, 4>(w, out, steps, stride), run<3, 4>(w, out, steps, stride), run<5, 4>(w, out, steps, stride),
               run<7, 4>(w, out, steps, stride));
    }
    for (int rep = 0; rep < 2; ++rep)
        printf("4 waves x FT=4, PF=2: mfma %.3f  glob %.3f  lds %.3f  glob+lds %.3f | mfma+glob %.3f  mfma+lds %.3f  all %.3f ms\n",
               run<1, 2, 4, 4>(w, out, steps, stride), run<2, 2, 4, 4>(w, out, steps, stride), run<4, 2, 4, 4>(w, out, steps, stride),
               run<6, 2, 4, 4>(w, out, steps, stride), run<3, 2, 4, 4>(w, out, steps, stride), run<5, 2, 4, 4>(w, out, steps, stride),
               run<7, 2, 4, 4>(w, out, steps, stride));
    printf("ideal mfma: %d steps x 24 MFMA x 32 cycles / ~1.9 GHz = %.3f ms; bytes per step per CU = 32 KB global + 32 KB LDS\n", steps,
           steps * 768.0 / 1.9e6);
    return 0;
}
