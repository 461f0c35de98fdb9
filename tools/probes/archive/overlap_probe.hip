// Issue-overlap probe (MI355X): can one wave's VALU epilogue math hide behind its own (and its SIMD partner's)
// f16 MFMAs?  8 waves per workgroup, one workgroup per CU (LDS-limited like ks_rollout), three variants:
//   A: 12 MFMAs per step            B: NV mish-like VALU element updates per step            C: both in one stream
// build: hipcc --offload-arch=gfx950 -O3 -o overlap_probe overlap_probe.hip ; run: ./overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float mishy(float x, float rstd, float shift, float g, float b) {
    const float y = fmaf(fmaf(x, rstd, shift), g, b);
    const float e = __expf(fminf(y, 20.f));
    const float n = e * (e + 2.f);
    return y * (n * __builtin_amdgcn_rcpf(n + 2.f));
}

template <int MODE, int NV>
__global__ __launch_bounds__(512) void probe(float *out, int steps, float seed) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + lane * 0.001f + i); b[i] = (_Float16)(seed * 0.5f + i); }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float v[NV];
    for (int i = 0; i < NV; ++i) v[i] = seed + 0.01f * i + lane * 1e-3f;
    _Float16 h[NV], l[NV];
    for (int s = 0; s < steps; ++s) {
        if (MODE != 1) {  // 0, 2, 3
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
        }
        if (MODE != 0) {  // 1, 2, 3
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float y = mishy(v[i], 1.01f, 0.02f, 0.99f, 0.01f) * 32.f;
                h[i] = (_Float16)y;
                l[i] = (_Float16)(y - (float)h[i]);
                v[i] = (float)h[i] + (float)l[i] * 0.03125f;  // keep the chain alive
            }
        }
        if (MODE == 3) {  // pin the interleave: one MFMA, then its share of the VALU work
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, (NV * 20 + 11) / 12, 0);
            }
        }
    }
    float r = 0.f;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) r += acc[t][i];
    for (int i = 0; i < NV; ++i) r += v[i];
    if (r == 12345.678f) out[threadIdx.x] = r + lds[threadIdx.x];
}

template <int MODE, int NV>
float run(float *out, int steps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)probe<MODE, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    probe<MODE, NV><<<256, 512, 140 * 1024>>>(out, 64, 0.5f);
    hipEventRecord(e0);
    probe<MODE, NV><<<256, 512, 140 * 1024>>>(out, steps, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *out; hipMalloc(&out, 4096);
    const int steps = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        const float a = run<0, 4>(out, steps);
        printf("A (12 MFMA/step)            : %.3f ms  -> %.1f cycles/step/SIMD@1.9GHz (2 waves)\n", a, a * 1e-3 * 1.9e9 / steps);
        const float b4 = run<1, 4>(out, steps), c4 = run<2, 4>(out, steps), d4 = run<3, 4>(out, steps);
        printf("NV=4 : B %.3f ms  C %.3f ms  pinned %.3f ms (A+B %.3f, max %.3f)\n", b4, c4, d4, a + b4, a > b4 ? a : b4);
        const float b6 = run<1, 6>(out, steps), c6 = run<2, 6>(out, steps), d6 = run<3, 6>(out, steps);
        printf("NV=6 : B %.3f ms  C %.3f ms  pinned %.3f ms (A+B %.3f, max %.3f)\n", b6, c6, d6, a + b6, a > b6 ? a : b6);
        const float b8 = run<1, 8>(out, steps), c8 = run<2, 8>(out, steps), d8 = run<3, 8>(out, steps);
        printf("NV=8 : B %.3f ms  C %.3f ms  pinned %.3f ms (A+B %.3f, max %.3f)\n", b8, c8, d8, a + b8, a > b8 ? a : b8);
    }
    return 0;
}
