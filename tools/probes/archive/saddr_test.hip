#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void a_gld(f32x4 &dst, unsigned voff, const char *sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF));
}
__global__ void k(const float *src, float *dst, int blocks) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char *u0 = reinterpret_cast<const char *>(src) + (size_t)wave * blocks * 2048;
    f32x4 a, b;
    float acc = 0.f;
    for (int kk = 0; kk < blocks; ++kk) {
        const char *pn = u0 + (size_t)kk * 2048;
        a_gld<0>(a, lane * 16u, pn);
        a_gld<1024>(b, lane * 16u, pn);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += a[0] + b[0];
    }
    dst[threadIdx.x] = acc;
}
int main() {
    const int blocks = 35, waves = 8;
    size_t n = (size_t)waves * blocks * 512;
    float *h = (float *)malloc(n * 4);
    for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 97);
    float *s, *d;
    hipMalloc(&s, n * 4); hipMalloc(&d, 512 * 4);
    hipMemcpy(s, h, n * 4, hipMemcpyHostToDevice);
    k<<<1, 512>>>(s, d, blocks);
    float out[512];
    hipError_t e = hipMemcpy(out, d, 512 * 4, hipMemcpyDeviceToHost);
    printf("err %d\n", (int)e);
    int bad = 0;
    for (int t = 0; t < 512; ++t) {
        int w = t / 64, l = t % 64; float ref = 0;
        for (int kk = 0; kk < blocks; ++kk) { size_t base = ((size_t)w * blocks + kk) * 512; ref += h[base + l * 4] + h[base + 256 + l * 4]; }
        if (ref != out[t]) { if (bad < 5) printf("mismatch t=%d got %f want %f\n", t, out[t], ref); ++bad; }
    }
    printf("bad %d\n", bad);
    return 0;
}
