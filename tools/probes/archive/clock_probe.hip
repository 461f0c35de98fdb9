#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long *o) {
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 20000; ++i) __builtin_amdgcn_s_sleep(10);
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
    o[0] = r1 - r0; o[1] = c1 - c0; o[2] = wall_clock64();
}
int main() {
    int v = 0; hipDeviceGetAttribute(&v, hipDeviceAttributeWallClockRate, 0); printf("wall clock rate %d kHz\n", v);
    unsigned long long *d, h[3]; hipMalloc(&d, 24);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); k<<<1, 1>>>(d); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("kernel %.3f ms: s_memrealtime ticks %llu (%.1f MHz), s_memtime ticks %llu (%.1f MHz)\n", ms, h[0], h[0] / ms / 1e3, h[1], h[1] / ms / 1e3);
}
