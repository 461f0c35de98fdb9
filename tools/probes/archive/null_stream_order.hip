// r03k hunt (profiles/README.md): does the NULL stream still run its kernels in order after it has waited on an event of a
// non-blocking stream (the layered family's fork / join)?  Plain HIP, no library of this repository.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/null_stream_order.hip -o /tmp/null_stream_order && /tmp/null_stream_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

__global__ void slow_fill(int *buf, int n, int val, long long spin) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = val;
}
__global__ void copy_k(const int *src, int *dst, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void busy(int *p, long long spin) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (p) atomicAdd(p, 1);
}

static int chain_ok(int *a, int *b, int *c, int n, int val, std::vector<int> &host) {
    // three dependent kernels back to back on the NULL stream: a := val (slow), b := a, c := b
    if (hipMemsetAsync(a, 0, n * 4, 0) != hipSuccess) return -1;
    if (hipMemsetAsync(b, 0, n * 4, 0) != hipSuccess) return -1;
    if (hipMemsetAsync(c, 0, n * 4, 0) != hipSuccess) return -1;
    hipLaunchKernelGGL(slow_fill, dim3(8), dim3(256), 0, 0, a, n, val, 20000LL);  // 200 us at 100 MHz
    hipLaunchKernelGGL(copy_k, dim3(8), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(copy_k, dim3(8), dim3(256), 0, 0, b, c, n);
    if (hipMemcpy(host.data(), c, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += host[i] != val;
    return bad;
}

int main() {
    const int n = 1 << 16;
    int *a, *b, *c, *cnt;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&cnt, 4));
    std::vector<int> host(n);
    int val = 1;
    for (int mode = 0; mode < 4; ++mode) {
        // 0: nothing before; 1: fork / join NULL <-> non-blocking stream, stream + events kept; 2: the same, then destroyed;
        // 3: the same with the join missing its device synchronize (the chain follows at once)
        int failures = 0, trials = 200;
        for (int t = 0; t < trials; ++t) {
            hipStream_t s = nullptr; hipEvent_t ef = nullptr, es = nullptr;
            if (mode) {
                CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
                CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&es, hipEventDisableTiming));
                for (int rep = 0; rep < 3; ++rep) {
                    hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, 0, cnt, 2000LL);
                    CK(hipEventRecord(ef, 0)); CK(hipStreamWaitEvent(s, ef, 0));
                    hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, s, cnt, 5000LL);
                    hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, 0, cnt, 3000LL);
                    CK(hipEventRecord(es, s)); CK(hipStreamWaitEvent(0, es, 0));
                    hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, 0, cnt, 1000LL);
                }
                if (mode != 3) CK(hipDeviceSynchronize());
                if (mode == 2) { CK(hipEventDestroy(ef)); CK(hipEventDestroy(es)); CK(hipStreamDestroy(s)); }
            }
            const int bad = chain_ok(a, b, c, n, ++val, host);
            if (bad < 0) { printf("HIP error in chain\n"); return 2; }
            failures += bad != 0;
            if (mode == 1 || mode == 3) { CK(hipDeviceSynchronize()); CK(hipEventDestroy(ef)); CK(hipEventDestroy(es)); CK(hipStreamDestroy(s)); }
        }
        printf("mode %d: %d of %d chains on the NULL stream read a value before it was written\n", mode, failures, trials);
    }
    return 0;
}
