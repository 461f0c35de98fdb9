// What does one layer-to-layer hand-over between the workgroups of a CLUSTER cost on MI355X?
// (single-plan latency: the 512 output features of a layer split over C workgroups on C CUs, every layer ends with
//  "write my 32 rows x 64 features of raw sums, wait for the other C - 1, read all 32 x 512")
// Each workgroup (512 threads) per round: one 16-byte store per thread (its 8 KB slice, register order), cluster barrier,
// 8 x 16-byte loads per thread (the 64 KB tile), a few FMAs.  Variants:
//   ST: 0 plain stores, 1 agent-scope (sc1 write-through) stores
//   LD: 0 sc0 loads (miss the L1, may hit this XCD's L2), 1 sc1 loads (agent scope)
//   BAR: 0 one atomic counter per cluster (add, then poll), 1 one flag word per member (store, then 8 lanes poll)
//   GROUP: 0 cluster = consecutive workgroup ids (members spread over the 8 XCDs), 1 cluster = ids equal mod 8 (same XCD)
// Every wait is bounded (no hang): a timed-out wait sets err and the kernel runs on.
// Checks the data too: round r, member m writes value f(r, m, thread); every reader sums what it read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int ST>
__device__ __forceinline__ void st16(float *p, f32x4 v) {
    if (ST == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
template <int LD>
__device__ __forceinline__ void ld16(f32x4 &v, const float *p) {
    if (LD == 0) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
}

constexpr int MAXSPIN = 1 << 18;

template <int ST, int LD, int BAR, int C>
__global__ __launch_bounds__(512) void k(float *xbuf, unsigned *bar, int rounds, int group, int nclusters, float *out, unsigned *err,
                                         unsigned long long *cyc) {
    const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int cl, rank;
    if (group == 0) { cl = wg / C; rank = wg % C; }
    else { const int x = wg % 8, q = wg / 8; cl = x + 8 * (q / C); rank = q % C; }  // same XCD (round-robin dispatch)
    if (cl >= nclusters) return;
    float *xb = xbuf + (size_t)cl * 2 * C * 2048;  // two buffers of C slices of 2048 floats (8 KB)
    unsigned *cnt = bar + (size_t)cl * 64;          // counter at [0], flags at [16 + m] (own 256-byte line group)
    float acc = 0.f;
    unsigned long long t_bar = 0, t_ld = 0;
    for (int r = 0; r < rounds; ++r) {
        float *dst = xb + (size_t)(r & 1) * C * 2048 + (size_t)rank * 2048 + tid * 4;
        const float base = (float)((r * 7 + rank * 3) & 255);
        f32x4 v = {base, base + 1.f, (float)(tid & 15), 1.f};
        st16<ST>(dst, v);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (BAR == 0) {
            if (tid == 0) {
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)(r + 1) * C;
                int spin = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spin > MAXSPIN) { atomicOr(err, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        } else {
            if (tid == 0) __hip_atomic_store(cnt + 16 + rank, (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < C) {
                int spin = 0;
                while (__hip_atomic_load(cnt + 16 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r + 1)) {
                    if (++spin > MAXSPIN) { atomicOr(err, 2u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        __syncthreads();
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        const float *src = xb + (size_t)(r & 1) * C * 2048;
        f32x4 x[C];
#pragma unroll
        for (int m = 0; m < C; ++m) ld16<LD>(x[m], src + (size_t)m * 2048 + ((wave * 64 + lane) * 4));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int m = 0; m < C; ++m) {
            const float want = (float)((r * 7 + m * 3) & 255);
            if (x[m][0] != want || x[m][1] != want + 1.f || x[m][2] != (float)(tid & 15)) atomicOr(err, 4u);
            acc += x[m][0] + x[m][3];
        }
        t_bar += t1 - t0;
        t_ld += t2 - t1;
    }
    out[(size_t)wg * 512 + tid] = acc;
    if (tid == 0) {
        atomicAdd(cyc + 0, t_bar);
        atomicAdd(cyc + 1, t_ld);
    }
}

template <int ST, int LD, int BAR, int C>
void run(const char *name, int group, int nclusters, int rounds) {
    const int nwg = group == 0 ? nclusters * C : ((nclusters + 7) / 8) * 8 * C;
    float *xbuf, *out;
    unsigned *bar, *err;
    unsigned long long *cyc;
    CK(hipMalloc(&xbuf, (size_t)nclusters * 2 * C * 2048 * 4));
    CK(hipMalloc(&out, (size_t)nwg * 512 * 4));
    CK(hipMalloc(&bar, (size_t)nclusters * 64 * 4));
    CK(hipMalloc(&err, 4));
    CK(hipMalloc(&cyc, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned herr = 0;
    unsigned long long hc[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(bar, 0, (size_t)nclusters * 64 * 4));
        CK(hipMemset(err, 0, 4));
        CK(hipMemset(cyc, 0, 16));
        CK(hipMemset(xbuf, 0, (size_t)nclusters * 2 * C * 2048 * 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        k<ST, LD, BAR, C><<<nwg, 512>>>(xbuf, bar, rounds, group, nclusters, out, err, cyc);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        unsigned h;
        CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        herr |= h;
        CK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
    }
    printf("%-34s C=%d group=%s clusters=%3d: %7.3f us / round   (wait %6.0f + load %6.0f ticks of 10 ns per round)  err=%u\n", name, C,
           group ? "same-xcd" : "spread  ", nclusters, best * 1e3f / rounds, (double)hc[0] / (nclusters * C) / rounds,
           (double)hc[1] / (nclusters * C) / rounds, herr);
    fflush(stdout);
    CK(hipFree(xbuf)); CK(hipFree(out)); CK(hipFree(bar)); CK(hipFree(err)); CK(hipFree(cyc));
}

int main() {
    const int R = 2000;
    for (int group = 0; group < 2; ++group) {
        run<1, 1, 0, 8>("sc1 store, sc1 load, counter", group, 16, R);
        run<1, 1, 1, 8>("sc1 store, sc1 load, flags", group, 16, R);
        run<0, 1, 1, 8>("plain store, sc1 load, flags", group, 16, R);
        run<0, 0, 1, 8>("plain store, sc0 load, flags", group, 16, R);
        run<1, 0, 1, 8>("sc1 store, sc0 load, flags", group, 16, R);
        run<0, 0, 0, 8>("plain store, sc0 load, counter", group, 16, R);
        run<1, 1, 1, 4>("sc1 store, sc1 load, flags", group, 16, R);
        run<1, 1, 1, 16>("sc1 store, sc1 load, flags", group, 16, R);
        run<1, 1, 1, 8>("sc1 store, sc1 load, flags", group, 1, R);
        run<1, 1, 1, 8>("sc1 store, sc1 load, flags", group, 32, R);
    }
    return 0;
}
