// MI355X (gfx950 / CDNA4) native TD-MPC2 MPPI/CEM planner.
//
// What this file implements (reference: nicklashansen/tdmpc2, paths relative to its root):
//   TDMPC2._plan after encode()            tdmpc2/tdmpc2.py:154-206
//   TDMPC2._estimate_value                 tdmpc2/tdmpc2.py:122-136
//   WorldModel.next / reward / pi / Q      tdmpc2/common/world_model.py:114-216
//   NormedLinear / SimNorm                 tdmpc2/common/layers.py:74-118
//   two_hot_inv / symexp / log_std / gumbel_softmax_sample   tdmpc2/common/math.py
//
// This file: the C ABI, the handle, weight packing, the elite-refit kernel shared by every path, and the host side of
// the fused family.  Kernels:
//   fused_kernels.cuh    fused 512-wide family (ks_setup / ks_pitraj / ks_rollout): one persistent workgroup keeps 32 or
//                        64 sample rows of one plan in LDS for a whole CEM iteration; templated on the arithmetic
//                        (f16x2 split on the f16 matrix pipe -- the default -- or exact fp32 MFMA), the action padding
//                        and the workgroup geometry
//                        + ks_value: pi + two Q heads on a batch of latent rows (TDMPC2._td_target, tdmpc2.py:239-254,
//                        and the forward half of update_pi, tdmpc2.py:208-225)
//   layered_kernels.cuh, layered_split.cuh, layered_host.cuh
//                        layer-at-a-time family for every other model size and for episodic planning
//   encoder_kernels.cuh  WorldModel.encode for state observations (world_model.py:103-112)
// Weights are re-packed once (bind) into MFMA fragment order so that a wave's global_load_dwordx4 reads 1 KiB
// contiguous; one small workgroup per plan does nan_to_num + top-k + score + mean/std refit (+ the final Gumbel pick)
// between rollout launches (k_refit below).  DESIGN.md has the full account.
#include <cstdarg>
#include <cstring>
#include <new>

#include "handle.h"
#include "launch.h"

namespace {

// one workgroup per plan (layered family; tdmpc2_plan_refit)
__global__ void k_refit(RefitParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    refit_plan(p, blockIdx.x, smem, threadIdx.x, blockDim.x);
}

}  // namespace
int tdk::launch_refit(const RefitParams &fp, int E, int N, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL(k_refit, dim3(E), dim3(refit_threads(N)), lds, st, fp);
    LAUNCH_CHECK();
    return 0;
}
namespace {

// ================================================================ weight packing kernels
// dst[ct][kb][lane][r] = W[row = ct*32 + (lane&31)][k = kb*8 + 4*(lane>>5) + r]; the packed k axis is
// [z columns (nz) | action columns (na, zero padded to a multiple of 8)], source columns are
// [z (nz) | task_emb (nt) | action (na)] (tdmpc2/common/world_model.py:118-120).
__global__ void k_pack_weight(const float *W, int out, int in, int nz, int nt, int na, int CT, int KB, float *dst) {
    const size_t total = (size_t)CT * KB * 256;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int r = idx & 3, lane = (idx >> 2) & 63;
        const size_t blk = idx >> 8;
        const int kb = blk % KB, ct = blk / KB;
        const int row = ct * 32 + (lane & 31);
        const int k = kb * 8 + 4 * (lane >> 5) + r;
        float v = 0.f;
        if (row < out) {
            int src = -1;
            if (k < nz) src = k;
            else if (k - nz < na) src = nz + nt + (k - nz);
            if (src >= 0 && src < in) v = W[(size_t)row * in + src];
        }
        dst[idx] = v;
    }
}
__global__ void k_copy_cols(const float *W, int out, int in, int col0, int ncols, float *dst) {
    const size_t total = (size_t)out * ncols;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = idx % ncols;
        const size_t r = idx / ncols;
        dst[idx] = W[r * in + col0 + c];
    }
}
__global__ void k_copy_pad(const float *src, int n, int npad, float *dst) {
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < npad; idx += gridDim.x * blockDim.x)
        dst[idx] = idx < n ? src[idx] : 0.f;
}

#include "bind_kernels.cuh"
#include "encoder_kernels.cuh"

// ================================================================ the in-kernel generator as a noise tape (tdmpc2_plan_export_noise)
// Every draw of a tape = NULL plan, written with the index formulas and device functions of the kernels that consume them
// (ks_rollout / ks_rollout_cl / ks_pitraj / l_sample / l_pi_head / l_qidx / refit_plan), for environments [e0, e0 + n).
struct NoiseExportParams {
    int e0, n, H, N, P, A, K, I, nq, hp;
    unsigned long long seed;
    unsigned int call;
    tdmpc2_noise_out o;
};
__global__ void k_export_noise(NoiseExportParams p) {
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
    const int NS = p.N - p.P, hpa = (p.A + 1) / 2;
    if (p.o.pi_traj_eps)  // [n,H,P,A]: rng_normal(SITE_PITRAJ, iter = t, row * A + a)
        for (size_t i = gtid; i < (size_t)p.n * p.H * p.P * p.A; i += gsz) {
            const unsigned ra = (unsigned)(i % ((size_t)p.P * p.A));
            const size_t r = i / ((size_t)p.P * p.A);
            const int t = (int)(r % p.H), e = (int)(r / p.H);
            p.o.pi_traj_eps[i] = rng_normal(p.seed, p.call, SITE_PITRAJ, t, p.e0 + e, ra);
        }
    if (p.o.sample_eps)  // [n,I,H,N-P,A]: Philox pairs over (t, n, a / 2) with the fused tile's action padding
        for (size_t i = gtid; i < (size_t)p.n * p.I * p.H * NS * hpa; i += gsz) {
            const int a0 = 2 * (int)(i % hpa);
            size_t r = i / hpa;
            const int ns = (int)(r % NS);
            r /= NS;
            const int t = (int)(r % p.H);
            r /= p.H;
            const int it = (int)(r % p.I), e = (int)(r / p.I);
            float z0, z1;
            rng_normal2(p.seed, p.call, SITE_SAMPLE, it, p.e0 + e, (unsigned)(((size_t)t * NS + ns) * p.hp + a0 / 2), z0, z1);
            float *dst = p.o.sample_eps + ((((size_t)e * p.I + it) * p.H + t) * NS + ns) * p.A + a0;
            dst[0] = z0;
            if (a0 + 1 < p.A) dst[1] = z1;
        }
    if (p.o.pi_eps)  // [n,I,N,A]: rng_normal(SITE_PI, iter, n * A + a)
        for (size_t i = gtid; i < (size_t)p.n * p.I * p.N * p.A; i += gsz) {
            const unsigned ra = (unsigned)(i % ((size_t)p.N * p.A));
            const size_t r = i / ((size_t)p.N * p.A);
            const int it = (int)(r % p.I), e = (int)(r / p.I);
            p.o.pi_eps[i] = rng_normal(p.seed, p.call, SITE_PI, it, p.e0 + e, ra);
        }
    if (p.o.qidx)  // [n,I,2]: two distinct heads, uniform over ordered pairs (the randperm(nq)[:2] of world_model.py:212)
        for (size_t i = gtid; i < (size_t)p.n * p.I; i += gsz) {
            const int it = (int)(i % p.I), e = (int)(i / p.I);
            const uint4 r = rng_raw(p.seed, p.call, SITE_QIDX, it, p.e0 + e, 0);
            int q0 = (int)(r.x % (unsigned)p.nq), q1 = (int)(r.y % (unsigned)(p.nq - 1));
            if (q1 >= q0) ++q1;
            p.o.qidx[2 * i] = q0;
            p.o.qidx[2 * i + 1] = q1;
        }
    if (p.o.gumbel_exp)  // [n,K]
        for (size_t i = gtid; i < (size_t)p.n * p.K; i += gsz)
            p.o.gumbel_exp[i] = rng_exponential(p.seed, p.call, SITE_GUMBEL, 0, p.e0 + (int)(i / p.K), (unsigned)(i % p.K));
    if (p.o.final_eps)  // [n,A]
        for (size_t i = gtid; i < (size_t)p.n * p.A; i += gsz)
            p.o.final_eps[i] = rng_normal(p.seed, p.call, SITE_FINAL, 0, p.e0 + (int)(i / p.A), (unsigned)(i % p.A));
}

// ================================================================ host side
thread_local std::string g_err;
}  // namespace
int tdk::fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
namespace {

// Every entry point that allocates or launches runs on the handle's device and restores the caller's current device
// (a C caller with several GPUs may have another one current).
struct DevGuard {
    int prev = -1;
    bool ok = true;
    explicit DevGuard(int dev) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (cur != dev) {
            ok = hipSetDevice(dev) == hipSuccess;
            if (ok) prev = cur;
        }
    }
    ~DevGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DevGuard(const DevGuard &) = delete;
    DevGuard &operator=(const DevGuard &) = delete;
};


}  // namespace


// The layered path's second stream and its three events come from a process-wide pool and go back to it when a handle is
// destroyed (creating a stream costs a hardware-queue set-up; handles come and go in a training script's evaluation loop).
// (Round 3 believed the pool also cured garbage plans seen after hipStreamDestroy -- profiles/README.md r03k.  It did not: the
// cause was a test moving ONE handle between two streams without ordering them, see StreamTurn below; tools/gpu_r4n.sh
// reproduced the failure with and without the destroys.)
struct SideRes {
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
};
struct SidePool {
    std::mutex mu;
    std::vector<SideRes> free_list[64];
};
static SidePool g_side_pool;
static bool side_acquire(int dev, SideRes *out) {
    const int d = dev >= 0 && dev < 64 ? dev : 0;
    {
        std::lock_guard<std::mutex> lk(g_side_pool.mu);
        if (!g_side_pool.free_list[d].empty()) {
            *out = g_side_pool.free_list[d].back();
            g_side_pool.free_list[d].pop_back();
            return true;
        }
    }
    SideRes r;
    if (hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) != hipSuccess) return false;  // (a priority above / below the caller's stream loses 2-13 %: profiles/README.md r4u)
    for (int i = 0; i < 3; ++i)
        if (hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming) != hipSuccess) return false;  // (a failed create leaks what it made)
    *out = r;
    return true;
}
static void side_release(int dev, const SideRes &r) {
    if (!r.stream) return;
    (void)hipStreamSynchronize(r.stream);
    std::lock_guard<std::mutex> lk(g_side_pool.mu);
    g_side_pool.free_list[dev >= 0 && dev < 64 ? dev : 0].push_back(r);
}


namespace {

// A handle owns one workspace and one call counter: two threads inside the same handle would corrupt both.  The second
// caller gets TDMPC2_ERR_STATE instead (use one handle per thread / stream; handles share nothing).
struct Busy {
    tdmpc2_plan *h;
    bool ok;
    explicit Busy(tdmpc2_plan *hh) : h(hh), ok(false) {
        int expected = 0;
        ok = h->busy.compare_exchange_strong(expected, 1, std::memory_order_acquire);
    }
    ~Busy() {
        if (ok) h->busy.store(0, std::memory_order_release);
    }
    Busy(const Busy &) = delete;
    Busy &operator=(const Busy &) = delete;
};
#define ENTER(h)                                                                                                   \
    Busy busy_(h);                                                                                                 \
    if (!busy_.ok) return fail(TDMPC2_ERR_STATE, "the handle is in use by another call (handles are not reentrant)"); \
    DevGuard dev_((h)->cfg.device);                                                                                \
    if (!dev_.ok) return fail(TDMPC2_ERR_HIP, "hipSetDevice(%d) failed", (h)->cfg.device)

// One workspace also means one stream at a time.  A caller that moves a handle to another stream owes the ordering between the
// two (as with any stream-bound workspace); the handle nevertheless orders its own calls: every call on a stream leaves an event
// behind, and the first call on a DIFFERENT stream waits for it.  (Found the hard way, profiles/README.md "r03k, root cause": a
// test planned on the NULL stream and warmed up a graph capture on a non-blocking stream right behind it -- the second plan's
// set-up kernels overwrote the first plan's workspace whenever a second hardware queue already existed.)  Not while the
// stream is capturing: a captured call replays wherever its graph is launched, which the handle cannot see.
struct StreamTurn {
    tdmpc2_plan *h;
    hipStream_t st;
    bool live;
    hipError_t err;
    StreamTurn(tdmpc2_plan *hh, hipStream_t s) : h(hh), st(s), live(false), err(hipSuccess) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
#ifdef TDMPC2_TEST_HOOKS
        static const bool off = getenv("TDMPC2_DEBUG_NO_TURN") != nullptr;  // negative control of the test that pins this (hooks build only)
#else
        constexpr bool off = false;
#endif
        if (!h->turn_ev || off) return;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        if (cap != hipStreamCaptureStatusNone) return;
        live = true;
        if (h->turn_valid && h->turn_stream != st) err = hipStreamWaitEvent(st, h->turn_ev, 0);
    }
    ~StreamTurn() {
        if (!live) return;
        if (hipEventRecord(h->turn_ev, st) == hipSuccess) {
            h->turn_stream = st;
            h->turn_valid = true;
        }
    }
    StreamTurn(const StreamTurn &) = delete;
    StreamTurn &operator=(const StreamTurn &) = delete;
};
#define ENTER_ON(h, stream)                                \
    ENTER(h);                                              \
    StreamTurn turn_((h), (hipStream_t)(stream));          \
    if (turn_.err != hipSuccess) return fail(TDMPC2_ERR_HIP, "hipStreamWaitEvent (stream hand-over of the handle) failed: %s", hipGetErrorString(turn_.err))

int dev_alloc(tdmpc2_plan *h, void **p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
    h->allocs.push_back(*p);
    h->bytes += bytes;
    // TDMPC2_POISON=1 (tests): every allocation starts as NaN bit patterns instead of whatever the allocator hands out
    // (fresh pages read as zero and hide a read of memory that was never written; recycled memory does not)
#ifdef TDMPC2_TEST_HOOKS
    static const bool poison = getenv("TDMPC2_POISON") != nullptr;
    if (poison) HIP_TRY(hipMemset(*p, 0xFF, bytes ? bytes : 16));
#endif
    return 0;
}

// g_gemm_w's K-split workspaces (256 KiB per split tile and part, one workspace per chain), sized for what the handle's mode can
// use (every layered handle used to pay for 1 024 slots per chain -- 0.5 GiB -- whatever its mode): mode 0 none; mode 2
// (the default) only splits launches of at most cus / 2 tiles = at most cus / 16 tail tiles per XCD; mode 1 the last round of any
// launch = 32 tail tiles per XCD.  Called at create and when tdmpc2_plan_set_tuning raises the mode (grows, never shrinks; the
// smaller buffers stay with the handle until destroy).
int ksws_ensure(tdmpc2_plan *h) {
    Layered &L = h->lay;
    if (!L.ks_tiles || L.ksplit == 0) return 0;
    const size_t cus = (size_t)(h->num_cus > 0 ? h->num_cus : 256);
    const size_t cap_slots = L.ksplit == 2 ? 8 * ((cus / 16 + 3) / 4 * 4) * 4 : 8 * 32 * 4;
    const size_t want = std::min<size_t>(cap_slots, L.ks_tiles * 4);
    if (want <= L.ksws_slots) return 0;
    float *a = nullptr, *b = nullptr;
    int rc;
    if ((rc = dev_alloc(h, (void **)&a, want * 65536 * 4)) || (L.side && (rc = dev_alloc(h, (void **)&b, want * 65536 * 4)))) return rc;
    L.ksws = a; L.ksws2 = b; L.ksws_slots = want;
    return 0;
}

void dev_release(tdmpc2_plan *h, void *p) {  // free one allocation made with dev_alloc (tables that are re-grown)
    if (!p) return;
    for (size_t i = 0; i < h->allocs.size(); ++i)
        if (h->allocs[i] == p) {
            h->allocs.erase(h->allocs.begin() + (long)i);
            break;
        }
    (void)hipFree(p);
}

template <class NET> NET to_dev(const HostNet &n);
template <> NetS to_dev<NetS>(const HostNet &n) {  // split: hi/lo f16 packing + per-matrix scale; exact fp32: fp32 packing, scale 1
    NetS w;
    for (int i = 0; i < 3; ++i)
        w.l[i] = LayerS{n.l[i].wps ? n.l[i].wps : reinterpret_cast<const _Float16 *>(n.l[i].wp), n.l[i].bias, n.l[i].g, n.l[i].b,
                        n.l[i].oscale, n.l[i].ascale, n.l[i].KB, n.l[i].CT};
    return w;
}

HostNet *net_of(tdmpc2_plan *h, int net, int head) {
    switch (net) {
        case TDMPC2_NET_DYNAMICS: return &h->dyn;
        case TDMPC2_NET_REWARD: return &h->rew;
        case TDMPC2_NET_PI: return &h->pi;
        case TDMPC2_NET_Q: return &h->q[head];
        case TDMPC2_NET_TERMINATION: return &h->term;
        case TDMPC2_NET_TARGET_Q: return &h->tq[head];
    }
    return nullptr;
}

int check_ready(tdmpc2_plan *h) {
    for (int i = 0; i < 3; ++i) {
        if (!h->dyn.l[i].bound || !h->rew.l[i].bound || !h->pi.l[i].bound)
            return fail(TDMPC2_ERR_STATE, "weights of layer %d of dynamics/reward/pi are not bound", i);
        for (int qh = 0; qh < h->cfg.num_q; ++qh)
            if (!h->q[qh].l[i].bound) return fail(TDMPC2_ERR_STATE, "weights of Q head %d layer %d are not bound", qh, i);
        if (h->cfg.episodic && !h->term.l[i].bound)
            return fail(TDMPC2_ERR_STATE, "weights of layer %d of the termination head are not bound", i);
    }
    return 0;
}


// The fused kernels are instantiated per action padding (compile-time LDS strides) and arithmetic (AR 0 = f16x2 split,
// 1 = exact fp32 MFMA); ks_rollout / ks_pitraj also per workgroup geometry -- one translation unit per padding (k_fused.hip,
// k_cluster.hip), reached through the tables of launch.h.
#ifndef TDMPC2_DEFAULT_THROUGHPUT_ST
#define TDMPC2_DEFAULT_THROUGHPUT_ST 2  // sample tiles per workgroup when a call has enough plans to fill the chip
#endif
const FusedOps &fused_ops(int apad) {
#ifdef TDMPC2_ONLY_APAD  // experiment builds (tools/ablate.sh): one action padding only, a quarter of the compile time
#define TDK_OPS_CAT_(a, b) a##b
#define TDK_OPS_CAT(a, b) TDK_OPS_CAT_(a, b)
    (void)apad;
    return TDK_OPS_CAT(fused_ops_ap, TDMPC2_ONLY_APAD)();
#else
    switch (apad) {
        case 16: return fused_ops_ap16();
        case 32: return fused_ops_ap32();
        case 48: return fused_ops_ap48();
        default: return fused_ops_ap64();
    }
#endif
}
constexpr int CL2_SLOTS_HOST = 6 + 2 * MAXH;  // = CL2_SLOTS of cluster2_kernels.cuh (exchange tiles per cluster of ks_rollout_cl2)
const ClusterOps &cluster_ops(int apad) {
#ifdef TDMPC2_ONLY_APAD
    (void)apad;
    return TDK_OPS_CAT(cluster_ops_ap, TDMPC2_ONLY_APAD)();
#else
    switch (apad) {
        case 16: return cluster_ops_ap16();
        case 32: return cluster_ops_ap32();
        case 48: return cluster_ops_ap48();
        default: return cluster_ops_ap64();
    }
#endif
}
template <class NET> struct Kern;
template <> struct Kern<NetS> {
    // 32- or 64-row workgroups.  A 64-row workgroup reuses every weight fragment for two row tiles and is the
    // efficient one when the chip is full; a call with few plans is better served by twice as many 32-row workgroups.
    // Model: one workgroup per CU at a time, a round of 32-row workgroups takes 0.61 of a round of 64-row ones
    // (measured, c1: 0.34 vs 0.556 ms); pick the geometry with the shorter sum of rounds (E = 16: +30 %).
    static int sample_tiles(const tdmpc2_plan *h, int E, bool tracing) {
        if (tracing) return 2;  // the activation trace is laid out per 64-row tile
        if (h->force_rows) return h->force_rows / 32;
        const long cus = h->num_cus > 0 ? h->num_cus : 256;
        const long w2 = (long)E * h->tiles, w1 = 2 * w2;
        const long r2 = (w2 + cus - 1) / cus, r1 = (w1 + cus - 1) / cus;
        return 0.61 * (double)r1 < (double)r2 ? 1 : TDMPC2_DEFAULT_THROUGHPUT_ST;
    }
    // Always 8 wavefronts per workgroup.  A 4-wave, 32-row geometry (two workgroups per CU, so that one's VALU epilogue
    // overlaps the other's MFMA k-loop; the device code is templated for it: CtxT<APAD, 1, 4>) was measured and lost
    // 5.83 vs 4.88 ms per launch: each weight fragment then feeds one row tile, the k-loop needs 85 B/clk/CU of
    // fragment loads and becomes L1-bound.
    static int waves(const tdmpc2_plan *, int, int) { return 8; }
    static void setup(const tdmpc2_plan *h, const SetupParamsT<NetS> &p, int E, hipStream_t st) {
        fused_ops(h->Apad).setup(h->split ? 0 : 1, p, E, h->lds_bytes, st);
    }
    static void pitraj(const tdmpc2_plan *h, const PiTrajParamsT<NetS> &p, int E, hipStream_t st) {
        // one 32-row tile holds the policy-prior trajectories when P <= 32 (the reference uses 24)
        const int nst = p.P <= 32 ? 1 : 2;
        const size_t lds = nst == 1 ? h->lds_bytes - (size_t)32 * h->row_bytes : h->lds_bytes;
        fused_ops(h->Apad).pitraj(h->split ? 0 : 1, nst, p, E, lds, st);
    }
    static void rollout(const tdmpc2_plan *h, const RolloutParamsT<NetS> &p, int grid, hipStream_t st, int nst, int nw) {
        (void)nw;
        const bool tracing = p.trace_tiles || p.trace_scalars;  // (the host forces 64-row workgroups for a trace call)
        const size_t lds = nst == 2 ? h->lds_bytes : h->lds_bytes - (size_t)32 * h->row_bytes;
        fused_ops(h->Apad).rollout(h->split ? 0 : 1, nst, h->cfg.episodic != 0, tracing, p, grid, lds, st);
    }
    // cluster path: `clusters` row tiles of 32 samples, 8 workgroups each, in groups of 8 clusters (one per XCD)
    static void rollout_cluster(const tdmpc2_plan *h, const RolloutParamsT<NetS> &p, int clusters, hipStream_t st) {
        cluster_ops(h->Apad).rollout_cl(h->cfg.episodic != 0, p, (clusters + 7) / 8 * 64, h->cl_lds, st);
    }
};

// Cluster launches of different streams are serialised per device: a launch whose workgroups are only partly resident
// spins on members that wait for a CU, and two such launches could wait for each other.  (Other kernels may share the
// device: they finish without the cluster's help.)  The lock covers [wait for the previous plan's event, this plan's
// launches, record]; nothing here blocks the host on the GPU.
struct ClusterGate {
    std::mutex mu;
    hipEvent_t ev[64] = {};
    bool have[64] = {};
    hipStream_t last[64] = {};
};
ClusterGate g_cluster_gate;

template <class NET>
int launch_setup(tdmpc2_plan *h, int E, const float *z0, const float *task_emb, const float *prev_mean,
                 const unsigned char *t0, hipStream_t st, bool skip_cvec = false) {
    SetupParamsT<NET> p{};
    p.E = E; p.H = h->cfg.horizon; p.A = h->cfg.action_dim; p.T = h->cfg.task_dim; p.multitask = h->cfg.multitask;
    p.nq = h->cfg.num_q; p.nnets = h->nnets; p.stride = h->stride; p.max_std = h->cfg.max_std;
    p.dyn = to_dev<NET>(h->dyn); p.rew = to_dev<NET>(h->rew); p.pi = to_dev<NET>(h->pi);
    for (int i = 0; i < h->cfg.num_q; ++i) p.q[i] = to_dev<NET>(h->q[i]);
    p.wemb[BE_DYN] = h->dyn.l[0].wemb; p.wemb[BE_REW] = h->rew.l[0].wemb; p.wemb[BE_PI] = h->pi.l[0].wemb;
    for (int i = 0; i < h->cfg.num_q; ++i) p.wemb[BE_Q0 + i] = h->q[i].l[0].wemb;
    p.z0 = z0; p.task_emb = task_emb; p.prev_mean = prev_mean; p.t0 = t0;
    p.beff = h->beff; p.cvec = h->cvec; p.mean = h->mean; p.std = h->std;
    // cluster path: the arrival words of this call's clusters start every plan at zero (phase numbers grow through its launches)
    p.cl_flags = (h->cl_max_clusters && (long)E * h->tiles * 2 <= h->cl_max_clusters) ? h->cl_flags : nullptr;
    p.cl_flag_words = h->tiles * 2 * CL_FLAG_STRIDE;
    p.cl2_flags = (p.cl_flags && E == 1 && h->cl2_flags && h->cl2_mode && h->cluster_mode == 2) ? h->cl2_flags : nullptr;
    p.cl2_flag_words = h->tiles * 4 * CL_FLAG_STRIDE;
    p.skip_cvec = skip_cvec ? 1 : 0;
    p.err_clear = h->cl_err_dev;  // word 0 of the error line back to zero, in stream order (fault_fresh)
    Kern<NET>::setup(h, p, E, st);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class NET>
void fill_rollout(tdmpc2_plan *h, RolloutParamsT<NET> &p, int E) {
    const tdmpc2_plan_cfg &c = h->cfg;
    p.E = E; p.N = c.num_samples; p.H = c.horizon; p.A = c.action_dim; p.Apad = h->Apad; p.P = c.num_pi_trajs;
    p.stride = h->stride; p.tiles = h->tiles; p.nq = c.num_q; p.num_bins = c.num_bins; p.multitask = c.multitask;
    p.nnets = h->nnets; p.log_std_min = c.log_std_min; p.log_std_dif = c.log_std_dif;
    p.dyn = to_dev<NET>(h->dyn); p.rew = to_dev<NET>(h->rew); p.pi = to_dev<NET>(h->pi);
    if (c.episodic) p.term = to_dev<NET>(h->term);
    for (int i = 0; i < c.num_q; ++i) p.q[i] = to_dev<NET>(h->q[i]);
    p.bins = h->bins; p.beff = h->beff; p.cvec = h->cvec; p.mean = h->mean; p.std = h->std;
    p.actions = h->actions; p.value = h->value; p.zscratch = h->zscratch;
    p.ticket = h->ticket; p.fold_refit = 0;
    p.iters_total = c.iterations;
    p.timing = h->timing;
}

// A bounded inter-workgroup wait gave up (the handle's error word is set): the plan / call in flight returned NaN.  Switch to
// the paths without such waits, remember when, and let fault_clean() switch back after `rearm_after` clean calls.
// What the kernels run = what the caller asked for (user_*: environment at create, tdmpc2_plan_set_tuning) unless the handle is
// downgraded (a reported wait) or asked to plan once without inter-workgroup waits (TDMPC2_TUNE_SAFE_ONCE: the re-plan of a
// sharded plan).  The one place that writes cluster_mode / lay.fuse_ln after create.
void apply_modes(tdmpc2_plan *h) {
    const bool safe = h->degraded || h->safe_once;
    h->cluster_mode = safe ? 0 : h->user_cluster_mode;
    h->lay.fuse_ln = safe ? false : h->user_fuse_ln;
}
void fault_note(tdmpc2_plan *h) {
    if (h->degraded && h->rearm_after > 0 && h->rearm_after < 4096)
        h->rearm_after *= 2;  // (a fault on the downgraded paths cannot happen; this branch is a fault right after a re-arm raced in)
    h->degraded = true;
    apply_modes(h);
    h->clean_calls = 0;
    h->faults++;
    h->faults_total++;
    h->last_fault = std::chrono::steady_clock::now();
}
void fault_clean(tdmpc2_plan *h) {  // a call is about to be enqueued and no fault is pending
    if (!h->degraded) {  // a long clean run on the fast paths forgets the back-off
        if (h->rearm_after > h->rearm_base && ++h->clean_calls >= 16 * h->rearm_after) {
            h->rearm_after = h->rearm_base;
            h->clean_calls = 0;
        }
        return;
    }
    if (h->rearm_after <= 0 || h->in_shard) return;
    if (h->clean_calls < h->rearm_after) {  // this call still runs on the downgraded paths: rearm_after of them in all
        ++h->clean_calls;
        return;
    }
    h->degraded = false;
    apply_modes(h);
    h->clean_calls = 0;
    h->rearms++;
    if (h->rearm_after < 4096) h->rearm_after *= 2;  // the next downgrade lasts twice as long; a long clean run resets it (above)
}

// The host's look at the sticky word (common.cuh: raise_fault).  It never touches word 0 -- the verdict of the call in flight,
// which that call's own last kernel reads and the NEXT call clears in stream order (fault_fresh): calls of one handle may be
// pipelined without a sync and each still gets its own verdict.
bool fault_poll(tdmpc2_plan *h) {
    unsigned int *w = h->cl_err_host;
    // read-and-clear in ONE step: a device store that lands between a separate read and clear would be lost
    if (!w || !__atomic_exchange_n(w + 8, 0u, __ATOMIC_RELAXED)) return false;
    if (getenv("TDMPC2_DEBUG_FAULT")) fprintf(stderr, "[tdmpc2_plan] a bounded wait gave up: code 0x%08x (kind = code & 15)\n", w[0]);
    fault_note(h);
    return true;
}
// Start of a call that stands for itself (a plan, an estimate_value, a td_target ...; NOT the later calls of a sharded plan, whose
// final pick must still see a wait that gave up in its first iteration): word 0 back to zero, in stream order.  The fused
// family's ks_setup does it itself (SetupParamsT::err_clear).
int fault_fresh(tdmpc2_plan *h, hipStream_t st) {
    if (h->cl_err_dev) HIP_TRY(hipMemsetAsync(h->cl_err_dev, 0, 4, st));
    return 0;
}

int validate_envs(tdmpc2_plan *h, int E) {
    if (!h) return fail(TDMPC2_ERR_INVALID, "null handle");
    if (E < 1 || E > h->cfg.max_envs) return fail(TDMPC2_ERR_INVALID, "n_envs=%d outside [1, max_envs=%d]", E, h->cfg.max_envs);
    // a cluster hand-over / fused-epilogue wait of an EARLIER call gave up (bounded wait): that plan returned NaN actions and kept
    // its prev_mean (refit_plan), that td_target / policy_value returned NaN; the handle now runs the paths without waits
    // until fault_clean() re-arms the fast ones.  tdmpc2_plan_take_fault / tdmpc2_plan_fault_info report it.
    if (!fault_poll(h)) fault_clean(h);
    return check_ready(h);
}


// Everything of TDMPC2._plan after encode() (tdmpc2/tdmpc2.py:154-206) on the fused 512-wide path.
template <class NET>
int fused_run(tdmpc2_plan *h, hipStream_t st, int E, const float *z0, const float *task_emb, const float *act_mask,
              const float *disc_pow, float *prev_mean, const uint8_t *t0, int eval_mode, const tdmpc2_noise *tape,
              uint64_t seed, float *action, const tdmpc2_debug *dbg) {
    const tdmpc2_plan_cfg &c = h->cfg;
    const int H = c.horizon, N = c.num_samples, A = c.action_dim, K = c.num_elites, P = c.num_pi_trajs, I = c.iterations;
    const unsigned call = h->call++;
    int rc;
    // single-plan latency: 8 workgroups per 32-row tile when the whole call then still fits the chip in one round
    const long clusters = (long)E * h->tiles * 2;
    const bool cluster = h->cluster_mode != 0 && h->cl_max_clusters > 0 && clusters <= h->cl_max_clusters &&
                         (clusters + 7) / 8 * 64 <= (h->num_cus > 0 ? h->num_cus : 256);
    // ... which also computes the policy-prior trajectories (cluster 0 of each plan, first launch) and needs no z0 products
    const bool pi_fold = cluster && P > 0 && P <= 32;
    if ((rc = launch_setup<NET>(h, E, z0, task_emb, prev_mean, t0, st, cluster))) return rc;
    if (P > 0 && !pi_fold) {
        PiTrajParamsT<NET> p{};
        p.E = E; p.N = N; p.H = H; p.A = A; p.Apad = h->Apad; p.P = P; p.stride = h->stride; p.multitask = c.multitask;
        p.nnets = h->nnets; p.log_std_min = c.log_std_min; p.log_std_dif = c.log_std_dif;
        p.dyn = to_dev<NET>(h->dyn); p.pi = to_dev<NET>(h->pi);
        p.z0 = z0; p.beff = h->beff; p.act_mask = act_mask; p.pi_traj_eps = tape ? tape->pi_traj_eps : nullptr;
        p.seed = seed; p.call = call; p.actions = h->actions; p.zscratch = h->zscratch;
        p.zscratch_estride = (long)h->tiles * ROWS * WIDTH;
        Kern<NET>::pitraj(h, p, E, st);
        HIP_TRY(hipGetLastError());
    }
    RolloutParamsT<NET> rp{};
    fill_rollout<NET>(h, rp, E);
    rp.z0 = z0; rp.act_mask = act_mask; rp.disc_pow = disc_pow; rp.seed = seed; rp.call = call; rp.given_actions = 0;
    rp.pi_fold = pi_fold ? 1 : 0;
    rp.pi_traj_eps = tape ? tape->pi_traj_eps : nullptr;
    const int nst = cluster ? 1 : Kern<NET>::sample_tiles(h, E, false), nw = Kern<NET>::waves(h, E, nst);
    rp.tiles = h->tiles * (2 / nst);
    rp.cl_xbuf = h->cl_xbuf; rp.cl_flags = h->cl_flags; rp.cl_err = h->cl_err_dev; rp.cl_zs = h->cl_zs;
    rp.cl_fault = h->cl_fault;
    rp.cl2_xbuf = h->cl2_xbuf; rp.cl2_flags = h->cl2_flags; rp.cl2_zs = h->cl2_zs; rp.cl2_mail = h->cl2_mail;
    // a single non-episodic plan: from the second launch on the reward chain runs beside the dynamics chain on a second cluster
    // per tile (ks_rollout_cl2: all 256 CUs)
    const bool cluster2 = cluster && h->cluster_mode == 2 && E == 1 && h->cl2_xbuf && h->cl2_mode && !c.episodic && !h->cl_fault;
    std::unique_lock<std::mutex> gate;
    bool gate_record = false;
    const int gdev = c.device >= 0 && c.device < 64 ? c.device : 0;
    if (cluster) {
        gate = std::unique_lock<std::mutex>(g_cluster_gate.mu);
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (cap == hipStreamCaptureStatusNone) {  // (a captured plan replays on one stream: ordered by the graph itself)
            if (!g_cluster_gate.have[gdev]) {
                HIP_TRY(hipEventCreateWithFlags(&g_cluster_gate.ev[gdev], hipEventDisableTiming));
                g_cluster_gate.have[gdev] = true;
                g_cluster_gate.last[gdev] = st;
            } else if (g_cluster_gate.last[gdev] != st) {
                HIP_TRY(hipStreamWaitEvent(st, g_cluster_gate.ev[gdev], 0));
            }
            gate_record = true;
        }
    }
    // elite selection + refit: inside the rollout launch (last workgroup of each plan, LDS budget = the 32-row tile) or as
    // a launch of its own (TDMPC2_TUNE_FOLD_REFIT 0)
    int refit_stage = 0;
    size_t refit_lds = refit_lds_bytes(N, K, H, A, &refit_stage, (size_t)32 * h->row_bytes);
    // the in-launch refit stages the re-derived elite actions in the (then idle) tile memory; if they do not fit, or the
    // caller wants the per-iteration action dump, the refit runs as a launch of its own
    // auto: only when the whole launch is one round of workgroups (few plans: latency).  With several rounds every round
    // ends with the refits of the plans that completed in it, on CUs whose next workgroup then starts late: measured
    // +0.5 ms on the 4.4 ms launch of 256 plans, against 30 us for the separate k_refit launch.
    const bool one_round = cluster || (long)E * rp.tiles <= (h->num_cus > 0 ? h->num_cus : 256);
    const bool fold = refit_stage && (h->fold_refit == 1 || (h->fold_refit == 2 && one_round));
    if (!fold) refit_lds = refit_lds_bytes(N, K, H, A, &refit_stage);
    for (int it = 0; it < I; ++it) {
        rp.iter = it;
        if (tape) {
            rp.sample_eps = tape->sample_eps + (size_t)it * H * (N - P) * A;
            rp.sample_eps_estride = (long)I * H * (N - P) * A;
            rp.pi_eps = tape->pi_eps + (size_t)it * N * A;
            rp.pi_eps_estride = (long)I * N * A;
            rp.qidx = tape->qidx + (size_t)it * 2;
            rp.qidx_estride = (long)I * 2;
        }
        RefitParams &fp = rp.rf;
        fp = RefitParams{};
        fp.Nvalid = c.num_valid_samples; fp.E = E; fp.N = N; fp.H = H; fp.A = A; fp.K = K; fp.iter = it; fp.last = (it == I - 1); fp.eval_mode = eval_mode; fp.stage = refit_stage;
        fp.temperature = c.temperature; fp.min_std = c.min_std; fp.max_std = c.max_std;
        fp.value = h->value; fp.actions = h->actions; fp.act_mask = act_mask; fp.mean = h->mean; fp.std = h->std;
        fp.gumbel_exp = tape ? tape->gumbel_exp : nullptr; fp.final_eps = tape ? tape->final_eps : nullptr;
        fp.seed = seed; fp.call = call; fp.prev_mean = prev_mean; fp.action = action;
        fp.err = cluster ? h->cl_err_dev : nullptr;
        if (dbg) {
            if (dbg->value) { fp.dbg_value = dbg->value + (size_t)it * N; fp.dbg_value_es = (long)I * N; }
            if (dbg->elite_idx) { fp.dbg_idx = dbg->elite_idx + (size_t)it * K; fp.dbg_idx_es = (long)I * K; }
            if (dbg->score) { fp.dbg_score = dbg->score + (size_t)it * K; fp.dbg_score_es = (long)I * K; }
            if (dbg->mean) { fp.dbg_mean = dbg->mean + (size_t)it * H * A; fp.dbg_mean_es = (long)I * H * A; }
            if (dbg->std) { fp.dbg_std = dbg->std + (size_t)it * H * A; fp.dbg_std_es = (long)I * H * A; }
        }
        rp.fold_refit = fold ? 1 : 0;
        if (fold) {
            fp.regen = 1; fp.P = P; fp.Apad = h->Apad;
            fp.sample_eps = rp.sample_eps; fp.sample_eps_estride = rp.sample_eps_estride;
        }
        if (h->profiling && h->ev_used + 2 <= (int)h->ev.size()) HIP_TRY(hipEventRecord(h->ev[h->ev_used], st));
        if (cluster2) cluster_ops(h->Apad).rollout_cl2(rp, (int)((2 * clusters + 7) / 8 * 64), h->cl_lds, st);
        else if (cluster) Kern<NET>::rollout_cluster(h, rp, (int)clusters, st);
        else Kern<NET>::rollout(h, rp, E * rp.tiles, st, nst, nw);
        HIP_TRY(hipGetLastError());
        if (h->profiling && h->ev_used + 2 <= (int)h->ev.size()) {
            HIP_TRY(hipEventRecord(h->ev[h->ev_used + 1], st));
            h->ev_used += 2;
        }
        if (!fold) {
            hipLaunchKernelGGL(k_refit, dim3(E), dim3(refit_threads(N)), refit_lds, st, fp);
            HIP_TRY(hipGetLastError());
        }
        // the per-iteration action dump reads h->actions after the refit (which does not write them)
        if (dbg && dbg->actions)
            HIP_TRY(hipMemcpy2DAsync(dbg->actions + (size_t)it * H * N * A, (size_t)I * H * N * A * 4, h->actions,
                                     (size_t)H * N * A * 4, (size_t)H * N * A * 4, E, hipMemcpyDeviceToDevice, st));
    }
    if (gate_record) {
        HIP_TRY(hipEventRecord(g_cluster_gate.ev[gdev], st));
        g_cluster_gate.last[gdev] = st;
    }
    return TDMPC2_OK;
}

// TDMPC2._estimate_value (tdmpc2/tdmpc2.py:122-136) on given action sequences, fused path.
template <class NET>
int fused_estimate_value(tdmpc2_plan *h, hipStream_t st, int E, const float *z0, const float *task_emb,
                         const float *act_mask, const float *disc_pow, const float *actions, const float *pi_eps,
                         const int32_t *qidx, float *value, float *trace_tiles, float *trace_scalars) {
    const tdmpc2_plan_cfg &c = h->cfg;
    const int N = c.num_samples, A = c.action_dim;
    int rc;
    // setup needs prev_mean / t0 only for mean/std init, which this entry does not use: feed dummies
    HIP_TRY(hipMemsetAsync(h->mean, 0, (size_t)E * c.horizon * A * 4, st));
    HIP_TRY(hipMemsetAsync(h->value, 1, (size_t)E, st));  // E bytes of ones used as t0 = 1 flags (no warm start read)
    if ((rc = launch_setup<NET>(h, E, z0, task_emb, h->mean, reinterpret_cast<const unsigned char *>(h->value), st))) return rc;
    RolloutParamsT<NET> rp{};
    fill_rollout<NET>(h, rp, E);
    rp.z0 = z0; rp.act_mask = act_mask; rp.disc_pow = disc_pow; rp.given_actions = 1; rp.iter = 0;
    rp.actions = const_cast<float *>(actions);
    rp.value = value;
    rp.pi_eps = pi_eps; rp.pi_eps_estride = (long)N * A;
    rp.qidx = qidx; rp.qidx_estride = 2;
    rp.trace_tiles = trace_tiles; rp.trace_scalars = trace_scalars;
    const int nst = Kern<NET>::sample_tiles(h, E, trace_tiles != nullptr || trace_scalars != nullptr), nw = Kern<NET>::waves(h, E, nst);
    rp.tiles = h->tiles * (2 / nst);
    Kern<NET>::rollout(h, rp, E * rp.tiles, st, nst, nw);
    HIP_TRY(hipGetLastError());
    return TDMPC2_OK;
}

}  // namespace

namespace {
int run_impl(tdmpc2_plan *h, int n_envs, const float *z0, const float *task_emb, const float *act_mask, const float *disc_pow,
             float *prev_mean, const uint8_t *t0, int eval_mode, const tdmpc2_noise *tape, uint64_t seed, float *action,
             const tdmpc2_debug *dbg, void *stream);
}

extern "C" {

int tdmpc2_plan_abi_version(void) { return TDMPC2_PLAN_ABI_VERSION; }
const char *tdmpc2_last_error(void) { return g_err.c_str(); }

int tdmpc2_plan_create(const tdmpc2_plan_cfg *cfg, tdmpc2_plan_t **out) {
    if (!cfg || !out) return fail(TDMPC2_ERR_INVALID, "null argument");
    *out = nullptr;
    const tdmpc2_plan_cfg &c = *cfg;
    // ---- limits common to both kernel families
    if (c.action_dim < 1 || c.action_dim > 64) return fail(TDMPC2_ERR_UNSUPPORTED, "action_dim %d outside [1, 64]", c.action_dim);
    // num_bins 0 / 1: the reference's regression heads -- one output column, two_hot_inv = identity / symexp (math.py:76-79)
    if (c.num_bins < 0 || c.num_bins > 128) return fail(TDMPC2_ERR_UNSUPPORTED, "num_bins %d outside [0, 128]", c.num_bins);
    if (c.num_q < 2 || c.num_q > MAXQ) return fail(TDMPC2_ERR_UNSUPPORTED, "num_q %d outside [2, %d]", c.num_q, MAXQ);
    if (c.horizon < 1 || c.horizon > MAXH) return fail(TDMPC2_ERR_UNSUPPORTED, "horizon %d outside [1, %d]", c.horizon, MAXH);
    if (c.num_samples % ROWS != 0 || c.num_samples < ROWS || c.num_samples > 1024)
        return fail(TDMPC2_ERR_UNSUPPORTED, "num_samples %d must be a multiple of %d in [%d, 1024]", c.num_samples, ROWS, ROWS);
    if (c.num_pi_trajs < 0 || c.num_pi_trajs > ROWS || c.num_pi_trajs >= c.num_samples)
        return fail(TDMPC2_ERR_UNSUPPORTED, "num_pi_trajs %d outside [0, %d]", c.num_pi_trajs, ROWS);
    if (c.num_elites < 1 || c.num_elites > c.num_samples) return fail(TDMPC2_ERR_INVALID, "num_elites %d", c.num_elites);
    if (c.num_valid_samples != 0 && (c.num_valid_samples < c.num_elites || c.num_valid_samples > c.num_samples ||
                                     c.num_valid_samples <= c.num_pi_trajs || c.num_samples - c.num_valid_samples >= GBM))
        return fail(TDMPC2_ERR_INVALID, "num_valid_samples %d: 0, or the true sample count behind num_samples %d rounded up to the row tile "
                    "(>= num_elites %d, > num_pi_trajs %d)", c.num_valid_samples, c.num_samples, c.num_elites, c.num_pi_trajs);
    if (c.simnorm_dim != 8) return fail(TDMPC2_ERR_UNSUPPORTED, "simnorm_dim %d (kernels are built for 8)", c.simnorm_dim);
    if (c.multitask && c.task_dim < 1) return fail(TDMPC2_ERR_INVALID, "multitask needs task_dim > 0");
    if (c.multitask && c.episodic)  // the reference asserts the same: tdmpc2/common/world_model.py:136
        return fail(TDMPC2_ERR_UNSUPPORTED, "termination head with task ids is not supported (reference world_model.py:136)");
    if (c.iterations < 1 || c.max_envs < 1) return fail(TDMPC2_ERR_INVALID, "iterations / max_envs must be positive");
    if (c.latent_dim < 8 || c.mlp_dim < 8 || c.latent_dim % 8 != 0)
        return fail(TDMPC2_ERR_UNSUPPORTED, "latent_dim %d / mlp_dim %d", c.latent_dim, c.mlp_dim);
    // ---- kernel family
    const bool fits_fused = c.latent_dim == WIDTH && c.mlp_dim == WIDTH;
    const bool fits_layered = c.latent_dim % 32 == 0 && c.mlp_dim % 32 == 0 && c.num_samples % GBM == 0;
    int path = c.path;
    if (path == TDMPC2_PATH_AUTO) path = fits_fused ? TDMPC2_PATH_FUSED : TDMPC2_PATH_LAYERED;
    if (path == TDMPC2_PATH_FUSED && !fits_fused)
        return fail(TDMPC2_ERR_UNSUPPORTED, "fused planner kernels are built for latent_dim == mlp_dim == %d (got %d / %d)",
                    WIDTH, c.latent_dim, c.mlp_dim);
    if (path == TDMPC2_PATH_LAYERED && !fits_layered)
        return fail(TDMPC2_ERR_UNSUPPORTED, "layered planner kernels need latent_dim %% 32 == 0, mlp_dim %% 32 == 0 and "
                    "num_samples %% %d == 0 (got %d / %d / %d)", GBM, c.latent_dim, c.mlp_dim, c.num_samples);
    if (path != TDMPC2_PATH_FUSED && path != TDMPC2_PATH_LAYERED) return fail(TDMPC2_ERR_INVALID, "unknown path %d", c.path);
    int prec = c.precision;
    if (prec == TDMPC2_PREC_AUTO) prec = TDMPC2_PREC_SPLIT_F16;
    if (prec != TDMPC2_PREC_FP32 && prec != TDMPC2_PREC_SPLIT_F16) return fail(TDMPC2_ERR_INVALID, "unknown precision %d", c.precision);
    if (prec == TDMPC2_PREC_SPLIT_F16 && path == TDMPC2_PATH_LAYERED && (c.mlp_dim > 4096 || c.latent_dim > 4096))
        return fail(TDMPC2_ERR_UNSUPPORTED, "the layered f16x2-split row kernels hold a row of at most 4096 columns in registers");
    DevGuard dev_(c.device);
    if (!dev_.ok) return fail(TDMPC2_ERR_HIP, "hipSetDevice(%d) failed", c.device);

    tdmpc2_plan *h = new (std::nothrow) tdmpc2_plan();
    if (!h) return fail(TDMPC2_ERR_INVALID, "out of host memory");
    h->cfg = c;
    h->cfg.path = path;
    h->cfg.precision = prec;
    h->lay.on = (path == TDMPC2_PATH_LAYERED);
    h->lay.qarr = h->q;
    h->split = (prec == TDMPC2_PREC_SPLIT_F16);
    if (hipDeviceGetAttribute(&h->num_cus, hipDeviceAttributeMultiprocessorCount, c.device) != hipSuccess) h->num_cus = 0;
    h->Apad = (c.action_dim + 15) / 16 * 16;  // the fused kernels are instantiated for action paddings 16 / 32 / 48 / 64
    h->tiles = c.num_samples / ROWS;
    h->nnets = BE_Q0 + c.num_q;
    int rc = 0;
    const size_t E = c.max_envs, H = c.horizon, N = c.num_samples, A = c.action_dim;
    if (!h->lay.on) {
        if (h->split) {
            // operand form: row = [hi: SH halfs | lo: SH halfs | 8 pad]; row stride in dwords SH + 4 = 4 x odd
            const int SH = WIDTH + h->Apad;
            h->stride = 2 * SH + 8;  // in halfs
            h->row_bytes = (size_t)h->stride * 2;
        } else {
            // fp32 rows [z (512) | a (Apad) | 4 pad]: stride / 4 odd -> conflict-free ds_read_b128 across 16 rows
            h->stride = WIDTH + h->Apad + 4;  // in floats
            h->row_bytes = (size_t)h->stride * 4;
        }
        // after the tile: LayerNorm partials, LayerNorm affine (CtxT::gb), then mean / std [2 H A] of the rollout kernel or
        // the 64 row -> task entries of ks_value
        const size_t tail = std::max<size_t>((size_t)2 * c.horizon * c.action_dim * 4, 256);
        h->lds_bytes = (size_t)ROWS * h->row_bytes + 4096 /* LayerNorm partials */ + 4096 /* LayerNorm affine */ + tail + 64;
        if (h->lds_bytes > 160 * 1024) {
            const size_t need = h->lds_bytes;
            delete h;
            return fail(TDMPC2_ERR_UNSUPPORTED, "LDS tile of %zu bytes exceeds 160 KiB", need);
        }
    }
    // torch.linspace(vmin, vmax, num_bins) in fp32 (math.py:80): float step, product rounded once
    std::vector<float> bins(c.num_bins > 1 ? c.num_bins : 2, 0.f);
    if (c.num_bins > 1)
    {
        const float step = (c.vmax - c.vmin) / (float)(c.num_bins - 1);
        for (int i = 0; i < c.num_bins; ++i)
            bins[i] = (i < c.num_bins / 2) ? (float)((double)c.vmin + (double)step * i)
                                           : (float)((double)c.vmax - (double)step * (c.num_bins - 1 - i));
    }
    if ((rc = dev_alloc(h, (void **)&h->bins, bins.size() * 4)) ||
        (rc = dev_alloc(h, (void **)&h->actions, E * H * N * A * 4)) ||
        (rc = dev_alloc(h, (void **)&h->value, E * N * 4)) ||
        (rc = dev_alloc(h, (void **)&h->mean, E * H * A * 4)) ||
        (rc = dev_alloc(h, (void **)&h->std, E * H * A * 4)) ||
        (rc = dev_alloc(h, (void **)&h->ticket, E * 4)) || (rc = dev_alloc(h, (void **)&h->qidx_buf, E * 2 * 4))) {
        tdmpc2_plan_destroy(h);
        return rc;
    }
    if (hipMemset(h->ticket, 0, E * 4) != hipSuccess) {
        tdmpc2_plan_destroy(h);
        return fail(TDMPC2_ERR_HIP, "hipMemset(ticket) failed");
    }
    if (!h->split) {  // unit scale of the exact arithmetic (LayerS::oscale / ascale)
        const float one = 1.f;
        if ((rc = dev_alloc(h, (void **)&h->one, 4)) || hipMemcpy(h->one, &one, 4, hipMemcpyHostToDevice) != hipSuccess) {
            tdmpc2_plan_destroy(h);
            return fail(TDMPC2_ERR_HIP, "allocating the unit output scale failed");
        }
    }
    if (!h->lay.on) {
        if ((rc = dev_alloc(h, (void **)&h->cvec, E * 2 * WIDTH * 4)) ||
            (rc = dev_alloc(h, (void **)&h->beff, E * h->nnets * WIDTH * 4)) ||
            (rc = dev_alloc(h, (void **)&h->zscratch, E * h->tiles * ROWS * WIDTH * 4))) {
            tdmpc2_plan_destroy(h);
            return rc;
        }
    } else {
        Layered &L = h->lay;
        L.Kin = (int)round_up((size_t)c.latent_dim + A, GBK);
        L.Mp = c.mlp_dim;
        L.ldl = (int)round_up((size_t)(c.num_bins > 2 * c.action_dim ? c.num_bins : 2 * c.action_dim), 32);
        L.Ppad = (int)round_up((size_t)(c.num_pi_trajs > 0 ? c.num_pi_trajs : 1), 32);
        const size_t Rp = round_up(E * N, GBM);  // the policy-prior pass (E * Ppad rows) reuses the same buffers
        // beff is read per ROW of a GEMM tile; tiles of the policy-prior pass may run past the last plan: + 4 plans
        if ((rc = dev_alloc(h, (void **)&L.X, Rp * L.Kin * 4)) || (rc = dev_alloc(h, (void **)&L.HA, Rp * L.Mp * 4)) ||
            (rc = dev_alloc(h, (void **)&L.HB, Rp * L.Mp * 4)) || (rc = dev_alloc(h, (void **)&L.LG, Rp * L.ldl * 4)) ||
            (rc = dev_alloc(h, (void **)&L.G, Rp * 4)) || (rc = dev_alloc(h, (void **)&L.QT, Rp * 4)) ||
            (rc = dev_alloc(h, (void **)&L.TERM, Rp * 4)) || (rc = dev_alloc(h, (void **)&L.qidx, E * 2 * 4)) ||
            (rc = dev_alloc(h, (void **)&h->beff, (round_up(E, GBM) + 4) * h->nnets * L.Mp * 4))) {
            tdmpc2_plan_destroy(h);
            return rc;
        }
        L.bias_tab = h->beff;
        if (h->split) {  // lay_cvec: z0 rows in operand form (one per plan, padded to a GEMM tile), cvec [2][rows][Mp]
            L.cvec_rows = round_up(E, GBM);
            if ((rc = dev_alloc(h, (void **)&L.Z0X, L.cvec_rows * L.Kin * 4)) || (rc = dev_alloc(h, (void **)&L.cvec, 2 * L.cvec_rows * L.Mp * 4))) {
                tdmpc2_plan_destroy(h);
                return rc;
            }
            if (hipMemset(L.Z0X, 0, L.cvec_rows * L.Kin * 4) != hipSuccess) {
                tdmpc2_plan_destroy(h);
                return fail(TDMPC2_ERR_HIP, "hipMemset(workspace) failed");
            }
        }
        // second chain (reward || dynamics, Q head || Q head): buffers, stream, events
        if (!getenv("TDMPC2_ONE_STREAM")) {
            if ((rc = dev_alloc(h, (void **)&L.HA2, Rp * L.Mp * 4)) || (rc = dev_alloc(h, (void **)&L.HB2, Rp * L.Mp * 4)) ||
                (rc = dev_alloc(h, (void **)&L.LG2, Rp * L.ldl * 4))) {
                tdmpc2_plan_destroy(h);
                return rc;
            }
            SideRes sr;
            if (hipMemset(L.HA2, 0, Rp * L.Mp * 4) != hipSuccess || hipMemset(L.HB2, 0, Rp * L.Mp * 4) != hipSuccess ||
                !side_acquire(c.device, &sr)) {
                tdmpc2_plan_destroy(h);
                return fail(TDMPC2_ERR_HIP, "creating the second stream of the layered path failed");
            }
            L.side = sr.stream; L.ev_fork = sr.ev[0]; L.ev_side = sr.ev[1]; L.ev_xread = sr.ev[2];
        }
        if (h->split) {  // fused NormedLinear epilogue: exchange buffer, counters, error word
            const size_t maxct = (size_t)(std::max(c.mlp_dim, c.latent_dim) + 31) / 32;
            L.stats_cap = Rp * ((maxct + 3) / 4) * 2;
            L.arrive_cap = 64 * (Rp / 32);
            L.arrive_cap *= 2;  // two chains
            if ((rc = dev_alloc(h, (void **)&L.stats, L.stats_cap * 4)) || (rc = dev_alloc(h, (void **)&L.stats2, L.stats_cap * 4)) ||
                (rc = dev_alloc(h, (void **)&L.arrive, L.arrive_cap * 4))) {
                tdmpc2_plan_destroy(h);
                return rc;
            }
            if (hipMemset(L.arrive, 0, L.arrive_cap * 4) != hipSuccess ||
                hipHostMalloc((void **)&h->cl_err_host, 64, hipHostMallocMapped) != hipSuccess) {
                tdmpc2_plan_destroy(h);
                return fail(TDMPC2_ERR_HIP, "allocating the fused-epilogue counters / error word failed");
            }
            memset(h->cl_err_host, 0, 64);  // hipHostMalloc does not zero: word 0 (verdict) AND word 8 (sticky, fault_poll)
            if (hipHostGetDevicePointer((void **)&h->cl_err_dev, h->cl_err_host, 0) != hipSuccess) {
                tdmpc2_plan_destroy(h);
                return fail(TDMPC2_ERR_HIP, "hipHostGetDevicePointer failed");
            }
            L.fuse_ln = true;
            if (const char *fl = getenv("TDMPC2_FUSE_LN")) L.fuse_ln = atoi(fl) != 0;
            // g_gemm_w's K-split tail: 256 KiB per (split tile, part); at most 32 tail tiles per XCD x 4 parts, or all the tiles
            // of the handle's largest call (one workspace per chain)
            if (const char *ks = getenv("TDMPC2_KSPLIT")) L.ksplit = std::min(2, std::max(0, atoi(ks)));
            L.ks_tiles = (Rp % 256 == 0 && maxct >= 8) ? (Rp / 256) * ((maxct + 7) / 8) : 0;
            if ((rc = ksws_ensure(h))) {
                tdmpc2_plan_destroy(h);
                return rc;
            }
            // the few-row path (layered_mid.cuh): partial sums of a launch's two problems -- parts x tiles <= #CUs tiles of 64 x 256
            // (128 x 256) floats each.  Only where a single plan fits the chip in one round of such tiles, and the second buffer set exists.
            {
                const size_t cus = (size_t)(h->num_cus > 0 ? h->num_cus : 256);
                if (L.HA2 && ((size_t)N + 127) / 128 * ((maxct + 7) / 8) <= cus) {
                    L.mws_cap = cus * 128 * 256;
                    if ((rc = dev_alloc(h, (void **)&L.mws[0], L.mws_cap * 4)) || (rc = dev_alloc(h, (void **)&L.mws[1], L.mws_cap * 4))) {
                        tdmpc2_plan_destroy(h);
                        return rc;
                    }
                }
                if (const char *fr = getenv("TDMPC2_FEWROW")) L.mid = atoi(fr) != 0;
            }
            // fp32 pre-activations of the NormedLinear layers whose epilogue is not fused (the fallback after a reported wait,
            // TDMPC2_TUNE_FUSE_LN = 0, tiles the fused path does not take): one buffer per chain
#ifdef GW_TIMING
            if (getenv("TDMPC2_GW_TIMING")) {
                if ((rc = dev_alloc(h, (void **)&L.gw_timing, 32 * 8)) || hipMemset(L.gw_timing, 0, 32 * 8) != hipSuccess) {
                    tdmpc2_plan_destroy(h);
                    return rc ? rc : fail(TDMPC2_ERR_HIP, "hipMemset failed");
                }
            }
#endif
            L.ldpre = std::max(L.Mp, (int)round_up((size_t)c.latent_dim, 32));
            if ((rc = dev_alloc(h, (void **)&L.PRE, Rp * L.ldpre * 4)) || (L.side && (rc = dev_alloc(h, (void **)&L.PRE2, Rp * L.ldpre * 4)))) {
                tdmpc2_plan_destroy(h);
                return rc;
            }
        }
        // stale rows of padded tiles are computed but never read back; start them finite
        if (hipMemset(L.X, 0, Rp * L.Kin * 4) != hipSuccess || hipMemset(L.HA, 0, Rp * L.Mp * 4) != hipSuccess ||
            hipMemset(L.HB, 0, Rp * L.Mp * 4) != hipSuccess || hipMemset(h->beff, 0, (round_up(E, GBM) + 4) * h->nnets * L.Mp * 4) != hipSuccess) {
            tdmpc2_plan_destroy(h);
            return fail(TDMPC2_ERR_HIP, "hipMemset(workspace) failed");
        }
    }
    if (hipMemcpy(h->bins, bins.data(), bins.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        tdmpc2_plan_destroy(h);
        return fail(TDMPC2_ERR_HIP, "hipMemcpy(bins) failed");
    }
    if (hipEventCreateWithFlags(&h->turn_ev, hipEventDisableTiming) != hipSuccess) {  // (StreamTurn)
        h->turn_ev = nullptr;
        tdmpc2_plan_destroy(h);
        return fail(TDMPC2_ERR_HIP, "hipEventCreate failed");
    }
    if (!h->lay.on) {
        const int ar = h->split ? 0 : 1;
        rc = fused_ops(h->Apad).set_lds(ar, c.episodic, h->lds_bytes);
        if (rc) {
            tdmpc2_plan_destroy(h);
            return TDMPC2_ERR_HIP;
        }
    }
    // cluster path (single-plan latency): split arithmetic, sized for the calls that fit the chip in one round
    if (!h->lay.on && h->split) {
        const long cus = h->num_cus > 0 ? h->num_cus : 256;
        const long per_env = (long)h->tiles * 2;                       // 32-row tiles = clusters per plan
        long envs = std::min<long>((long)c.max_envs, cus / (per_env * CL));
        const size_t tail = ((size_t)2 * c.horizon * c.action_dim * 4 + 15) / 16 * 16;
        h->cl_lds = (size_t)32 * h->row_bytes + 8192 + tail + (size_t)2 * 8 * 4 * 64 * 16 + 64;
        if (envs >= 1 && h->cl_lds <= 160 * 1024) {
            const size_t ncl = (size_t)(envs * per_env);
            if ((rc = dev_alloc(h, (void **)&h->cl_xbuf, ncl * CL_SLOTS * CL_TILE * 4)) ||
                (rc = dev_alloc(h, (void **)&h->cl_zs, ncl * CL * 32 * WIDTH * 4)) ||
                (rc = dev_alloc(h, (void **)&h->cl_flags, ncl * CL_FLAG_STRIDE * 4))) {
                tdmpc2_plan_destroy(h);
                return rc;
            }
            if (hipMemset(h->cl_flags, 0, ncl * CL_FLAG_STRIDE * 4) != hipSuccess ||
                hipHostMalloc((void **)&h->cl_err_host, 64, hipHostMallocMapped) != hipSuccess) {
                tdmpc2_plan_destroy(h);
                return fail(TDMPC2_ERR_HIP, "allocating the cluster path's arrival / error words failed");
            }
            memset(h->cl_err_host, 0, 64);  // hipHostMalloc does not zero: word 0 (verdict) AND word 8 (sticky, fault_poll)
            if (hipHostGetDevicePointer((void **)&h->cl_err_dev, h->cl_err_host, 0) != hipSuccess) {
                tdmpc2_plan_destroy(h);
                return fail(TDMPC2_ERR_HIP, "hipHostGetDevicePointer failed");
            }
            int rcl = 0;
            rcl = cluster_ops(h->Apad).set_lds(c.episodic, h->cl_lds);
            if (rcl) {
                tdmpc2_plan_destroy(h);
                return TDMPC2_ERR_HIP;
            }
            h->cl_max_clusters = (int)ncl;
            // two clusters per tile for ONE plan (ks_rollout_cl2): needs every CU of the chip
            if (!c.episodic && 2 * per_env * CL <= cus && c.horizon <= MAXH) {
                const size_t n2 = (size_t)(2 * per_env);
                if ((rc = dev_alloc(h, (void **)&h->cl2_xbuf, n2 * CL2_SLOTS_HOST * CL_TILE * 4)) ||
                    (rc = dev_alloc(h, (void **)&h->cl2_zs, n2 * CL * 32 * WIDTH * 4)) ||
                    (rc = dev_alloc(h, (void **)&h->cl2_flags, n2 * CL_FLAG_STRIDE * 4)) ||
                    (rc = dev_alloc(h, (void **)&h->cl2_mail, (size_t)per_env * 32 * 2 * 4))) {
                    tdmpc2_plan_destroy(h);
                    return rc;
                }
                if (hipMemset(h->cl2_flags, 0, n2 * CL_FLAG_STRIDE * 4) != hipSuccess) {
                    tdmpc2_plan_destroy(h);
                    return fail(TDMPC2_ERR_HIP, "hipMemset(cl2 flags) failed");
                }
            }
        }
    }
    if (const char *cm = getenv("TDMPC2_CLUSTER")) h->cluster_mode = atoi(cm);
    h->user_cluster_mode = h->cluster_mode;  // what was asked for (environment or default); set_tuning / fault recovery go through apply_modes
    h->user_fuse_ln = h->lay.fuse_ln;
#ifdef TDMPC2_TEST_HOOKS  // libtdmpc2_plan_hooks.so (built beside the product library, loaded by the GPU tests of the fault paths only)
    if (const char *cf = getenv("TDMPC2_CLUSTER_FAULT")) h->cl_fault = atoi(cf);
#endif
    if (h->cl_fault) h->lay.mid = false;  // the hook mutes a workgroup of the WAITING paths: the handle runs them (the few-row path has no waits)
#ifdef SPLIT_TIMING  // in-kernel phase timers exist in -DSPLIT_TIMING builds only (tools/ablate.sh)
    if (getenv("TDMPC2_TIMING")) {
        if (dev_alloc(h, (void **)&h->timing, 16 * 8) == 0) (void)hipMemset(h->timing, 0, 16 * 8);
    }
#endif
    *out = h;
    return TDMPC2_OK;
}

void tdmpc2_plan_destroy(tdmpc2_plan_t *h) {
    if (!h) return;
    DevGuard dev_(h->cfg.device);
    if (h->timing) {  // in-kernel phase timers of a -DSPLIT_TIMING build (tools/ablate.sh)
        unsigned long long t[16];
        if (hipMemcpy(t, h->timing, sizeof t, hipMemcpyDeviceToHost) == hipSuccess && t[15] > 0) {
            static const char *names[16] = {"kloop", "epi_post", "head", "actions", "park/unpark", "tile_from_global", "epi_stats", "epi_sync",
                                            "epi_bias", "epi_combine", "epi_math_store", "cluster_wait", "head_kloop", "head_stage", "total", "workgroups"};
            fprintf(stderr, "[tdmpc2_plan timing max_envs=%d] mean cycles per workgroup (one wave, SPLIT_TIMING_WAVE):", h->cfg.max_envs);
            for (int i = 0; i < 15; ++i)
                if (names[i][0]) fprintf(stderr, " %s=%.0f", names[i], (double)t[i] / (double)t[15]);
            fprintf(stderr, "\n");
        }
    }
    if (h->lay.gw_timing) {  // phase clocks of g_gemm_w (-DGW_TIMING build + TDMPC2_GW_TIMING=1)
        unsigned long long t[32];
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(t, h->lay.gw_timing, sizeof t, hipMemcpyDeviceToHost) == hipSuccess) {
            static const char *cls[4] = {"mish K<1024", "mish K>=1024", "simnorm K<1024", "simnorm K>=1024"};
            for (int c = 0; c < 4; ++c) {
                const unsigned long long *q = t + 8 * c;
                if (!q[6]) continue;
                const double n = (double)q[6];
                fprintf(stderr, "[g_gemm_w timing max_envs=%d, %s] workgroups=%llu mean cycles of wave 0: loop_end=%.0f stats_stored=%.0f "
                        "peers_arrived=%.0f row_stats=%.0f end=%.0f\n", h->cfg.max_envs, cls[c], q[6], q[1] / n, q[2] / n, q[3] / n, q[4] / n, q[5] / n);
            }
        }
    }
    if (h->lay.side) {  // back to the pool, never destroyed (see SidePool)
        SideRes sr;
        sr.stream = h->lay.side; sr.ev[0] = h->lay.ev_fork; sr.ev[1] = h->lay.ev_side; sr.ev[2] = h->lay.ev_xread;
        side_release(h->cfg.device, sr);
    }
    for (void *p : h->allocs) (void)hipFree(p);
    if (h->cl_err_host) (void)hipHostFree(h->cl_err_host);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    if (h->turn_ev) (void)hipEventDestroy(h->turn_ev);
    delete h;
}

uint64_t tdmpc2_plan_device_bytes(const tdmpc2_plan_t *h) { return h ? h->bytes : 0; }
int tdmpc2_plan_path(const tdmpc2_plan_t *h) { return h ? h->cfg.path : -1; }
int tdmpc2_plan_precision(const tdmpc2_plan_t *h) { return h ? h->cfg.precision : -1; }

namespace {
// Shape of one nn.Linear of the world model as the planner stores it (tdmpc2/common/world_model.py:26-30).
struct LayerShape {
    int in, out;      // nn.Linear in_features / out_features
    int nz, nt, na;   // source columns: latent or hidden | task embedding | action
    int KB, CT;
    bool has_ln, mish;  // LayerNorm after the Linear; Mish (hidden layers) as opposed to SimNorm / none
    int heads;
    size_t wsz /* floats' worth of packed weights per head */, bsz, gsz, esz;
};
LayerShape layer_shape(const tdmpc2_plan *h, int net, int layer) {
    const tdmpc2_plan_cfg &c = h->cfg;
    LayerShape s{};
    const bool is_q = net == TDMPC2_NET_Q || net == TDMPC2_NET_TARGET_Q;
    const bool takes_action = (net == TDMPC2_NET_DYNAMICS || net == TDMPC2_NET_REWARD || is_q);
    s.heads = is_q ? c.num_q : 1;
    if (layer == 0) {
        s.in = c.latent_dim + c.task_dim + (takes_action ? c.action_dim : 0);
        s.out = c.mlp_dim;
    } else if (layer == 1) {
        s.in = c.mlp_dim;
        s.out = c.mlp_dim;
    } else {
        s.in = c.mlp_dim;
        s.out = net == TDMPC2_NET_DYNAMICS ? c.latent_dim : net == TDMPC2_NET_PI ? 2 * c.action_dim
                : net == TDMPC2_NET_TERMINATION ? 1 : (c.num_bins > 1 ? c.num_bins : 1);
    }
    s.has_ln = (layer < 2) || net == TDMPC2_NET_DYNAMICS;
    s.mish = layer < 2;
    s.nz = (layer == 0) ? c.latent_dim : c.mlp_dim;
    s.nt = (layer == 0) ? c.task_dim : 0;
    s.na = (layer == 0 && takes_action) ? c.action_dim : 0;
    // packed contraction length: the fused kernels pad the action columns to 16, the layered GEMMs the whole row to GBK
    const int Kp = h->lay.on ? (int)round_up((size_t)s.nz + s.na, GBK) : s.nz + (s.na + 15) / 16 * 16;
    s.KB = h->split ? Kp / 16 : Kp / 8;
    s.CT = (s.out + 31) / 32;
    s.wsz = h->split ? (size_t)s.CT * s.KB * 512 /* floats' worth of 1024 halfs */ : (size_t)s.CT * s.KB * 256;
    s.bsz = (size_t)s.CT * 32;
    s.gsz = round_up((size_t)s.out, 4);
    s.esz = (size_t)s.out * (s.nt > 0 ? s.nt : 0);
    return s;
}

// One slab per tensor kind holding all ensemble members at a constant stride (the layered GEMM selects a member per plan
// by stride); allocated on the first bind (or import) of a (net, layer).
int ensure_layer_alloc(tdmpc2_plan *h, int net, int layer, const LayerShape &sh) {
    HostLayer &L0 = net_of(h, net, 0)->l[layer];
    if (L0.alloc) return 0;
    int rc;
    if (h->split && !net_of(h, net, 0)->scal) {  // the net's [heads][3] scalar table, defaults: scales 2^5, kw 0
        LayerScal *tab = nullptr;
        if ((rc = dev_alloc(h, (void **)&tab, (size_t)sh.heads * 3 * sizeof(LayerScal)))) return rc;
        std::vector<LayerScal> init((size_t)sh.heads * 3, LayerScal{1.f, 1.f / ACT_SCALE, 0u, 0, ACT_SCALE, ACT_SCALE_LOG2, 0u, 0u});
        HIP_TRY(hipMemcpy(tab, init.data(), init.size() * sizeof(LayerScal), hipMemcpyHostToDevice));
        for (int hd = 0; hd < sh.heads; ++hd) net_of(h, net, hd)->scal = tab + (size_t)hd * 3;
    }
    float *wslab = nullptr, *bslab = nullptr, *gslab = nullptr, *betaslab = nullptr, *eslab = nullptr;
    if ((rc = dev_alloc(h, (void **)&wslab, sh.heads * sh.wsz * 4))) return rc;
    if ((rc = dev_alloc(h, (void **)&bslab, sh.heads * sh.bsz * 4))) return rc;
    if (sh.has_ln) {
        if ((rc = dev_alloc(h, (void **)&gslab, sh.heads * sh.gsz * 4))) return rc;
        if ((rc = dev_alloc(h, (void **)&betaslab, sh.heads * sh.gsz * 4))) return rc;
    }
    if (sh.nt > 0 && (rc = dev_alloc(h, (void **)&eslab, sh.heads * sh.esz * 4))) return rc;
    for (int hd = 0; hd < sh.heads; ++hd) {
        HostNet *N = net_of(h, net, hd);
        HostLayer &L = N->l[layer];
        L.alloc = true;
        L.KB = sh.KB; L.CT = sh.CT; L.out = sh.out;
        L.wbytes = sh.wsz * 4;
        if (h->split) {
            L.wps = reinterpret_cast<_Float16 *>(wslab + hd * sh.wsz);
            L.scal = N->scal + layer;
            L.oscale = &L.scal->oscale;
            L.ascale = &L.scal->ascale;
        } else {
            L.wp = wslab + hd * sh.wsz;
            L.oscale = h->one;  // the exact arithmetic has no scales
            L.ascale = h->one;
        }
        L.bias = bslab + hd * sh.bsz;
        L.g = sh.has_ln ? gslab + hd * sh.gsz : nullptr;
        L.b = sh.has_ln ? betaslab + hd * sh.gsz : nullptr;
        L.wemb = sh.nt > 0 ? eslab + hd * sh.esz : nullptr;
    }
    return 0;
}
}  // namespace

int tdmpc2_plan_bind_weights(tdmpc2_plan_t *h, int net, int layer, const float *W, const float *b, const float *ln_g,
                             const float *ln_b, int out_features, int in_features, void *stream) {
    if (!h || !W || !b) return fail(TDMPC2_ERR_INVALID, "null argument");
    if (layer < 0 || layer > 2) return fail(TDMPC2_ERR_INVALID, "layer %d outside [0, 2]", layer);
    if (net < TDMPC2_NET_DYNAMICS || net > TDMPC2_NET_TARGET_Q) return fail(TDMPC2_ERR_INVALID, "unknown net %d", net);
    const tdmpc2_plan_cfg &c = h->cfg;
    if (net == TDMPC2_NET_TERMINATION && !c.episodic)
        return fail(TDMPC2_ERR_INVALID, "termination head bound on a non-episodic planner");
    ENTER_ON(h, stream);
    hipStream_t st = (hipStream_t)stream;
    const LayerShape sh = layer_shape(h, net, layer);
    if (in_features != sh.in || out_features != sh.out)
        return fail(TDMPC2_ERR_INVALID, "net %d layer %d: got [%d, %d], expected [%d, %d]", net, layer, out_features,
                    in_features, sh.out, sh.in);
    if (sh.has_ln && (!ln_g || !ln_b)) return fail(TDMPC2_ERR_INVALID, "net %d layer %d needs LayerNorm parameters", net, layer);
    int rc = ensure_layer_alloc(h, net, layer, sh);
    if (rc) return rc;
    for (int hd = 0; hd < sh.heads; ++hd) {
        HostNet *N = net_of(h, net, hd);
        HostLayer &L = N->l[layer];
        const float *Wh = W + (size_t)hd * out_features * in_features;
        if (h->split) {
            HIP_TRY(hipMemsetAsync(&L.scal->maxbits, 0, 4, st));
            hipLaunchKernelGGL(k_absmax, dim3(256), dim3(256), 0, st, Wh, (size_t)out_features * in_features, &L.scal->maxbits);
            hipLaunchKernelGGL(k_wscale, dim3(1), dim3(1), 0, st, L.scal);
            hipLaunchKernelGGL(k_pack_split, dim3(512), dim3(256), 0, st, Wh, out_features, in_features, sh.nz, sh.nt, sh.na, sh.CT,
                               sh.KB, &L.scal->wscale, L.wps);
            if (sh.has_ln) {
                HIP_TRY(hipMemsetAsync(&L.scal->gmax, 0, 8, st));
                hipLaunchKernelGGL(k_absmax, dim3(4), dim3(256), 0, st, ln_g + (size_t)hd * out_features, (size_t)out_features, &L.scal->gmax);
                hipLaunchKernelGGL(k_absmax, dim3(4), dim3(256), 0, st, ln_b + (size_t)hd * out_features, (size_t)out_features, &L.scal->bmax);
            }
            hipLaunchKernelGGL(k_ascale, dim3(1), dim3(1), 0, st, L.scal, out_features, sh.has_ln && sh.mish ? 1 : 0);
            hipLaunchKernelGGL(k_net_scales, dim3(1), dim3(1), 0, st, N->scal);
        } else {
            hipLaunchKernelGGL(k_pack_weight, dim3(512), dim3(256), 0, st, Wh, out_features, in_features, sh.nz, sh.nt, sh.na, sh.CT,
                               sh.KB, L.wp);
        }
        hipLaunchKernelGGL(k_copy_pad, dim3(1), dim3(256), 0, st, b + (size_t)hd * out_features, out_features, sh.CT * 32, L.bias);
        if (sh.has_ln) {
            const int gb = (out_features + 255) / 256;
            hipLaunchKernelGGL(k_copy_pad, dim3(gb), dim3(256), 0, st, ln_g + (size_t)hd * out_features, out_features, out_features, L.g);
            hipLaunchKernelGGL(k_copy_pad, dim3(gb), dim3(256), 0, st, ln_b + (size_t)hd * out_features, out_features, out_features, L.b);
        }
        if (sh.nt > 0)
            hipLaunchKernelGGL(k_copy_cols, dim3(64), dim3(256), 0, st, Wh, out_features, in_features, sh.nz, sh.nt, L.wemb);
        HIP_TRY(hipGetLastError());
        L.bound = true;
    }
    return TDMPC2_OK;
}

namespace {
int ensure_enc_alloc(tdmpc2_plan *h, int layer, int in_features, int out_features) {
    const tdmpc2_plan_cfg &c = h->cfg;
    tdmpc2_plan::Enc &L = h->enc[layer];
    if (L.wt && (L.in != in_features || L.out != out_features))
        return fail(TDMPC2_ERR_STATE, "encoder layer %d re-bound with a different shape", layer);
    int rc;
    if (!L.wt) {
        if ((rc = dev_alloc(h, (void **)&L.wt, (size_t)in_features * out_features * 4))) return rc;
        if ((rc = dev_alloc(h, (void **)&L.bias, (size_t)out_features * 4))) return rc;
        if ((rc = dev_alloc(h, (void **)&L.g, (size_t)out_features * 4))) return rc;
        if ((rc = dev_alloc(h, (void **)&L.b, (size_t)out_features * 4))) return rc;
        L.in = in_features;
        L.out = out_features;
    }
    if (!h->zenc && (rc = dev_alloc(h, (void **)&h->zenc, (size_t)c.max_envs * c.latent_dim * 4))) return rc;
    const int w = std::max(in_features, out_features);
    if (w > ENC_WIDE && w > h->enc_ws_width) {  // workspace of the layer-at-a-time encoder path (grows with the widest layer)
        if ((rc = dev_alloc(h, (void **)&h->enc_y, (size_t)c.max_envs * w * 4))) return rc;
        if ((rc = dev_alloc(h, (void **)&h->enc_x, (size_t)c.max_envs * w * 4))) return rc;
        h->enc_ws_width = w;
    }
    return 0;
}
}  // namespace

int tdmpc2_plan_bind_encoder(tdmpc2_plan_t *h, int layer, int n_layers, const float *W, const float *b, const float *ln_g,
                             const float *ln_b, int out_features, int in_features, void *stream) {
    if (!h || !W || !b || !ln_g || !ln_b) return fail(TDMPC2_ERR_INVALID, "null argument");
    if (n_layers < 1 || n_layers > ENC_MAX_LAYERS) return fail(TDMPC2_ERR_INVALID, "encoder depth %d outside [1, %d]", n_layers, ENC_MAX_LAYERS);
    if (layer < 0 || layer >= n_layers) return fail(TDMPC2_ERR_INVALID, "encoder layer %d outside [0, %d)", layer, n_layers);
    if (out_features < 1 || out_features > ENC_THREADS * ENC_MAX_PER_THREAD || in_features < 1)
        return fail(TDMPC2_ERR_UNSUPPORTED, "encoder layer %d: width %d outside [1, %d]", layer, out_features, ENC_THREADS * ENC_MAX_PER_THREAD);
    const tdmpc2_plan_cfg &c = h->cfg;
    if (layer == n_layers - 1 && out_features != c.latent_dim)
        return fail(TDMPC2_ERR_INVALID, "the last encoder layer has %d outputs, latent_dim is %d", out_features, c.latent_dim);
    if (layer == n_layers - 1 && (c.latent_dim % c.simnorm_dim || (c.simnorm_dim & (c.simnorm_dim - 1)) || c.simnorm_dim > 64))
        return fail(TDMPC2_ERR_UNSUPPORTED, "SimNorm groups of %d over %d latents", c.simnorm_dim, c.latent_dim);
    if (h->enc_layers && h->enc_layers != n_layers) return fail(TDMPC2_ERR_STATE, "encoder depth changed from %d to %d", h->enc_layers, n_layers);
    ENTER_ON(h, stream);
    int rc = ensure_enc_alloc(h, layer, in_features, out_features);
    if (rc) return rc;
    tdmpc2_plan::Enc &L = h->enc[layer];
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)in_features * out_features;
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, st, W, L.wt, out_features, in_features);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(L.bias, b, (size_t)out_features * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(L.g, ln_g, (size_t)out_features * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(L.b, ln_b, (size_t)out_features * 4, hipMemcpyDeviceToDevice, st));
    L.bound = true;
    h->enc_layers = n_layers;
    return TDMPC2_OK;
}

namespace {
int launch_encode(tdmpc2_plan *h, int E, const float *obs, int obs_dim, const float *task_emb, float *z, hipStream_t st) {
    const tdmpc2_plan_cfg &c = h->cfg;
    if (!h->enc_layers) return fail(TDMPC2_ERR_STATE, "no encoder bound (tdmpc2_plan_bind_encoder)");
    EncodeParams p{};
    int maxw = 0;
    for (int l = 0; l < h->enc_layers; ++l) {
        const tdmpc2_plan::Enc &L = h->enc[l];
        if (!L.bound) return fail(TDMPC2_ERR_STATE, "encoder layer %d of %d is not bound", l, h->enc_layers);
        const int exp_in = l == 0 ? obs_dim + c.task_dim : h->enc[l - 1].out;
        if (L.in != exp_in) return fail(TDMPC2_ERR_INVALID, "encoder layer %d takes %d inputs, the data brings %d", l, L.in, exp_in);
        p.l[l] = EncLayerDev{L.wt, L.bias, L.g, L.b, L.in, L.out};
        maxw = std::max(maxw, std::max(L.in, L.out));
    }
    if (c.task_dim > 0 && !task_emb) return fail(TDMPC2_ERR_INVALID, "multitask encoder needs task_emb");
    if (maxw > ENC_WIDE) {  // layer-at-a-time across the chip (encoder_kernels.cuh)
        if (E > c.max_envs) return fail(TDMPC2_ERR_INVALID, "encoders wider than %d take at most max_envs = %d rows per call (got %d)", ENC_WIDE, c.max_envs, E);
        if (!h->enc_y || h->enc_ws_width < maxw) return fail(TDMPC2_ERR_STATE, "encoder workspace missing (bind every layer first)");
        for (int l = 0; l < h->enc_layers; ++l) {
            const tdmpc2_plan::Enc &L = h->enc[l];
            EncGemvParams g{};
            g.wt = L.wt; g.bias = L.bias; g.in = L.in; g.out = L.out; g.y = h->enc_y;
            if (l == 0) { g.obs = obs; g.emb = task_emb; g.obs_dim = obs_dim; g.T = c.task_dim; }
            else { g.x = h->enc_x; g.ldx = h->enc[l - 1].out; }
            hipLaunchKernelGGL(k_enc_gemv, dim3((L.out + 63) / 64, E), dim3(256), ((size_t)L.in + 256) * 4, st, g);
            EncNormParams n{};
            const bool last = l == h->enc_layers - 1;
            n.y = h->enc_y; n.g = L.g; n.b = L.b; n.out = last ? z : h->enc_x; n.width = L.out; n.last = last; n.simnorm_dim = c.simnorm_dim;
            hipLaunchKernelGGL(k_enc_norm, dim3(E), dim3(ENC_THREADS), 0, st, n);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    p.nl = h->enc_layers; p.obs_dim = obs_dim; p.T = c.task_dim; p.maxw = maxw; p.simnorm_dim = c.simnorm_dim;
    p.obs = obs; p.task_emb = task_emb; p.z = z;
    const size_t lds = ((size_t)2 * maxw + ENC_THREADS / 64) * 4;
    hipLaunchKernelGGL(k_encode, dim3(E), dim3(ENC_THREADS), lds, st, p);
    HIP_TRY(hipGetLastError());
    return 0;
}
}  // namespace

int tdmpc2_plan_encode(tdmpc2_plan_t *h, int n_envs, const float *obs, int obs_dim, const float *task_emb, float *z_out,
                       void *stream) {
    if (!h || !obs || !z_out) return fail(TDMPC2_ERR_INVALID, "null argument");
    if (n_envs < 1) return fail(TDMPC2_ERR_INVALID, "n_envs %d < 1", n_envs);
    ENTER_ON(h, stream);
    return launch_encode(h, n_envs, obs, obs_dim, task_emb, z_out, (hipStream_t)stream);
}

int tdmpc2_plan_run_obs(tdmpc2_plan_t *h, int n_envs, const float *obs, int obs_dim, const float *task_emb,
                        const float *act_mask, const float *discount_pow, float *prev_mean, const uint8_t *t0, int eval_mode,
                        const tdmpc2_noise *tape, uint64_t seed, float *action, void *stream) {
    if (!h || !obs) return fail(TDMPC2_ERR_INVALID, "null argument");
    if (n_envs < 1 || n_envs > h->cfg.max_envs) return fail(TDMPC2_ERR_INVALID, "n_envs %d outside [1, %d]", n_envs, h->cfg.max_envs);
    ENTER_ON(h, stream);
    int rc = launch_encode(h, n_envs, obs, obs_dim, task_emb, h->zenc, (hipStream_t)stream);
    if (rc) return rc;
    return run_impl(h, n_envs, h->zenc, task_emb, act_mask, discount_pow, prev_mean, t0, eval_mode, tape, seed, action, nullptr,
                    stream);
}

namespace {
// per-task tables of a multitask value call: effective first-layer biases of pi and of the chosen Q ensemble, action
// masks, discounts; (re)built on every call (the weights may have been re-bound), storage grown on demand
int build_task_tables(tdmpc2_plan *h, const tdmpc2_task_tables *tk, bool target, size_t rows_p, hipStream_t st) {
    const tdmpc2_plan_cfg &c = h->cfg;
    const int nt = tk->n_tasks;
    const size_t width = h->lay.on ? (size_t)h->lay.Mp : (size_t)WIDTH;
    int rc;
    if (nt > h->tab_tasks) {  // grow: the old tables are released (stream-ordered: earlier calls on `st` are done with them)
        HIP_TRY(hipStreamSynchronize(st));
        dev_release(h, h->beff_tab); dev_release(h, h->mask_tab); dev_release(h, h->disc_tab);
        h->beff_tab = h->mask_tab = h->disc_tab = nullptr;
        h->tab_tasks = 0;
        if ((rc = dev_alloc(h, (void **)&h->beff_tab, (size_t)nt * h->nnets * width * 4))) return rc;
        if ((rc = dev_alloc(h, (void **)&h->mask_tab, (size_t)nt * c.action_dim * 4))) return rc;
        if ((rc = dev_alloc(h, (void **)&h->disc_tab, (size_t)nt * 4))) return rc;
        h->tab_tasks = nt;
    }
    if (rows_p > h->task_rows_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        dev_release(h, h->task_rows);
        h->task_rows = nullptr;
        h->task_rows_cap = 0;
        if ((rc = dev_alloc(h, (void **)&h->task_rows, rows_p * 4))) return rc;
        h->task_rows_cap = rows_p;
    }
    HIP_TRY(hipMemcpyAsync(h->mask_tab, tk->act_mask, (size_t)nt * c.action_dim * 4, hipMemcpyDeviceToDevice, st));
    if (tk->discount) HIP_TRY(hipMemcpyAsync(h->disc_tab, tk->discount, (size_t)nt * 4, hipMemcpyDeviceToDevice, st));
    const HostNet *qarr = target ? h->tq : h->q;
    if (h->lay.on) return lay_setup(h, st, nt, tk->task_emb, nullptr, nullptr, false, h->beff_tab, qarr);
    TaskBiasParams p{};
    p.T = c.task_dim; p.nq = c.num_q; p.nnets = h->nnets; p.task_emb = tk->task_emb; p.beff_tab = h->beff_tab;
    p.wemb[BE_PI] = h->pi.l[0].wemb; p.bias[BE_PI] = h->pi.l[0].bias;
    for (int i = 0; i < c.num_q; ++i) { p.wemb[BE_Q0 + i] = qarr[i].l[0].wemb; p.bias[BE_Q0 + i] = qarr[i].l[0].bias; }
    hipLaunchKernelGGL(ks_task_bias, dim3(nt), dim3(WIDTH), 0, st, p);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_value(tdmpc2_plan *h, int rows, const float *z, bool target, bool reduce_min, const float *pi_eps, const int32_t *qidx,
                 uint64_t seed, const float *reward, const float *terminated, float discount, const tdmpc2_task_tables *tk,
                 float *action, float *out, hipStream_t st) {
    const tdmpc2_plan_cfg &c = h->cfg;
    if (c.multitask) {
        if (!tk || !tk->task_ids || !tk->task_emb || !tk->act_mask || tk->n_tasks < 1)
            return fail(TDMPC2_ERR_INVALID, "multitask policy_value / td_target need the row -> task map and the per-task tables");
        if (reward && !tk->discount) return fail(TDMPC2_ERR_INVALID, "multitask td_target needs the per-task discounts");
    } else if (tk) {
        return fail(TDMPC2_ERR_INVALID, "task tables given to a single-task handle");
    }
    int rc = check_ready(h);
    if (rc) return rc;
    (void)fault_poll(h);  // (a wait that gave up in an earlier call: this one already runs on the paths without waits)
    // between the calls of a sharded plan word 0 is that plan's verdict (its final pick reads it): keep it
    if (!h->in_shard && (rc = fault_fresh(h, st))) return rc;
    if (target)
        for (int i = 0; i < 3; ++i)
            for (int qh = 0; qh < c.num_q; ++qh)
                if (!h->tq[qh].l[i].bound)
                    return fail(TDMPC2_ERR_STATE, "layer %d of target Q head %d is not bound (net TDMPC2_NET_TARGET_Q)", i, qh);
    const unsigned call = h->call++;
    if (h->lay.on) {
        const size_t rows_p = round_up((size_t)rows, GBM), cap = round_up((size_t)c.max_envs * c.num_samples, GBM);
        if (rows_p > cap)
            return fail(TDMPC2_ERR_INVALID, "the layered workspace holds %zu rows (max_envs x num_samples); got %d", cap, rows);
        if (c.multitask) {
            if ((rc = build_task_tables(h, tk, target, rows_p, st))) return rc;
            HIP_TRY(hipMemsetAsync(h->task_rows, 0, rows_p * 4, st));
            HIP_TRY(hipMemcpyAsync(h->task_rows, tk->task_ids, (size_t)rows * 4, hipMemcpyDeviceToDevice, st));
        }
        // the two heads: given, or randperm(num_q)[:2] once per call (world_model.py:212)
        if ((rc = lay_set_qidx(h, st, 1, qidx, 2L, c.num_q, 0, seed, call, h->lay.qidx))) return rc;
        return lay_value(h, st, rows, z, target, reduce_min, pi_eps, h->lay.qidx, seed, call, reward, terminated, discount,
                         c.multitask ? h->task_rows : nullptr, action, out);
    }
    if (c.multitask && (rc = build_task_tables(h, tk, target, 0, st))) return rc;
    ValueParamsT<NetS> p{};
    p.rows = rows; p.A = c.action_dim; p.Apad = h->Apad; p.nq = c.num_q; p.num_bins = c.num_bins; p.reduce_min = reduce_min ? 1 : 0;
    p.log_std_min = c.log_std_min; p.log_std_dif = c.log_std_dif; p.discount = discount;
    p.pi = to_dev<NetS>(h->pi);
    for (int i = 0; i < c.num_q; ++i) p.q[i] = to_dev<NetS>(target ? h->tq[i] : h->q[i]);
    p.bins = h->bins; p.z = z; p.pi_eps = pi_eps; p.qidx = qidx; p.seed = seed; p.call = call;
    p.reward = reward; p.terminated = terminated; p.action = action; p.out = out;
    p.nnets = h->nnets;
    if (c.multitask) {
        p.task_ids = tk->task_ids; p.beff_tab = h->beff_tab; p.mask_tab = h->mask_tab;
        p.disc_tab = reward ? h->disc_tab : nullptr;
    }
    const int grid = (rows + ROWS - 1) / ROWS;
    const size_t lds = h->lds_bytes;
    const int ar = h->split ? 0 : 1;
    fused_ops(h->Apad).value(ar, p, grid, lds, st);
    HIP_TRY(hipGetLastError());
    return 0;
}
}  // namespace

int tdmpc2_plan_policy_value_mt(tdmpc2_plan_t *h, int n_rows, const float *z, const tdmpc2_task_tables *tasks, int use_target,
                                int reduce_min, const float *pi_eps, const int32_t *qidx, uint64_t seed, float *action, float *q,
                                void *stream) {
    if (!h || !z || !q) return fail(TDMPC2_ERR_INVALID, "null argument");
    if (n_rows < 1) return fail(TDMPC2_ERR_INVALID, "n_rows %d < 1", n_rows);
    ENTER_ON(h, stream);
    return launch_value(h, n_rows, z, use_target != 0, reduce_min != 0, pi_eps, qidx, seed, nullptr, nullptr, 0.f, tasks, action, q,
                        (hipStream_t)stream);
}

int tdmpc2_plan_policy_value(tdmpc2_plan_t *h, int n_rows, const float *z, int use_target, int reduce_min, const float *pi_eps,
                             const int32_t *qidx, uint64_t seed, float *action, float *q, void *stream) {
    return tdmpc2_plan_policy_value_mt(h, n_rows, z, nullptr, use_target, reduce_min, pi_eps, qidx, seed, action, q, stream);
}

int tdmpc2_plan_td_target_mt(tdmpc2_plan_t *h, int n_rows, const float *next_z, const float *reward, const float *terminated,
                             float discount, const tdmpc2_task_tables *tasks, const float *pi_eps, const int32_t *qidx,
                             uint64_t seed, float *td, void *stream) {
    if (!h || !next_z || !reward || !terminated || !td) return fail(TDMPC2_ERR_INVALID, "null argument");
    if (n_rows < 1) return fail(TDMPC2_ERR_INVALID, "n_rows %d < 1", n_rows);
    ENTER_ON(h, stream);
    return launch_value(h, n_rows, next_z, true, true, pi_eps, qidx, seed, reward, terminated, discount, tasks, nullptr, td,
                        (hipStream_t)stream);
}

int tdmpc2_plan_td_target(tdmpc2_plan_t *h, int n_rows, const float *next_z, const float *reward, const float *terminated,
                          float discount, const float *pi_eps, const int32_t *qidx, uint64_t seed, float *td, void *stream) {
    return tdmpc2_plan_td_target_mt(h, n_rows, next_z, reward, terminated, discount, nullptr, pi_eps, qidx, seed, td, stream);
}

// ---------------------------------------------------------------- one plan sharded over several GPUs (last row)
// The sample rows of every plan are split over the ranks; everything else is replicated: every rank holds the weights, runs
// the identical set-up, draws the identical actions (same tape / same Philox seed) and performs the identical elite
// selection + refit on the all-gathered values.  Per CEM iteration a rank calls shard_values for ITS row range, the host
// all-gathers value[E, N / G] over RCCL (4 KB per plan), and every rank calls shard_refit.  tdmpc2_amd/dist.py drives it.
namespace {
void fill_refit(tdmpc2_plan *h, RefitParams &fp, int E, int it, int eval_mode, float *value, const float *act_mask,
                const tdmpc2_noise *tape, uint64_t seed, unsigned call, float *prev_mean, float *action, const tdmpc2_debug *dbg,
                int stage) {
    const tdmpc2_plan_cfg &c = h->cfg;
    const int H = c.horizon, N = c.num_samples, A = c.action_dim, K = c.num_elites, I = c.iterations;
    fp = RefitParams{};
    fp.Nvalid = c.num_valid_samples; fp.E = E; fp.N = N; fp.H = H; fp.A = A; fp.K = K; fp.iter = it; fp.last = (it == I - 1); fp.eval_mode = eval_mode; fp.stage = stage;
    fp.temperature = c.temperature; fp.min_std = c.min_std; fp.max_std = c.max_std;
    fp.value = value; fp.actions = h->actions; fp.act_mask = act_mask; fp.mean = h->mean; fp.std = h->std;
    fp.gumbel_exp = tape ? tape->gumbel_exp : nullptr; fp.final_eps = tape ? tape->final_eps : nullptr;
    fp.seed = seed; fp.call = call; fp.prev_mean = prev_mean; fp.action = action;
    // a bounded inter-workgroup wait (fused NormedLinear epilogue) that gave up in ANY iteration of this sharded plan: NaN
    // action, prev_mean kept -- word 0 is cleared by shard_begin only (in stream order), not by the calls in between
    fp.err = h->cl_err_dev; fp.err2 = nullptr;
    if (dbg) {
        if (dbg->value) { fp.dbg_value = dbg->value + (size_t)it * N; fp.dbg_value_es = (long)I * N; }
        if (dbg->elite_idx) { fp.dbg_idx = dbg->elite_idx + (size_t)it * K; fp.dbg_idx_es = (long)I * K; }
        if (dbg->score) { fp.dbg_score = dbg->score + (size_t)it * K; fp.dbg_score_es = (long)I * K; }
        if (dbg->mean) { fp.dbg_mean = dbg->mean + (size_t)it * H * A; fp.dbg_mean_es = (long)I * H * A; }
        if (dbg->std) { fp.dbg_std = dbg->std + (size_t)it * H * A; fp.dbg_std_es = (long)I * H * A; }
    }
}
}  // namespace

int tdmpc2_plan_shard_begin(tdmpc2_plan_t *h, int n_envs, const float *z0, const float *task_emb, const float *act_mask,
                            const float *prev_mean, const uint8_t *t0, const tdmpc2_noise *tape, uint64_t seed, void *stream) {
    if (!h) return fail(TDMPC2_ERR_INVALID, "null handle");
    ENTER_ON(h, stream);
    int rc = validate_envs(h, n_envs);
    if (rc) return rc;
    if (!z0 || !prev_mean || !t0) return fail(TDMPC2_ERR_INVALID, "null argument");
    const tdmpc2_plan_cfg &c = h->cfg;
    if (c.multitask && (!task_emb || !act_mask)) return fail(TDMPC2_ERR_INVALID, "multitask plan needs task_emb and act_mask");
    if (!c.multitask) { task_emb = nullptr; act_mask = nullptr; }
    if (tape && c.num_pi_trajs > 0 && !tape->pi_traj_eps) return fail(TDMPC2_ERR_INVALID, "noise tape has null fields");
    hipStream_t st = (hipStream_t)stream;
    const unsigned call = h->call++;
    h->shard_call = call;
    // in_shard: no re-arm between the calls of this plan (fault_clean), word 0 of the error line kept until the final pick.  Set
    // only when the whole prologue has been enqueued: a shard_begin that fails leaves the handle as an ordinary call would.
    struct ShardGuard {
        tdmpc2_plan *h; bool ok;
        ~ShardGuard() { h->in_shard = ok; }
    } guard{h, false};
    const int E = n_envs, P = c.num_pi_trajs;
    if (h->lay.on) {
        if ((rc = fault_fresh(h, st))) return rc;
        if ((rc = lay_setup(h, st, E, task_emb, prev_mean, t0, true))) return rc;
        if ((rc = lay_cvec(h, st, E, z0))) return rc;
        if (P > 0 && (rc = lay_pitraj(h, st, E, z0, act_mask, tape ? tape->pi_traj_eps : nullptr, seed, call))) return rc;
        guard.ok = true;
        return TDMPC2_OK;
    }
    if ((rc = launch_setup<NetS>(h, E, z0, task_emb, prev_mean, t0, st))) return rc;
    if (P > 0) {
        PiTrajParamsT<NetS> p{};
        p.E = E; p.N = c.num_samples; p.H = c.horizon; p.A = c.action_dim; p.Apad = h->Apad; p.P = P; p.stride = h->stride;
        p.multitask = c.multitask; p.nnets = h->nnets; p.log_std_min = c.log_std_min; p.log_std_dif = c.log_std_dif;
        p.dyn = to_dev<NetS>(h->dyn); p.pi = to_dev<NetS>(h->pi);
        p.z0 = z0; p.beff = h->beff; p.act_mask = act_mask; p.pi_traj_eps = tape ? tape->pi_traj_eps : nullptr;
        p.seed = seed; p.call = call; p.actions = h->actions; p.zscratch = h->zscratch;
        p.zscratch_estride = (long)h->tiles * ROWS * WIDTH;
        Kern<NetS>::pitraj(h, p, E, st);
        HIP_TRY(hipGetLastError());
    }
    guard.ok = true;
    return TDMPC2_OK;
}

int tdmpc2_plan_shard_values(tdmpc2_plan_t *h, int n_envs, int iter, int row_begin, int row_end, const float *z0,
                             const float *act_mask, const float *disc_pow, const tdmpc2_noise *tape, uint64_t seed,
                             float *value, void *stream) {
    if (!h) return fail(TDMPC2_ERR_INVALID, "null handle");
    ENTER_ON(h, stream);
    int rc = validate_envs(h, n_envs);
    if (rc) return rc;
    if (!z0 || !disc_pow || !value) return fail(TDMPC2_ERR_INVALID, "null argument");
    const tdmpc2_plan_cfg &c = h->cfg;
    const int E = n_envs, N = c.num_samples, A = c.action_dim, I = c.iterations;
    if (iter < 0 || iter >= I) return fail(TDMPC2_ERR_INVALID, "iteration %d outside [0, %d)", iter, I);
    const int gran = h->lay.on ? GBM : ROWS;
    if (row_begin < 0 || row_end > N || row_begin >= row_end || row_begin % gran || row_end % gran)
        return fail(TDMPC2_ERR_INVALID, "row range [%d, %d) must be non-empty, inside [0, %d) and aligned to %d rows", row_begin, row_end, N, gran);
    if (c.multitask && !act_mask) return fail(TDMPC2_ERR_INVALID, "multitask plan needs act_mask");
    if (!c.multitask) act_mask = nullptr;
    if (tape && (!tape->sample_eps || !tape->pi_eps || !tape->qidx)) return fail(TDMPC2_ERR_INVALID, "noise tape has null fields");
    hipStream_t st = (hipStream_t)stream;
    const unsigned call = h->shard_call;
    // (1) the actions of ALL rows of this iteration (replicated: the refit needs every elite's actions)
    int *qbuf = h->lay.on ? h->lay.qidx : h->qidx_buf;
    if ((rc = lay_sample_iteration(h, st, E, iter, act_mask, tape, seed, call, qbuf))) return rc;
    const float *pi_eps = tape ? tape->pi_eps + (size_t)iter * N * A : nullptr;
    // (2) this rank's rows
    if (h->lay.on)
        return lay_estimate_value(h, st, E, z0, act_mask, disc_pow, h->actions, pi_eps, (long)I * N * A, qbuf, seed, call, iter, value,
                                  nullptr, row_begin, row_end - row_begin);
    RolloutParamsT<NetS> rp{};
    fill_rollout<NetS>(h, rp, E);
    rp.z0 = z0; rp.act_mask = act_mask; rp.disc_pow = disc_pow; rp.seed = seed; rp.call = call; rp.given_actions = 1; rp.iter = iter;
    rp.value = value; rp.pi_eps = pi_eps; rp.pi_eps_estride = (long)I * N * A; rp.qidx = qbuf; rp.qidx_estride = 2;
    const int nst = h->force_rows == 32 ? 1 : 2;
    const int trows = 32 * nst;
    rp.tiles = (row_end - row_begin) / trows; rp.tile_off = row_begin / trows;
    Kern<NetS>::rollout(h, rp, E * rp.tiles, st, nst, 8);
    HIP_TRY(hipGetLastError());
    return TDMPC2_OK;
}

int tdmpc2_plan_shard_refit(tdmpc2_plan_t *h, int n_envs, int iter, float *value, const float *act_mask, float *prev_mean,
                            int eval_mode, const tdmpc2_noise *tape, uint64_t seed, float *action, const tdmpc2_debug *dbg,
                            void *stream) {
    if (!h) return fail(TDMPC2_ERR_INVALID, "null handle");
    ENTER_ON(h, stream);
    if (n_envs < 1 || n_envs > h->cfg.max_envs) return fail(TDMPC2_ERR_INVALID, "n_envs=%d outside [1, %d]", n_envs, h->cfg.max_envs);
    if (!value || !prev_mean || !action) return fail(TDMPC2_ERR_INVALID, "null argument");
    const tdmpc2_plan_cfg &c = h->cfg;
    if (iter < 0 || iter >= c.iterations) return fail(TDMPC2_ERR_INVALID, "iteration %d outside [0, %d)", iter, c.iterations);
    if (tape && (!tape->gumbel_exp || (!eval_mode && !tape->final_eps))) return fail(TDMPC2_ERR_INVALID, "noise tape has null fields");
    hipStream_t st = (hipStream_t)stream;
    int stage = 0;
    const size_t lds = refit_lds_bytes(c.num_samples, c.num_elites, c.horizon, c.action_dim, &stage);
    RefitParams fp;
    fill_refit(h, fp, n_envs, iter, eval_mode, value, c.multitask ? act_mask : nullptr, tape, seed, h->shard_call, prev_mean, action, dbg, stage);
    hipLaunchKernelGGL(k_refit, dim3(n_envs), dim3(refit_threads(c.num_samples)), lds, st, fp);
    HIP_TRY(hipGetLastError());
    if (fp.last) {
        h->in_shard = false;
        if (h->safe_once) { h->safe_once = false; apply_modes(h); }
    }
    if (dbg && dbg->actions)
        HIP_TRY(hipMemcpy2DAsync(dbg->actions + (size_t)iter * c.horizon * c.num_samples * c.action_dim,
                                 (size_t)c.iterations * c.horizon * c.num_samples * c.action_dim * 4, h->actions,
                                 (size_t)c.horizon * c.num_samples * c.action_dim * 4, (size_t)c.horizon * c.num_samples * c.action_dim * 4,
                                 n_envs, hipMemcpyDeviceToDevice, st));
    return TDMPC2_OK;
}

// ---------------------------------------------------------------- packed weight file
// Everything a bind produces -- fragment-ordered (and, for the split arithmetic, hi/lo-split and scaled) weights, padded
// biases, LayerNorm parameters, task-embedding columns, the per-layer scale records, the transposed encoder -- as one
// host blob: [PackHdr][segment sizes][segment data].  Importing it is a sequence of plain host-to-device copies: no
// packing kernels, no fp32 checkpoint on the device (reference load path: tdmpc2/tdmpc2.py:81-95).
namespace {
struct PackHdr {
    char magic[8];
    uint32_t version, abi;
    tdmpc2_plan_cfg cfg;  // as resolved by create (path and precision are concrete); max_envs / device are not compared
    uint32_t has_target, enc_layers;
    int32_t enc_in[ENC_MAX_LAYERS], enc_out[ENC_MAX_LAYERS];
    uint64_t nseg, data_bytes;
};
struct Seg {
    void *p;
    size_t bytes;
};
const int kPackNets[] = {TDMPC2_NET_DYNAMICS, TDMPC2_NET_REWARD, TDMPC2_NET_PI, TDMPC2_NET_TERMINATION, TDMPC2_NET_Q, TDMPC2_NET_TARGET_Q};

bool net_in_pack(const tdmpc2_plan *h, int net, bool has_target) {
    if (net == TDMPC2_NET_TERMINATION) return h->cfg.episodic != 0;
    if (net == TDMPC2_NET_TARGET_Q) return has_target;
    return true;
}
// canonical walk; every visited layer must be allocated
void collect_segments(tdmpc2_plan *h, bool has_target, int enc_layers, std::vector<Seg> &out) {
    for (int net : kPackNets) {
        if (!net_in_pack(h, net, has_target)) continue;
        const int heads = layer_shape(h, net, 0).heads;
        for (int hd = 0; hd < heads; ++hd) {
            HostNet *N = net_of(h, net, hd);
            for (int l = 0; l < 3; ++l) {
                const LayerShape sh = layer_shape(h, net, l);
                HostLayer &L = N->l[l];
                out.push_back({h->split ? (void *)L.wps : (void *)L.wp, sh.wsz * 4});
                out.push_back({L.bias, sh.bsz * 4});
                if (sh.has_ln) {
                    out.push_back({L.g, sh.gsz * 4});
                    out.push_back({L.b, sh.gsz * 4});
                }
                if (sh.nt > 0) out.push_back({L.wemb, sh.esz * 4});
            }
            if (h->split) out.push_back({N->scal, 3 * sizeof(LayerScal)});
        }
    }
    for (int l = 0; l < enc_layers; ++l) {
        tdmpc2_plan::Enc &E = h->enc[l];
        out.push_back({E.wt, (size_t)E.in * E.out * 4});
        out.push_back({E.bias, (size_t)E.out * 4});
        out.push_back({E.g, (size_t)E.out * 4});
        out.push_back({E.b, (size_t)E.out * 4});
    }
}
bool target_bound(const tdmpc2_plan *h) {
    for (int i = 0; i < 3; ++i)
        for (int q = 0; q < h->cfg.num_q; ++q)
            if (!h->tq[q].l[i].bound) return false;
    return true;
}
bool same_model(const tdmpc2_plan_cfg &a, const tdmpc2_plan_cfg &b) {
    return a.horizon == b.horizon && a.num_samples == b.num_samples && a.action_dim == b.action_dim && a.latent_dim == b.latent_dim &&
           a.mlp_dim == b.mlp_dim && a.task_dim == b.task_dim && a.num_bins == b.num_bins && a.num_q == b.num_q &&
           a.simnorm_dim == b.simnorm_dim && a.multitask == b.multitask && a.episodic == b.episodic && a.path == b.path &&
           a.precision == b.precision;
}
}  // namespace

int tdmpc2_plan_packed_size(tdmpc2_plan_t *h, uint64_t *bytes) {
    if (!h || !bytes) return fail(TDMPC2_ERR_INVALID, "null argument");
    ENTER(h);
    int rc = check_ready(h);
    if (rc) return rc;
    std::vector<Seg> segs;
    collect_segments(h, target_bound(h), h->enc_layers, segs);
    uint64_t total = sizeof(PackHdr) + segs.size() * sizeof(uint64_t);
    for (const Seg &sg : segs) total += sg.bytes;
    *bytes = total;
    return TDMPC2_OK;
}

int tdmpc2_plan_export_packed(tdmpc2_plan_t *h, void *host_buf, uint64_t bytes, void *stream) {
    if (!h || !host_buf) return fail(TDMPC2_ERR_INVALID, "null argument");
    ENTER_ON(h, stream);
    int rc = check_ready(h);
    if (rc) return rc;
    for (int l = 0; l < h->enc_layers; ++l)
        if (!h->enc[l].bound) return fail(TDMPC2_ERR_STATE, "encoder layer %d of %d is not bound", l, h->enc_layers);
    const bool has_target = target_bound(h);
    std::vector<Seg> segs;
    collect_segments(h, has_target, h->enc_layers, segs);
    PackHdr hdr{};
    memcpy(hdr.magic, "TDMPC2PK", 8);
    hdr.version = 1; hdr.abi = TDMPC2_PLAN_ABI_VERSION; hdr.cfg = h->cfg; hdr.has_target = has_target ? 1 : 0;
    hdr.enc_layers = (uint32_t)h->enc_layers;
    for (int l = 0; l < h->enc_layers; ++l) { hdr.enc_in[l] = h->enc[l].in; hdr.enc_out[l] = h->enc[l].out; }
    hdr.nseg = segs.size();
    for (const Seg &sg : segs) hdr.data_bytes += sg.bytes;
    const uint64_t need = sizeof(PackHdr) + segs.size() * sizeof(uint64_t) + hdr.data_bytes;
    if (bytes < need) return fail(TDMPC2_ERR_INVALID, "buffer of %llu bytes, the packed weights need %llu", (unsigned long long)bytes, (unsigned long long)need);
    char *dst = static_cast<char *>(host_buf);
    memcpy(dst, &hdr, sizeof hdr);
    uint64_t *sizes = reinterpret_cast<uint64_t *>(dst + sizeof hdr);
    char *data = dst + sizeof hdr + segs.size() * sizeof(uint64_t);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));  // binds on this stream have landed
    for (size_t i = 0; i < segs.size(); ++i) {
        sizes[i] = segs[i].bytes;
        HIP_TRY(hipMemcpy(data, segs[i].p, segs[i].bytes, hipMemcpyDeviceToHost));
        data += segs[i].bytes;
    }
    return TDMPC2_OK;
}

int tdmpc2_plan_import_packed(tdmpc2_plan_t *h, const void *host_buf, uint64_t bytes, void *stream) {
    if (!h || !host_buf) return fail(TDMPC2_ERR_INVALID, "null argument");
    ENTER_ON(h, stream);
    if (bytes < sizeof(PackHdr)) return fail(TDMPC2_ERR_INVALID, "packed weights: truncated header");
    PackHdr hdr;
    memcpy(&hdr, host_buf, sizeof hdr);
    if (memcmp(hdr.magic, "TDMPC2PK", 8) != 0 || hdr.version != 1) return fail(TDMPC2_ERR_INVALID, "not a packed weight blob (magic / version)");
    if (!same_model(hdr.cfg, h->cfg))
        return fail(TDMPC2_ERR_INVALID, "packed weights were exported for another model / kernel family / arithmetic "
                    "(latent %d mlp %d A %d path %d precision %d; this handle: %d %d %d %d %d)", hdr.cfg.latent_dim, hdr.cfg.mlp_dim,
                    hdr.cfg.action_dim, hdr.cfg.path, hdr.cfg.precision, h->cfg.latent_dim, h->cfg.mlp_dim, h->cfg.action_dim,
                    h->cfg.path, h->cfg.precision);
    if (hdr.enc_layers > (uint32_t)ENC_MAX_LAYERS) return fail(TDMPC2_ERR_INVALID, "packed weights: bad encoder depth");
    if (h->enc_layers && hdr.enc_layers && h->enc_layers != (int)hdr.enc_layers)
        return fail(TDMPC2_ERR_STATE, "encoder depth changed from %d to %u", h->enc_layers, hdr.enc_layers);
    int rc;
    for (int net : kPackNets) {
        if (!net_in_pack(h, net, hdr.has_target != 0)) continue;
        for (int l = 0; l < 3; ++l)
            if ((rc = ensure_layer_alloc(h, net, l, layer_shape(h, net, l)))) return rc;
    }
    for (uint32_t l = 0; l < hdr.enc_layers; ++l) {  // untrusted header fields: range-check before anything is allocated
        const int wmax = ENC_THREADS * ENC_MAX_PER_THREAD;
        if (hdr.enc_in[l] < 1 || hdr.enc_in[l] > 1 << 20 || hdr.enc_out[l] < 1 || hdr.enc_out[l] > wmax)
            return fail(TDMPC2_ERR_INVALID, "packed weights: encoder layer %u has shape [%d, %d]", l, hdr.enc_out[l], hdr.enc_in[l]);
    }
    for (uint32_t l = 0; l < hdr.enc_layers; ++l)
        if ((rc = ensure_enc_alloc(h, (int)l, hdr.enc_in[l], hdr.enc_out[l]))) return rc;
    std::vector<Seg> segs;
    collect_segments(h, hdr.has_target != 0, (int)hdr.enc_layers, segs);
    // the blob must hold exactly what this handle expects: header + size table + the sum of the EXPECTED segment sizes (the
    // header's own data_bytes is only compared, never trusted; no addition can wrap)
    uint64_t expect = 0;
    for (const Seg &sg : segs) {
        if (sg.bytes > (uint64_t)1 << 40 || expect > (uint64_t)1 << 41) return fail(TDMPC2_ERR_INVALID, "packed weights: implausible segment size");
        expect += sg.bytes;
    }
    const uint64_t need = sizeof(PackHdr) + (uint64_t)segs.size() * sizeof(uint64_t) + expect;
    if (segs.size() != hdr.nseg || hdr.data_bytes != expect || bytes < need)
        return fail(TDMPC2_ERR_INVALID, "packed weights: %llu segments / %llu data bytes in a buffer of %llu bytes; this handle expects %zu segments / "
                    "%llu data bytes (%llu in all)", (unsigned long long)hdr.nseg, (unsigned long long)hdr.data_bytes, (unsigned long long)bytes,
                    segs.size(), (unsigned long long)expect, (unsigned long long)need);
    const char *src = static_cast<const char *>(host_buf);
    const uint64_t *sizes = reinterpret_cast<const uint64_t *>(src + sizeof hdr);
    for (size_t i = 0; i < segs.size(); ++i)
        if (sizes[i] != segs[i].bytes) return fail(TDMPC2_ERR_INVALID, "packed weights: segment %zu has %llu bytes, expected %zu", i, (unsigned long long)sizes[i], segs[i].bytes);
    const char *data = src + sizeof hdr + segs.size() * sizeof(uint64_t);
    hipStream_t st = (hipStream_t)stream;
    for (size_t i = 0; i < segs.size(); ++i) {
        HIP_TRY(hipMemcpyAsync(segs[i].p, data, segs[i].bytes, hipMemcpyHostToDevice, st));
        data += segs[i].bytes;
    }
    HIP_TRY(hipStreamSynchronize(st));  // the host blob may be released on return
    for (int net : kPackNets) {
        if (!net_in_pack(h, net, hdr.has_target != 0)) continue;
        const int heads = layer_shape(h, net, 0).heads;
        for (int hd = 0; hd < heads; ++hd)
            for (int l = 0; l < 3; ++l) net_of(h, net, hd)->l[l].bound = true;
    }
    for (uint32_t l = 0; l < hdr.enc_layers; ++l) h->enc[l].bound = true;
    if (hdr.enc_layers) h->enc_layers = (int)hdr.enc_layers;
    return TDMPC2_OK;
}

int tdmpc2_plan_export_noise(tdmpc2_plan_t *h, int env_first, int n_envs, uint64_t seed, uint32_t call,
                             const tdmpc2_noise_out *out, void *stream) {
    if (!h || !out) return fail(TDMPC2_ERR_INVALID, "null argument");
    if (env_first < 0 || n_envs < 1) return fail(TDMPC2_ERR_INVALID, "environment range [%d, +%d)", env_first, n_envs);
    ENTER_ON(h, stream);
    const tdmpc2_plan_cfg &c = h->cfg;
    NoiseExportParams p{};
    p.e0 = env_first; p.n = n_envs; p.H = c.horizon; p.N = c.num_samples; p.P = c.num_pi_trajs; p.A = c.action_dim;
    p.K = c.num_elites; p.I = c.iterations; p.nq = c.num_q; p.hp = (c.action_dim + 15) / 16 * 8;
    p.seed = seed; p.call = call; p.o = *out;
    hipLaunchKernelGGL(k_export_noise, dim3(2048), dim3(256), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return TDMPC2_OK;
}

int tdmpc2_plan_call_counter(const tdmpc2_plan_t *h, uint32_t *next_call) {
    if (!h || !next_call) return fail(TDMPC2_ERR_INVALID, "null argument");
    *next_call = h->call;
    return TDMPC2_OK;
}

int tdmpc2_plan_set_call_counter(tdmpc2_plan_t *h, uint32_t next_call) {
    if (!h) return fail(TDMPC2_ERR_INVALID, "null handle");
    ENTER(h);
    h->call = next_call;
    return TDMPC2_OK;
}

int tdmpc2_plan_fault_word(tdmpc2_plan_t *h, uint32_t *dst_dev, void *stream) {
    if (!h || !dst_dev) return fail(TDMPC2_ERR_INVALID, "null argument");
    ENTER_ON(h, stream);
    hipStream_t st = (hipStream_t)stream;
    // word 0 of the host-mapped error line, copied IN STREAM ORDER: what the kernels enqueued so far have raised when the copy runs
    if (h->cl_err_dev) HIP_TRY(hipMemcpyAsync(dst_dev, h->cl_err_dev, 4, hipMemcpyDefault, st));
    else HIP_TRY(hipMemsetAsync(dst_dev, 0, 4, st));
    return TDMPC2_OK;
}

int tdmpc2_plan_take_fault(tdmpc2_plan_t *h, int *faults) {
    if (!h || !faults) return fail(TDMPC2_ERR_INVALID, "null argument");
    ENTER(h);
    (void)fault_poll(h);
    *faults = h->faults;
    h->faults = 0;
    return TDMPC2_OK;
}

int tdmpc2_plan_fault_info(tdmpc2_plan_t *h, tdmpc2_fault_info *info) {
    if (!h || !info) return fail(TDMPC2_ERR_INVALID, "null argument");
    ENTER(h);
    info->faults_total = h->faults_total;
    info->rearms = h->rearms;
    info->degraded = h->degraded ? 1 : 0;
    info->clean_calls = h->clean_calls;
    info->rearm_after = h->rearm_after;
    info->seconds_since_fault = h->faults_total
        ? std::chrono::duration<double>(std::chrono::steady_clock::now() - h->last_fault).count() : -1.0;
    return TDMPC2_OK;
}

int tdmpc2_plan_set_tuning(tdmpc2_plan_t *h, int key, int value) {
    if (!h) return fail(TDMPC2_ERR_INVALID, "null handle");
    if (key == TDMPC2_TUNE_ROWS_PER_WORKGROUP) {
        if (value != 0 && value != 32 && value != 64) return fail(TDMPC2_ERR_INVALID, "rows per workgroup must be 0, 32 or 64");
        h->force_rows = value;
        return TDMPC2_OK;
    }
    if (key == TDMPC2_TUNE_CLUSTER) {
        if (value < 0 || value > 2) return fail(TDMPC2_ERR_INVALID, "cluster must be 0 (never), 1 (whenever the call fits) or 2 (auto)");
        h->user_cluster_mode = value;  // an explicit setting also re-arms the handle at once
        h->degraded = false;
        apply_modes(h);
        return TDMPC2_OK;
    }
    if (key == TDMPC2_TUNE_FUSE_LN) {
        if (value < 0 || value > 1) return fail(TDMPC2_ERR_INVALID, "fuse_ln must be 0 or 1");
        h->user_fuse_ln = value != 0 && h->lay.stats != nullptr;
        h->degraded = false;
        apply_modes(h);
        return TDMPC2_OK;
    }
    if (key == TDMPC2_TUNE_REARM_AFTER) {
        if (value < 0) return fail(TDMPC2_ERR_INVALID, "rearm_after must be >= 0 (0: never)");
        h->rearm_after = h->rearm_base = value;
        return TDMPC2_OK;
    }
    if (key == TDMPC2_TUNE_SAFE_ONCE) {
        // the next whole plan (tdmpc2_plan_run*, or shard_begin .. the last shard_refit) runs on the paths without inter-workgroup
        // waits; the caller's settings, the downgrade state and the re-arm counter are not touched (dist.sharded_plan's re-plan)
        if (value < 0 || value > 1) return fail(TDMPC2_ERR_INVALID, "safe_once must be 0 or 1");
        h->safe_once = value != 0;
        apply_modes(h);
        return TDMPC2_OK;
    }
    if (key == TDMPC2_TUNE_KSPLIT) {
        if (value < 0 || value > 2) return fail(TDMPC2_ERR_INVALID, "ksplit must be 0 (never), 1 (always) or 2 (few-tile launches)");
        h->lay.ksplit = value;
        {
            DevGuard dev_(h->cfg.device);
            return ksws_ensure(h);  // (a bigger workspace when the mode needs one)
        }
    }
    if (key >= TDMPC2_TUNE_EXPERT && key < TDMPC2_TUNE_EXPERT + LK_COUNT) {
        // measurement knobs of the layered family's tile choice (handle.h: LayKnob; include/tdmpc2_plan.h: tdmpc2_expert_knob).
        // INT32_MIN restores the default.  They move work between kernels that compute the same values to fp32 round-off.
        h->lay.knob[key - TDMPC2_TUNE_EXPERT] = value == INT32_MIN ? LAY_KNOB_DEFAULTS[key - TDMPC2_TUNE_EXPERT] : value;
        return TDMPC2_OK;
    }
    if (key == TDMPC2_TUNE_WAIT_US) {
        if (value < 100 || value > 10000000) return fail(TDMPC2_ERR_INVALID, "wait_us must be 100 .. 10 000 000");
        if (!h->cl_err_host) return fail(TDMPC2_ERR_STATE, "this handle has no path with inter-workgroup waits");
        // 100 MHz constant clock (s_memrealtime): 100 ticks per microsecond; word 12 of the host-mapped error line (WaitClock)
        __atomic_store_n(h->cl_err_host + ERR_WAIT_TICKS, (unsigned int)std::min<long long>(100LL * value, 0xffffffffLL), __ATOMIC_RELAXED);
        return TDMPC2_OK;
    }
    if (key == TDMPC2_TUNE_FEWROW) {
        if (value < 0 || value > 1) return fail(TDMPC2_ERR_INVALID, "fewrow must be 0 or 1");
        h->lay.mid = value != 0;
        return TDMPC2_OK;
    }
    if (key == TDMPC2_TUNE_FOLD_REFIT) {
        if (value < 0 || value > 2) return fail(TDMPC2_ERR_INVALID, "fold_refit must be 0 (never), 1 (always) or 2 (auto)");
        h->fold_refit = value;
        return TDMPC2_OK;
    }
    return fail(TDMPC2_ERR_INVALID, "unknown tuning key %d", key);
}

int tdmpc2_plan_set_profiling(tdmpc2_plan_t *h, int max_launches) {
    if (!h) return fail(TDMPC2_ERR_INVALID, "null handle");
    ENTER(h);
    h->profiling = max_launches > 0;
    if ((int)h->ev.size() < 2 * max_launches) {
        const size_t old = h->ev.size();
        h->ev.resize(2 * (size_t)max_launches);
        for (size_t i = old; i < h->ev.size(); ++i) HIP_TRY(hipEventCreate(&h->ev[i]));
    }
    h->ev_used = 0;
    return TDMPC2_OK;
}

int tdmpc2_plan_profile_read(tdmpc2_plan_t *h, float *rollout_ms_total, int *rollout_launches) {
    if (!h || !rollout_ms_total || !rollout_launches) return fail(TDMPC2_ERR_INVALID, "null argument");
    ENTER(h);
    float total = 0.f;
    for (int i = 0; i + 1 < h->ev_used; i += 2) {
        HIP_TRY(hipEventSynchronize(h->ev[i + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
        total += ms;
    }
    *rollout_ms_total = total;
    *rollout_launches = h->ev_used / 2;
    h->ev_used = 0;
    return TDMPC2_OK;
}

int tdmpc2_plan_run(tdmpc2_plan_t *h, int n_envs, const float *z0, const float *task_emb, const float *act_mask,
                    const float *disc_pow, float *prev_mean, const uint8_t *t0, int eval_mode, const tdmpc2_noise *tape,
                    uint64_t seed, float *action, const tdmpc2_debug *dbg, void *stream) {
    if (!h) return fail(TDMPC2_ERR_INVALID, "null handle");
    ENTER_ON(h, stream);
    return run_impl(h, n_envs, z0, task_emb, act_mask, disc_pow, prev_mean, t0, eval_mode, tape, seed, action, dbg, stream);
}

}  // extern "C"
namespace {
int run_impl(tdmpc2_plan *h, int n_envs, const float *z0, const float *task_emb, const float *act_mask, const float *disc_pow,
             float *prev_mean, const uint8_t *t0, int eval_mode, const tdmpc2_noise *tape, uint64_t seed, float *action,
             const tdmpc2_debug *dbg, void *stream) {
    if (h) h->in_shard = false;  // a whole plan on this handle abandons whatever sharded plan was left open (an exception in the caller)
    int rc = validate_envs(h, n_envs);
    if (rc) return rc;
    if (!z0 || !disc_pow || !prev_mean || !t0 || !action) return fail(TDMPC2_ERR_INVALID, "null argument");
    const tdmpc2_plan_cfg &c = h->cfg;
    if (c.multitask && (!task_emb || !act_mask)) return fail(TDMPC2_ERR_INVALID, "multitask plan needs task_emb and act_mask");
    if (!c.multitask) { task_emb = nullptr; act_mask = nullptr; }
    if (tape && (!tape->sample_eps || !tape->pi_eps || !tape->qidx || !tape->gumbel_exp ||
                 (c.num_pi_trajs > 0 && !tape->pi_traj_eps) || (!eval_mode && !tape->final_eps)))
        return fail(TDMPC2_ERR_INVALID, "noise tape has null fields");
    hipStream_t st = (hipStream_t)stream;
    struct SafeOnce {  // TDMPC2_TUNE_SAFE_ONCE covers exactly this plan
        tdmpc2_plan *h;
        ~SafeOnce() { if (h->safe_once) { h->safe_once = false; apply_modes(h); } }
    } once{h};
    if (h->lay.on) {
        if ((rc = fault_fresh(h, st))) return rc;
        return lay_run(h, st, n_envs, z0, task_emb, act_mask, disc_pow, prev_mean, t0, eval_mode, tape, seed, action, dbg);
    }
    return fused_run<NetS>(h, st, n_envs, z0, task_emb, act_mask, disc_pow, prev_mean, t0, eval_mode, tape, seed, action, dbg);
}
}  // namespace
extern "C" {

int tdmpc2_plan_estimate_value(tdmpc2_plan_t *h, int n_envs, const float *z0, const float *task_emb,
                               const float *act_mask, const float *disc_pow, const float *actions, const float *pi_eps,
                               const int32_t *qidx, float *value, void *stream) {
    return tdmpc2_plan_estimate_value_trace(h, n_envs, z0, task_emb, act_mask, disc_pow, actions, pi_eps, qidx, value,
                                            nullptr, nullptr, stream);
}

int tdmpc2_plan_estimate_value_trace(tdmpc2_plan_t *h, int n_envs, const float *z0, const float *task_emb,
                                     const float *act_mask, const float *disc_pow, const float *actions,
                                     const float *pi_eps, const int32_t *qidx, float *value, float *trace_tiles,
                                     float *trace_scalars, void *stream) {
    if (!h) return fail(TDMPC2_ERR_INVALID, "null handle");
    ENTER_ON(h, stream);
    int rc = validate_envs(h, n_envs);
    if (rc) return rc;
    if (!z0 || !disc_pow || !actions || !pi_eps || !qidx || !value) return fail(TDMPC2_ERR_INVALID, "null argument");
    const tdmpc2_plan_cfg &c = h->cfg;
    if (c.multitask && (!task_emb || !act_mask)) return fail(TDMPC2_ERR_INVALID, "multitask needs task_emb and act_mask");
    if (!c.multitask) { task_emb = nullptr; act_mask = nullptr; }
    hipStream_t st = (hipStream_t)stream;
    const int E = n_envs, N = c.num_samples, A = c.action_dim;
    if ((rc = fault_fresh(h, st))) return rc;
    if (h->lay.on) {
        if (trace_tiles) return fail(TDMPC2_ERR_UNSUPPORTED, "the layered path dumps trace_scalars only");
        if ((rc = lay_setup(h, st, E, task_emb, nullptr, nullptr, false))) return rc;
        if ((rc = lay_cvec(h, st, E, z0))) return rc;
        if ((rc = lay_set_qidx(h, st, E, qidx, 2L, c.num_q, 0, 0, 0, h->lay.qidx))) return rc;
        return lay_estimate_value(h, st, E, z0, act_mask, disc_pow, actions, pi_eps, (long)N * A, h->lay.qidx, 0, 0, 0, value,
                                  trace_scalars);
    }
    return fused_estimate_value<NetS>(h, st, E, z0, task_emb, act_mask, disc_pow, actions, pi_eps, qidx, value, trace_tiles,
                                      trace_scalars);
}

int tdmpc2_plan_refit(tdmpc2_plan_t *h, int n_envs, float *value, const float *actions, const float *act_mask,
                      float *mean, float *std, float *score, int32_t *elite_idx, void *stream) {
    if (!h) return fail(TDMPC2_ERR_INVALID, "null handle");
    if (n_envs < 1 || n_envs > h->cfg.max_envs) return fail(TDMPC2_ERR_INVALID, "n_envs=%d outside [1, %d]", n_envs, h->cfg.max_envs);
    if (!value || !actions) return fail(TDMPC2_ERR_INVALID, "null argument");
    ENTER_ON(h, stream);
    const tdmpc2_plan_cfg &c = h->cfg;
    hipStream_t st = (hipStream_t)stream;
    int refit_stage = 0;
    const size_t refit_lds = refit_lds_bytes(c.num_samples, c.num_elites, c.horizon, c.action_dim, &refit_stage);
    RefitParams fp{};
    fp.Nvalid = c.num_valid_samples; fp.E = n_envs; fp.N = c.num_samples; fp.H = c.horizon; fp.A = c.action_dim; fp.K = c.num_elites; fp.last = 0; fp.stage = refit_stage;
    fp.temperature = c.temperature; fp.min_std = c.min_std; fp.max_std = c.max_std;
    fp.value = value; fp.actions = actions; fp.act_mask = c.multitask ? act_mask : nullptr;
    fp.mean = mean ? mean : h->mean; fp.std = std ? std : h->std; fp.score = score; fp.elite_idx = elite_idx;
    hipLaunchKernelGGL(k_refit, dim3(n_envs), dim3(refit_threads(c.num_samples)), refit_lds, st, fp);
    HIP_TRY(hipGetLastError());
    return TDMPC2_OK;
}

}  // extern "C"
