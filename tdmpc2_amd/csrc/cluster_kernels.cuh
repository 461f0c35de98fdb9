// Single-plan latency: one CEM iteration's rollouts (TDMPC2._estimate_value, tdmpc2/tdmpc2.py:122-136, under the sampling
// of tdmpc2.py:176-181) with every 512-wide layer split over a CLUSTER of 8 workgroups per 32-row sample tile.
//
// Why: a single environment's plan (evaluate.py:80 -> tdmpc2.py:111) has 512 sample rows = 16 row tiles.  ks_rollout gives
// each tile to one workgroup, which then streams every layer's whole weight matrix (1 MB in hi / lo f16 form) through ONE
// CU's L1 at 64 B/clk: 23 MB per CEM iteration, 172 us of the 314 us the iteration takes, on 16 of 256 CUs
// (profiles/README.md).  Here 8 workgroups on 8 CUs of one XCD share a row tile: member r computes output features
// [64 r, 64 r + 64) of a layer (its 8 waves: 2 feature tiles x 4 quarters of the contraction, summed through LDS), writes
// the 32 x 64 raw sums to the cluster's exchange tile in L2, waits for the other seven and reads all 32 x 512 back -- in
// the register order of ks_rollout's epilogue, so LayerNorm / Mish / SimNorm / hi-lo split are epi_t itself, run
// redundantly by every member on the full row (LayerNorm needs it) into the member's own LDS operand tile.  The narrow
// heads (two-hot reward / Q, policy) are computed by every member from its own tile.  A member streams 1/8 of the weights.
//
// Hand-over protocol (measured stand-alone first: tools/probes/cluster_probe.hip, profiles/README.md r02k): data leave as
// agent-scope write-through stores and are read with agent-scope loads (correct wherever the members run; 2.1 us per
// exchange when they share an XCD, 2.8 us across XCDs); arrival = one word per member holding the phase number, written
// after every wave drained its stores and the workgroup barrier, polled by 8 lanes.  Phase numbers grow monotonically
// through the I rollout launches of a plan (ks_setup zeroes the words), so a launch needs no reset and a captured
// hipGraph replays.  Every wait is BOUNDED: a member that gives up raises the handle's error word (host-mapped; the next
// API call fails loudly and turns the cluster path off) instead of hanging the GPU.
// Members are placed on one XCD by construction of the block index (workgroups go to XCDs round-robin): blocks
// {64 g + x + 8 r : r = 0..7} form cluster 8 g + x.  Only speed depends on that.
// Co-residency: a cluster launch has at most one workgroup per CU (<= num_cus workgroups, 147 KB LDS each); the host
// serialises cluster launches of different streams (an event per device) so that two partially resident launches cannot
// wait for each other.
//
// Exchange tiles S0..S3 per cluster ([32 x 512] fp32 each); a tile exchanged at barrier j and read before the reader's arrival
// at barrier j + 1 may be rewritten from phase j + 2 on (the writer has passed barrier j + 1, so every member arrived there):
//   step t:   phase 4t+1  dyn.l0 -> S0, rew.l0 -> S1 (one barrier for both; S0 stays parked)      epilogue rew.l0 <- S1
//             phase 4t+2  rew.l1 -> S2                                                            epilogue, reward head
//                         epilogue dyn.l0 <- S0 (parked)
//             phase 4t+3  dyn.l1 -> S1                                                            epilogue
//             phase 4t+4  dyn.l2 -> S2                                                            SimNorm epilogue
//   (cluster 0 of a plan, first launch: the policy-prior trajectories, tdmpc2.py:154-160, are the rollout of the tile's first
//    P rows under a_t = pi(z_t): at the top of step t  pi.l0 -> S0, pi.l1 -> S3, policy head -> the P rows' action columns and
//    `actions`, z_t back into the tile; two more hand-overs per step -- phases are only required to grow)
//   (EP = 1, episodic models: termination(z_t), world_model.py:132-141 / tdmpc2.py:133-134, t >= 1: term.l0 -> S5 right behind the
//    step's first layers -- same z_t columns --, its epilogue, term.l1 -> S3 and the one-logit head after the reward head;
//    termination(z_H): term.l0 -> S5 in front of the policy layers, term.l1 -> S0 behind the policy head)
//   value:    phase 4H+1  pi.l0 -> S0;  4H+2  pi.l1 -> S1;  policy head;  z_H back into the tile
//             phase 4H+3  q1.l0 -> S2 (parked), q0.l0 -> S3;  4H+4  q0.l1 -> S0;  head;  epilogue q1.l0 <- S2
//             phase 4H+5  q1.l1 -> S1;  head.
#pragma once

// (bounded waits: WaitClock, common.cuh -- 5 ms by the wall clock)
__host__ __device__ constexpr int cl_phases(int H) { return 8 * H + 7; }  // hand-overs per launch (upper bound: policy prior + termination)
__host__ __device__ constexpr int cl_heads(int H) { return 3 * H + 4; }  // narrow heads per launch (upper bound): H reward, H policy prior, H + 1 termination, policy, two Q

struct ClState {
    float *xbuf;       // this cluster's CL_SLOTS exchange tiles
    unsigned *flags;   // this cluster's arrival words [CL]
    unsigned *err;     // the handle's error word
    float *red;        // LDS [2 layers][8 waves][4][64 lanes][4]: contraction-quarter partial sums
    int *dead;         // LDS flag: a wait of this member timed out -- it stops waiting (and the plan is reported invalid)
    int *fast;         // LDS flag: all members of the cluster run on one XCD (set at the launch's first barrier)
    int rank;
    unsigned xcc;      // this workgroup's XCC_ID
    unsigned phase;    // last phase this member arrived at
    unsigned hphase;   // narrow heads so far (arrival words 8..15 of the cluster)
    bool learned;      // the launch's first hand-over is behind us (x.fast is valid)
    bool mute;         // test hook (TDMPC2_CLUSTER_FAULT): this member never signals -- the others must time out, not hang
    int pub = 0;       // the next exchange / head tile is also read by ANOTHER cluster (cluster2_kernels.cuh): write-through stores
};

__device__ __forceinline__ void cl_st16(float *p, f32x4 v, int fast) {
    // members on one XCD share its L2: a plain (write-through-to-L2) store is visible to the others' agent-scope loads;
    // otherwise the line has to leave the XCD: agent-scope write-through
    // s_nop 1 in the same statement: the store reads its 128 bits of data after issue, a VALU write of those registers needs
    // 2 wait states behind it on gfx940+, and the hazard recognizer does not look inside inline asm (layered_wide.cuh: gw_st_sc1)
    if (fast) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// Agent-scope 16-byte loads AND THEIR WAIT IN ONE asm STATEMENT: the compiler does not know that an inline-asm load lands
// later -- a destination register it copied or spilled between a load statement and a separate wait statement would be read
// before the data is there (layered_wide.cuh: gw_ld4 met exactly that under register pressure).
__device__ __forceinline__ void cl_ld16x2(f32x4 &a, f32x4 &b, const float *pa, const float *pb) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b)
                 : "v"(pa), "v"(pb)
                 : "memory");
}
// eight of them 1 KiB apart (p, p + 256 floats, ...): two address pairs + the instruction's 12-bit offset
__device__ __forceinline__ void cl_ld16x8(f32x4 (&v)[8], const float *p) {
    const float *p4 = p + 1024;
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:3072 sc1\n\t"
                 "global_load_dwordx4 %4, %9, off sc1\n\tglobal_load_dwordx4 %5, %9, off offset:1024 sc1\n\t"
                 "global_load_dwordx4 %6, %9, off offset:2048 sc1\n\tglobal_load_dwordx4 %7, %9, off offset:3072 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p), "v"(p4)
                 : "memory");
}

// What a weight request needs to know of a layer, BY VALUE: a pointer to (an element of) the kernel-argument struct --
// let alone one selected at run time, p.q[q1] -- makes the compiler copy the 2.5 KB struct to scratch and chase the
// layer's fields through it (measured: 1.47 -> 2.1 ms per plan)
struct WRef {
    const _Float16 *wp = nullptr;  // null: no layer
    int KB = 0;
};
__device__ __forceinline__ WRef wref(const LayerS &ly) { return WRef{ly.wp, ly.KB}; }

// One wave's share of a layer: wave w takes feature tile ft = w >> 2 of the member's two and quarter kq = w & 3 of the
// k-blocks [kb0, kb1): column tile ct, k-blocks [k0, k0 + n).  N k-blocks as STRAIGHT-LINE code: every weight fragment is
// requested up front and block j's MFMAs wait for exactly its own two loads, so the matrix pipe runs under the rest of the
// weight stream (with a run-time trip count the compiler's wait before the first MFMA covers all loads: 5.2 k cycles per
// layer, profiles/README.md r02s).  Raw partial sums in the C layout of kloop_s (weight fragment = A operand: lane (j, hh)
// holds features 8 m + 4 hh + r of sample row j in register 4 m + r); three independent accumulator chains.
template <class CT, int N>
__device__ __forceinline__ void cl_kshare_n(const CT &c, const char *u, const _Float16 *a0p, f32x16 &out) {
    unsigned voff = (unsigned)c.lane * 16u;
    asm volatile("" : "+v"(voff));
    f16x8 wh[N > 0 ? N : 1], wl[N > 0 ? N : 1];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        wh[j] = ldw(u + (size_t)j * 2048, voff, 0);
        wl[j] = ldw(u + (size_t)j * 2048, voff, 1024);
    }
    // pins the requests here: the machine scheduler otherwise sinks every load to its first use (load, wait, MFMA, load ...)
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    f16x8 ah, al;
    if (N > 0) {
        ah = *reinterpret_cast<const f16x8 *>(a0p);
        al = *reinterpret_cast<const f16x8 *>(a0p + CT::SH);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {  // the NEXT block's activation fragments are read in front of this block's MFMAs
        f16x8 nh = ah, nl = al;
        if (j + 1 < N) {
            nh = *reinterpret_cast<const f16x8 *>(a0p + (j + 1) * 16);
            nl = *reinterpret_cast<const f16x8 *>(a0p + CT::SH + (j + 1) * 16);
        }
        acc[0] = SPLIT_MFMA(wh[j], ah, acc[0]);
        acc[1] = SPLIT_MFMA(wl[j], ah, acc[1]);
        acc[2] = SPLIT_MFMA(wh[j], al, acc[2]);
        __builtin_amdgcn_sched_barrier(0);
        ah = nh;
        al = nl;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) out[e] = acc[0][e] + (acc[1][e] + acc[2][e]);
}
template <class CT>
__device__ __forceinline__ void cl_kshare(const CT &c, const ClState &x, const WRef ly, int kb0, int kb1, f32x16 &out) {
    static_assert(CT::ARITH == 0 && CT::NST == 1, "cluster path: split arithmetic, 32-row tiles");
    const int ft = c.wave >> 2, kq = c.wave & 3;
    const int nk = kb1 - kb0;
    const int k0 = kb0 + (nk * kq) / 4;
    const int n = kb0 + (nk * (kq + 1)) / 4 - k0;  // wave-uniform: 8 or 9 (whole layers), 0 .. 1 (action columns at t = 0)
    const int ct = 2 * x.rank + ft;
    const char *u = reinterpret_cast<const char *>(ly.wp) + ((size_t)ct * ly.KB + k0) * 2048;
    const int i = c.lane & 31, hh = c.lane >> 5;
    const _Float16 *a0p = c.act + i * c.RSH + 8 * hh + k0 * 16;
    if (n == 8) cl_kshare_n<CT, 8>(c, u, a0p, out);
    else if (n == 9) cl_kshare_n<CT, 9>(c, u, a0p, out);
    else if (n == 1) cl_kshare_n<CT, 1>(c, u, a0p, out);
    else if (n == 0) cl_kshare_n<CT, 0>(c, u, a0p, out);
    else {  // any other split (not reached with latent 512 and action paddings 16 .. 64): block by block
#pragma unroll
        for (int e = 0; e < 16; ++e) out[e] = 0.f;
        for (int j = 0; j < n; ++j) {
            f32x16 part;
            cl_kshare_n<CT, 1>(c, u + (size_t)j * 2048, a0p + j * 16, part);
#pragma unroll
            for (int e = 0; e < 16; ++e) out[e] += part[e];
        }
    }
}

// partial sums of this wave -> LDS, [layer][wave][m][lane] float4
template <class CT>
__device__ __forceinline__ void cl_to_red(const CT &c, const ClState &x, int layer, const f32x16 &v) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x4 q;
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] = v[4 * m + r];
        *reinterpret_cast<f32x4 *>(x.red + ((size_t)((layer * 8 + c.wave) * 4 + m) * 64 + c.lane) * 4) = q;
    }
}

// arrive at the next phase (all data stores of this workgroup are issued), wait for the other members.  The arrival word
// carries the member's XCC_ID in its top byte: at the first barrier of a launch the pollers learn whether the cluster
// shares an XCD (x.fast), which later exchanges use to pick the store flavour.
template <class CT>
__device__ __forceinline__ void cl_barrier(const CT &c, ClState &x, bool learn) {
#ifdef CL_ABL_NO_WAIT  // timing experiment: no hand-over (results are wrong)
    x.phase += 1;
    __syncthreads();
    return;
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores are acknowledged (by the L2 / the fabric)
    __syncthreads();
    x.phase += 1;
    if (c.tid == 0 && !x.mute) __hip_atomic_store(x.flags + x.rank, x.phase | (x.xcc << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c.tid < CL && !*x.dead) {
        WaitClock wc;
        unsigned v;
        while (((v = __hip_atomic_load(x.flags + c.tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffffffu) < x.phase) {
            if (wc.expired(x.err)) {  // host-mapped word: a plain system-scope store
                raise_fault(x.err, 1u);
                *x.dead = 1;
                v = 0xff000000u;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (learn) {
            const unsigned long long same = __ballot((v >> 24) == x.xcc);
            if (c.tid == 0) *x.fast = (same & 0xffull) == 0xffull;
        }
    }
    __syncthreads();
}

// The contraction of one or two layers over the member's operand tile -> the cluster's exchange tile(s), one barrier.
template <class CT>
__device__ __forceinline__ void cl_gemm(const CT &c, ClState &x, const WRef la, int slot_a, const WRef lb, int slot_b,
                                        int kb0, int kb1) {
    const bool learn = !x.learned;  // the first hand-over of a launch: safe stores, and the pollers compare XCC ids
    x.learned = true;
#ifdef CL_ABL_NO_GEMM  // timing experiment
    kb1 = kb0;
#endif
    {
        f32x16 v;
        cl_kshare(c, x, la, kb0, kb1, v);
        cl_to_red(c, x, 0, v);
        if (lb.wp) {  // (both layers' shares in flight at once: 144 VGPRs, spills; measured)
            cl_kshare(c, x, lb, kb0, kb1, v);
            cl_to_red(c, x, 1, v);
        }
    }
    TIMER_MARK(c, T_KLOOP)
    __syncthreads();
    // thread (ft', m, lane') sums the four quarters of one float4 and stores it where wave `rank` of ks_rollout's register
    // order would park it: idx4 = (rank * 2 + ft') * 4 + m
    const int rft = c.tid >> 8, rm = (c.tid >> 6) & 3, rl = c.tid & 63;
    const int nl = lb.wp ? 2 : 1;
    const int fast = (learn || x.pub) ? 0 : *x.fast;
    for (int layer = 0; layer < nl; ++layer) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 pq = *reinterpret_cast<const f32x4 *>(x.red + ((size_t)((layer * 8 + rft * 4 + q) * 4 + rm) * 64 + rl) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] += pq[r];
        }
        float *dst = x.xbuf + (size_t)(layer == 0 ? slot_a : slot_b) * CL_TILE + ((size_t)((x.rank * 2 + rft) * 4 + rm) * 64 + rl) * 4;
        cl_st16(dst, s, fast);
    }
    TIMER_MARK(c, T_PARK)
    cl_barrier(c, x, learn);
    TIMER_MARK(c, T_CL_WAIT)
}

// this lane's bias values in accumulator order (what epi_t's HASBIAS path reads): requested in front of the exchange-tile
// loads, so that their latency overlaps
struct ClBias {
    f32x4 b[2][4];
};
template <class CT>
__device__ __forceinline__ void cl_bias_load(const CT &c, const float *bias, ClBias &bb) {
    const int poff = 64 * c.wave + 4 * (c.lane >> 5);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int m = 0; m < 4; ++m) bb.b[ft][m] = *reinterpret_cast<const f32x4 *>(bias + poff + 32 * ft + 8 * m);
}

// exchange tile -> raw sums in ks_rollout's accumulator layout (wave w: features [64 w, 64 w + 64) = member w's slice)
template <class CT>
__device__ __forceinline__ void cl_unpark(const CT &c, const ClState &x, int slot, f32x16 (&acc)[1][2]) {
    const float *src = x.xbuf + (size_t)slot * CL_TILE + ((size_t)(c.wave * 2) * 4 * 64 + c.lane) * 4;
    f32x4 v[8];
    cl_ld16x8(v, src);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[0][ft][4 * m + r] = v[ft * 4 + m][r];
    TIMER_MARK(c, T_TILE)
}

// Epilogue of a layer whose raw sums are in exchange tile `slot`: ACT(LayerNorm(sums * oscale + bias)) -> operand tile.
// `next`: the LayerNorm parameters of the epilogue after this one (gb protocol of fused_kernels.cuh).  The layer's scales
// and bias pointer come BY VALUE: a pointer to (an element of) the kernel-argument struct that the compiler cannot fold
// makes it copy the 2.5 KB struct to scratch and read every parameter from there.
template <int ACT, class CT>
__device__ __forceinline__ void cl_epi(const CT &c, const ClState &x, int slot, const float *oscale, const float *ascale,
                                       const float *bias, GB next, float *zcopy = nullptr) {
    if (next.g) gb_prefetch(c, next.g, next.b);
    const float osc = *oscale, asc = *ascale;
    ClBias bb;
    cl_bias_load(c, bias, bb);
    f32x16 acc[1][2];
    cl_unpark(c, x, slot, acc);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[0][ft][4 * m + r] = fmaf(acc[0][ft][4 * m + r], osc, bb.b[ft][m][r]);
#ifdef CL_ABL_NO_EPI  // timing experiment: raw sums straight into the tile
    regs_to_tile(c, acc, 1.f);
    if (zcopy) park(c, acc, zcopy);
    __syncthreads();
#else
    epi_t<ACT, CT, false>(c, acc, osc, asc, bias, GB{}, zcopy);
#endif
    epi_barrier(c);
    TIMER_MARK(c, T_EPI)
}

// a whole hidden layer: contraction over the member's tile, exchange, epilogue
template <int ACT, class CT>
__device__ __forceinline__ void cl_layer(const CT &c, ClState &x, const WRef ly, const float *oscale, const float *ascale,
                                         const float *bias, int kb0, int kb1, int slot, GB next, float *zcopy = nullptr) {
    if (next.g) gb_prefetch(c, next.g, next.b);
    cl_gemm(c, x, ly, slot, WRef{}, 0, kb0, kb1);
    cl_epi<ACT>(c, x, slot, oscale, ascale, bias, GB{}, zcopy);
}
// shorthand for the call sites: the fields of layer L that cl_layer / cl_epi take by value
#define CL_L(L) wref(L), (L).oscale, (L).ascale
#define CL_E(L) (L).oscale, (L).ascale


// A narrow head (two-hot reward / Q: 4 column tiles of 32 logits; policy: <= 4) across the cluster: member m < CT computes
// column tile m -- its 8 waves take 4 k-blocks each, summed through LDS -- and writes the 32 x 32 logits to the cluster's
// head tile; the consumers (member 0 for a two-hot head: only it needs the values; every member for the policy head) wait
// for the CT arrival words and copy the logits into their staging view, where the row math of fused_kernels.cuh takes
// over.  One member running the whole head -- 4 waves x 32 k-blocks, 4 blocks in flight -- took 13.4 k cycles, as much as
// two hidden layers' contractions (profiles/README.md r02m); the member alone on all 8 waves with every fragment in flight
// was slower still (r02v).  Members that are neither producer nor consumer walk through.
// Reuse of the head tile: two heads are always separated by a cluster barrier, which a consumer reaches after its reads.
template <class CT>
__device__ __forceinline__ void cl_head_logits(const CT &c, ClState &x, const LayerS &ly, bool consume, int slot = 4) {
    static_assert(CT::ZKB == 32, "8 waves x 4 k-blocks");
#ifdef CL_ABL_NO_HEAD  // timing experiment: results are wrong
    return;
#endif
    x.hphase += 1;
    float *hbuf = x.xbuf + (size_t)slot * CL_TILE;
    unsigned *hflags = x.flags + 8;
    const bool produce = x.rank < ly.CT;
    if (produce) {
        f32x16 acc[1];
        kloop_tile_s(c, ly, x.rank, 4 * c.wave, 4 * c.wave + 4, acc);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) x.red[(size_t)(c.wave * 16 + reg) * 64 + c.lane] = acc[0][reg];
    }
    __syncthreads();  // partials in LDS; every wave is past its reads of the operand tile
    if (produce) {
        const float osc = *ly.oscale;
        const int lane = c.tid & 63, col = lane & 31;
        const float bv = ly.bias[x.rank * 32 + col];
        const int fast = x.pub ? 0 : *x.fast;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int reg = (c.tid >> 6) + 8 * u;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += x.red[(size_t)(w * 16 + reg) * 64 + lane];
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            float *dst = hbuf + row * 128 + x.rank * 32 + col;
            const float v = fmaf(s, osc, bv);
            if (fast) asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(v) : "memory");
            else asm volatile("global_store_dword %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c.tid == 0) __hip_atomic_store(hflags + x.rank, x.hphase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!consume) return;
    if (c.tid < ly.CT && !*x.dead) {
        WaitClock wc;
        while (__hip_atomic_load(hflags + c.tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < x.hphase) {
            if (wc.expired(x.err)) {
                raise_fault(x.err, 1u);
                *x.dead = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    {  // head tile -> staging view: thread t takes row t >> 4, columns 4 (t & 15) .. +3 and 64 + 4 (t & 15) .. +3
        const int row = c.tid >> 4, c4 = (c.tid & 15) * 4;
        f32x4 v0, v1;
        cl_ld16x2(v0, v1, hbuf + row * 128 + c4, hbuf + row * 128 + 64 + c4);
        float *f = c.f32() + row * CT::RSF();
        *reinterpret_cast<f32x4 *>(f + c4) = v0;
        *reinterpret_cast<f32x4 *>(f + 64 + c4) = v1;
    }
    __syncthreads();
}


// two-hot head: the value is returned on member 0 only (the others get 0 and never use it)
template <class CT>
__device__ __forceinline__ float cl_head_twohot(const CT &c, ClState &x, const LayerS &ly, const float *bins, int num_bins) {
    const bool consume = x.rank == 0;
    cl_head_logits(c, x, ly, consume);
    if (!consume) return 0.f;
#ifdef CL_ABL_NO_HEAD
    return 1.f;
#endif
    const float r = twohot_rows_s(c, bins, num_bins);
    __syncthreads();
    return r;
}

// termination head (one logit): 1 if sigmoid(logit) > 0.5, on member 0 (the others get 0 and never use it)
template <class CT>
__device__ __forceinline__ float cl_head_term(const CT &c, ClState &x, const LayerS &ly) {
    const bool consume = x.rank == 0;
    cl_head_logits(c, x, ly, consume);
    if (!consume) return 0.f;
    const bool live = (c.tid >> 3) < CT::TROWS;
    const float v = c.f32()[(live ? c.tid >> 3 : 0) * CT::RSF()];
    const float pr = 1.f / (1.f + expf(-v));
    __syncthreads();
    return pr > 0.5f ? 1.f : 0.f;
}

template <int APAD, int EP>
__global__ __launch_bounds__(NTHREADS, 2) void ks_rollout_cl(RolloutParamsT<NetS> p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int s_is_last, s_dead, s_fast;
    typedef CtxT<APAD, 1, 8, 0> CT;
    constexpr int TROWS = CT::TROWS, NTHR = CT::NTHR;
    constexpr int ZKB16 = CT::ZKB;
    const int tid = threadIdx.x;
    // blocks {64 g + xc + 8 r : r = 0..7} = cluster 8 g + xc (one XCD under round-robin dispatch)
    const int xc = blockIdx.x & 7, bq = blockIdx.x >> 3;
    const int rank = bq & 7, cl = (bq >> 3) * 8 + xc;
    if (cl >= p.E * p.tiles) return;  // the whole cluster leaves
    const int e = cl / p.tiles, tile = cl % p.tiles;
    CT c{reinterpret_cast<_Float16 *>(smem), smem + TROWS * CT::RSF(), smem + TROWS * CT::RSF() + 1024, tid,
         __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63};
    float *sm_mean = smem + TROWS * CT::RSF() + 2048;
    float *sm_std = sm_mean + p.H * p.A;
    ClState x{p.cl_xbuf + (size_t)cl * CL_SLOTS * CL_TILE, p.cl_flags + (size_t)cl * CL_FLAG_STRIDE, p.cl_err,
              smem + TROWS * CT::RSF() + 2048 + ((2 * p.H * p.A + 3) & ~3), &s_dead, &s_fast, rank, 0u,
              (unsigned)(p.iter * cl_phases(p.H)), (unsigned)(p.iter * cl_heads(p.H)), false, p.cl_fault != 0 && cl == 0 && rank == 7};
    {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        x.xcc = xcc & 0xfu;
    }
    if (tid == 0) {
        s_dead = 0;
        s_fast = 0;
    }
    const int row0 = tile * TROWS;
    const float *mask = p.act_mask ? p.act_mask + (size_t)e * p.A : nullptr;
    const float *disc = p.disc_pow + (size_t)e * (p.H + 1);
    const int KBA = ZKB16 + p.Apad / CT::KBLK;
    float *zs = p.cl_zs + (size_t)(cl * CL + rank) * TROWS * WIDTH;
    const bool live = (tid >> 3) < TROWS;

    for (int idx = tid; idx < p.H * p.A; idx += NTHR) {
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
    // the policy-prior trajectories of this plan: tile 0's first P rows, first launch (host: P <= 32)
    const bool pifold = p.pi_fold && p.iter == 0 && tile == 0;
    if (pifold) gb_prefetch(c, p.pi.l[0].g, p.pi.l[0].b);
    else gb_prefetch(c, p.rew.l[0].g, p.rew.l[0].b);
    tile_broadcast_row_s(c, p.z0 + (size_t)e * WIDTH);  // z_0 in every row (tdmpc2.py:163); step 0 contracts the full [z | a] range
    epi_barrier(c);

    float G = 0.f;
    float termv = 0.f;  // EP: 1 once any latent of this row's trajectory was classified terminal (tdmpc2.py:133-134)
    TIMER_START(c)
    for (int t = 0; t < p.H; ++t) {
        const bool term_step = EP && t > 0;
        // ---- actions of step t (tdmpc2.py:176-181): every member fills its own tile; member 0 also writes them out
        {
            float *ag = p.actions + ((size_t)e * p.H + t) * p.N * p.A;
            const int hp = p.Apad / 2;
            for (int idx = tid; idx < TROWS * hp; idx += NTHR) {
                const int row = idx / hp, a0 = 2 * (idx % hp);
                const int n = row0 + row;
                float v[2] = {0.f, 0.f};
                const bool sampled = !(p.given_actions || n < p.P);
                float z[2] = {0.f, 0.f};
                if (sampled && a0 < p.A && !p.sample_eps) {
                    const unsigned pair = (unsigned)(((size_t)t * (p.N - p.P) + (n - p.P)) * hp + a0 / 2);
                    rng_normal2(p.seed, p.call, SITE_SAMPLE, p.iter, e, pair, z[0], z[1]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int a = a0 + u;
                    if (a < p.A) {
                        if (!sampled) {
                            v[u] = pifold ? 0.f : ag[(size_t)n * p.A + a];  // pifold: filled in by the policy head below
                        } else {
                            float r = z[u];
                            if (p.sample_eps)
                                r = p.sample_eps[(size_t)e * p.sample_eps_estride +
                                                 (unsigned)(((size_t)t * (p.N - p.P) + (n - p.P)) * p.A + a)];
                            v[u] = sample_action(sm_mean[t * p.A + a], sm_std[t * p.A + a], r);
                        }
                        if (mask && !p.given_actions) v[u] *= mask[a];
                        if (!p.given_actions && rank == 0 && sampled) ag[(size_t)n * p.A + a] = v[u];
                    }
                    put_action(c, row, a, v[u]);
                }
            }
        }
        __syncthreads();
        TIMER_MARK(c, T_ACT)
        if (pifold) {  // a_t = pi(z_t) for the rows < P (tdmpc2.py:156-160), then z_t back into the tile
            if (t == 0) {  // zs <- z_0 in register order (later steps: the SimNorm epilogue's copy)
                f32x16 y[1][2];
                const int hh = c.lane >> 5;
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const f32x4 z = *reinterpret_cast<const f32x4 *>(p.z0 + (size_t)e * WIDTH + 64 * c.wave + 32 * ft + 8 * m + 4 * hh);
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[0][ft][4 * m + r] = z[r];
                    }
                park(c, y, zs);
            }
            cl_layer<0>(c, x, CL_L(p.pi.l[0]), b_pi, 0, ZKB16, 0, gb_of(p.pi.l[1]));
            cl_layer<0>(c, x, CL_L(p.pi.l[1]), p.pi.l[1].bias, 0, ZKB16, 3, gb_of(p.rew.l[0]));
            const float *tape = p.pi_traj_eps ? p.pi_traj_eps + ((size_t)e * p.H + t) * p.P * p.A : nullptr;
            auto eps = [&](int row, int a) -> float {
                if (row >= p.P) return 0.f;
                if (tape) return tape[row * p.A + a];
                return rng_normal(p.seed, p.call, SITE_PITRAJ, t, e, (unsigned)(row * p.A + a));
            };
            cl_head_logits(c, x, p.pi.l[2], true);
            head_pi_rows_s(c, p.A, p.Apad, p.log_std_min, p.log_std_dif, mask, eps,
                           rank == 0 ? p.actions + ((size_t)e * p.H + t) * p.N * p.A : nullptr, p.P, nullptr, nullptr, nullptr, p.P, true);
            tile_from_global_s(c, zs);
            __syncthreads();
        }
        // ---- first layers of dynamics (parked in S0) and reward (S1) over the same [z_t | a_t] tile
        gb_prefetch(c, p.rew.l[1].g, p.rew.l[1].b);
        cl_gemm(c, x, wref(p.dyn.l[0]), 0, wref(p.rew.l[0]), 1, 0, KBA);
        if constexpr (EP) {
            if (term_step) cl_gemm(c, x, wref(p.term.l[0]), 5, WRef{}, 0, 0, ZKB16);  // termination(z_t): same z columns, parked in S5
        }
        cl_epi<0>(c, x, 1, CL_E(p.rew.l[0]), b_rew, GB{});
        // ---- reward: layer 2, two-hot head
        cl_layer<0>(c, x, CL_L(p.rew.l[1]), p.rew.l[1].bias, 0, ZKB16, 2, term_step ? gb_of(p.term.l[0]) : gb_of(p.dyn.l[0]));
        const float r = cl_head_twohot(c, x, p.rew.l[2], p.bins, p.num_bins);
        TIMER_MARK(c, T_HEAD)
        if constexpr (EP) {
            if (term_step) {
                cl_epi<0>(c, x, 5, CL_E(p.term.l[0]), p.term.l[0].bias, gb_of(p.term.l[1]));
                cl_layer<0>(c, x, CL_L(p.term.l[1]), p.term.l[1].bias, 0, ZKB16, 3, gb_of(p.dyn.l[0]));
                termv = fminf(termv + cl_head_term(c, x, p.term.l[2]), 1.f);
            }
        }
        G += disc[t] * (1.f - termv) * r;
        // ---- dynamics: the parked first layer, layers 2 and 3 (SimNorm)
        cl_epi<0>(c, x, 0, CL_E(p.dyn.l[0]), b_dyn, gb_of(p.dyn.l[1]));
        cl_layer<0>(c, x, CL_L(p.dyn.l[1]), p.dyn.l[1].bias, 0, ZKB16, 1, gb_of(p.dyn.l[2]));
        cl_layer<1>(c, x, CL_L(p.dyn.l[2]), p.dyn.l[2].bias, 0, ZKB16, 2, (t == p.H - 1 || pifold) ? gb_of(p.pi.l[0]) : gb_of(p.rew.l[0]),
                    (t == p.H - 1 || pifold) ? zs : nullptr);
    }
    // ---- a_H = pi(z_H) (tdmpc2.py:135); z_H was also saved to zs.  EP: termination(z_H)'s first layer from the same tile first.
    if constexpr (EP) cl_gemm(c, x, wref(p.term.l[0]), 5, WRef{}, 0, 0, ZKB16);
    cl_layer<0>(c, x, CL_L(p.pi.l[0]), b_pi, 0, ZKB16, 0, gb_of(p.pi.l[1]));
    cl_layer<0>(c, x, CL_L(p.pi.l[1]), p.pi.l[1].bias, 0, ZKB16, 1, EP ? gb_of(p.term.l[0]) : gb_of(p.q[q0].l[0]));
    {
        auto eps = [&](int row, int a) -> float {
            const unsigned ridx = (unsigned)((size_t)(row0 + row) * p.A + a);
            if (p.pi_eps) return p.pi_eps[(size_t)e * p.pi_eps_estride + ridx];
            return rng_normal(p.seed, p.call, SITE_PI, p.iter, e, ridx);
        };
        cl_head_logits(c, x, p.pi.l[2], true);
        head_pi_rows_s(c, p.A, p.Apad, p.log_std_min, p.log_std_dif, mask, eps, nullptr, 0, nullptr);
    }
    if constexpr (EP) {  // termination(z_H) (tdmpc2.py:133-134, last loop iteration): the hidden layers overwrite z columns only
        cl_epi<0>(c, x, 5, CL_E(p.term.l[0]), p.term.l[0].bias, gb_of(p.term.l[1]));
        cl_layer<0>(c, x, CL_L(p.term.l[1]), p.term.l[1].bias, 0, ZKB16, 0, gb_of(p.q[q0].l[0]));
        termv = fminf(termv + cl_head_term(c, x, p.term.l[2]), 1.f);
    }
    TIMER_MARK(c, T_HEAD)
    tile_from_global_s(c, zs);
    __syncthreads();
    TIMER_MARK(c, T_ACT)
    // ---- Q(z_H, a_H): the two selected heads (world_model.py:186-216); the second one's first layer is parked in S2
    gb_prefetch(c, p.q[q0].l[1].g, p.q[q0].l[1].b);
    cl_gemm(c, x, wref(p.q[q1].l[0]), 2, wref(p.q[q0].l[0]), 3, 0, KBA);
    cl_epi<0>(c, x, 3, CL_E(p.q[q0].l[0]), b_q0, GB{});
    cl_layer<0>(c, x, CL_L(p.q[q0].l[1]), p.q[q0].l[1].bias, 0, ZKB16, 0, gb_of(p.q[q1].l[0]));
    const float qa = cl_head_twohot(c, x, p.q[q0].l[2], p.bins, p.num_bins);
    TIMER_MARK(c, T_HEAD)
    cl_epi<0>(c, x, 2, CL_E(p.q[q1].l[0]), b_q1, gb_of(p.q[q1].l[1]));
    cl_layer<0>(c, x, CL_L(p.q[q1].l[1]), p.q[q1].l[1].bias, 0, ZKB16, 1, GB{});
    const float qb = cl_head_twohot(c, x, p.q[q1].l[2], p.bins, p.num_bins);
    if (rank != 0) return;  // member 0 owns the values
    TIMER_MARK(c, T_HEAD)
    TIMER_FLUSH(c, p.timing)
    const float val = G + disc[p.H] * (1.f - termv) * ((qa + qb) / 2.f);
    if (!p.fold_refit) {
        if ((tid & 7) == 0 && live) p.value[(size_t)e * p.N + row0 + (tid >> 3)] = val;
        return;
    }
    // ---- elite selection + refit by the last member-0 workgroup of the plan (the hand-over of ks_rollout)
    if ((tid & 7) == 0 && live)
        __hip_atomic_store(p.value + (size_t)e * p.N + row0 + (tid >> 3), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // every wave's value stores must be ACKNOWLEDGED before the ticket moves: a workgroup-scope release fence does not wait
    // for them on gfx950 (the emitted code is `global_store ... sc1; s_barrier; global_atomic_add`), so drain explicitly
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.ticket + e, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == (unsigned)(p.tiles - 1);
        if (last) __hip_atomic_store(p.ticket + e, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_is_last = last;
    }
    __syncthreads();
    if (!s_is_last) return;
    refit_plan(p.rf, e, smem, tid, NTHR);
}
