// The FEW-ROW path of the layer-at-a-time family (round 6): single plans -- the reference's own call pattern, one environment per
// plan() (tdmpc2/evaluate.py:80, trainer/offline_trainer.py:25-31) -- and calls of a few plans of the 48M / 317M world models
// (c3 at E <= 4, c4 at E = 1).  One plan is 512 (1 024) sample rows: a layer is a 512 x 1792 x 1792 (1 024 x 4096 x 4096)
// contraction whose MFMA time on the whole chip is 4.5 (41) us, and a plan is ~105 such layers one behind the other.  What the
// 4-wave tiles of g_gemm_s make of it (profiles/r6a_c3e1_timeline.txt): 32 x 128 tiles, 20 operand bytes from L1 per 1.6 MFMA
// -- the loop runs at the L1's rate, a hidden layer takes 20-25 us alone and 35-40 us beside its partner chain, the narrow
// layers (heads, the SimNorm output layer: 16-96 workgroups walking all of K) 18-30 us.
//
// g_gemm_m: one 512-thread workgroup (8 waves, one per CU) computes a 128-row x 256-column tile over ONE K-PART of the
// contraction and writes raw fp32 partial sums.  Both operands are fragment-packed in HBM (layered_split.cuh); a k32-slab of the
// tile is 4 row tiles x 4 KiB of A and 8 column tiles x 4 KiB of W = 48 planes of 1 KiB, each ONE wavefront-wide LDS-DMA
// (global_load_lds_dwordx4), 6 per wave and slab; the LDS holds a ring of 3 slabs (144 KiB).  A wave owns two row tiles x two
// column tiles: per slab 16 ds_read_b128 (conflict-free at lane * 16) and 24 MFMAs (4 accumulators x 3 products x 2 k16-blocks),
// 8 operand bytes from L1 per MFMA.  Pipeline = g_gemm_w's (layered_wide.cuh): top of phase s: my fragments of slab s are in
// registers (lgkmcnt(0)), my DMA of slab s + 1 has landed (counted vmcnt: one newer slab stays in flight), barrier, request slab
// s + 3 into the slot slab s has just left, 24 MFMAs on slab s with the reads of slab s + 1 between the first eight.  The K-parts are what fills
// the chip when there are few rows: `parts` is chosen per launch so that tiles x parts ~ #CUs (c3 hidden pair: 2 x 28 tiles x 4
// parts; the 317M model's hidden pair: 2 x 128 tiles, whole K).
// No workgroup waits for another one: partial sums leave as 1 KiB-per-instruction write-through stores (the tile is transposed
// through the idle ring), and the consumer is the NEXT launch -- the guide's price list puts a dependent launch boundary at
// 1.1-1.9 us against 4-13 us for an in-kernel split-K seam or a grid barrier (MI355X_MICROARCH.md, "boundary", "splitk-seam",
// "barrier-xcd"), which is also why this family is NOT one persistent kernel per CEM iteration: the fused family is (activations
// stay in LDS); here every layer boundary moves the activations through L2 anyway and a launch is the cheapest seam there is.
// A launch carries up to TWO problems (the reward chain beside the dynamics chain of a step, the two Q heads): one stream, no
// events, half the launches.
//
// m_rows: the consumer -- one wavefront per row sums the K-parts IN PART ORDER (deterministic), applies the layer's output scale and
// bias, and runs what follows the nn.Linear in the reference: LayerNorm + Mish / SimNorm + operand split (NormedLinear,
// tdmpc2/common/layers.py:94-118) -> the next GEMM's packed operand; two_hot_inv (math.py:74-83) -> G / value; the policy head
// (world_model.py:152-174).  The SimNorm rows of a dynamics step also write the NEXT step's action columns of X.
// Sums differ from the whole-K tiles' in the last bits (fp32 association; LayerNorm statistics two-pass instead of Chan-folded):
// the path belongs to TDMPC2_TUNE_KSPLIT != 0 / TDMPC2_TUNE_FEWROW (default on); with either at 0 a plan computes the same bits
// alone and in any batch, as before.
// Included by k_layered.hip after layered_wide.cuh.
#pragma once

constexpr int GM_TM = 128;        // rows of a tile
constexpr int GM_SLOT = 49152;    // bytes of a k32-slab in the ring: [A: 4 row tiles x (kb, plane)][W: 8 column tiles x (kb, plane)] x 1 KiB
constexpr int GM_NS = 3;          // ring slots (144 KiB): two slabs = 96 KiB in flight per CU, two phases (~1.7 us) for a slab to arrive.
                                  // History of this loop (profiles/README.md r6c, r6d, r6v): 64-row tiles with k32-slabs ran at the L2-miss
                                  // latency (80 KiB in flight for 108 GB/s per CU); 128-row tiles with SIX k16-slabs fixed that and ran at
                                  // 0.64 us per 12-MFMA phase against 0.43 MFMA-bound -- one barrier, one pair of counted waits and a pipe
                                  // drain per 12 MFMAs of a wave (g_gemm_w pays the same ~230 cycles per 24); hence k32-slabs again, on 128 rows
constexpr int GM_U = 6;           // phases per unrolled trip: whole turns of the ring (3) and of the register sets (2)
constexpr int GM_LDT = 260;       // floats per row of the transposed output tile in LDS (260 mod 32 = 4: conflict-free 16-byte writes)

struct GemmMProb {
    const _Float16 *A;   // fragment-packed operand buffer, KBa k16-blocks per row; first block contracted a_kb0
    int KBa, a_kb0;
    int nk;              // k32-slabs of the contraction
    const _Float16 *wp;  // split-packed [CT][kbs][2][64][8], + sel * w_sel_stride (halfs); first k16-block kb0
    long w_sel_stride;
    int kb0, kbs;
    int CT, ncolblk, nrowblk, parts;
    const int *sel;      // per-plan ensemble member (Q heads) or null
    long sel_stride;
    int rows_per_env;
    float *ws;           // partial sums: ws[part * part_stride + row * ldw + col], ldw = 256 ncolblk
    long part_stride;
    int ldw;
    int nblk;            // workgroups of this problem: 8 * ceil(ncolblk * parts / 8) * nrowblk
    int rows;            // rows that exist (the rest of the last row block is padding: computed, never stored)
    // epi != 0 (parts == 1 only): the NormedLinear epilogue INSIDE this launch -- g_gemm_w's protocol on 128-row tiles (the column blocks
    // of a row block exchange per-row (mean, M2) partials, canonical combination order: the same bits as every other fused tile) --
    // instead of partial sums + m_rows.  1: Mish, 2: SimNorm.  Every workgroup of a few-row launch is resident (grid <= #CUs, one
    // stream), so the wait is for running peers; it is bounded all the same (WaitClock) and reports like the other fused epilogues.
    int epi;
    const float *oscale; long osc_sel_stride;
    const float *bias; long bias_env_stride, bias_sel_stride;
    const float *ln_g, *ln_b; long gb_sel_stride;
    const float *ascale; long asc_sel_stride;
    int width;
    float *stats;        // [row block][128-column group][128 rows][2]
    unsigned int *arrive;  // [row block] (zero on entry)
    unsigned int *err;
    int fault;
    float *out;          // fragment-packed operand buffer, KBo k16-blocks per row
    int KBo;
};
struct GemmMParams {
    GemmMProb pr[2];
    int nprob;
    int split_xcd;       // two problems of one shape: problem 0 on XCDs 0-3, problem 1 on XCDs 4-7 (see g_gemm_m)
};

struct GmFrags {
    f16x8 ah[2][2], al[2][2], wh[2][2], wl[2][2];  // [k16-block of the slab][row tile / column tile of this wave]
};

// fragment planes of the slab whose read addresses are la / lw (this wave's first row tile / first column tile), two per step.
// LDS image of a row / column tile inside a slot: [kb0: hi, lo][kb1: hi, lo] x 1 KiB.
template <int STEP>
__device__ __forceinline__ void gm_read2(GmFrags &f, unsigned la, unsigned lw) {
    if constexpr (STEP == 0) {
        gw_dsrd<0>(f.wh[0][0], lw);
        gw_dsrd<0>(f.ah[0][0], la);
    } else if constexpr (STEP == 1) {
        gw_dsrd<4096>(f.wh[0][1], lw);
        gw_dsrd<4096>(f.ah[0][1], la);
    } else if constexpr (STEP == 2) {
        gw_dsrd<1024>(f.wl[0][0], lw);
        gw_dsrd<4096 + 1024>(f.wl[0][1], lw);
    } else if constexpr (STEP == 3) {
        gw_dsrd<1024>(f.al[0][0], la);
        gw_dsrd<4096 + 1024>(f.al[0][1], la);
    } else if constexpr (STEP == 4) {
        gw_dsrd<2048>(f.wh[1][0], lw);
        gw_dsrd<2048>(f.ah[1][0], la);
    } else if constexpr (STEP == 5) {
        gw_dsrd<4096 + 2048>(f.wh[1][1], lw);
        gw_dsrd<4096 + 2048>(f.ah[1][1], la);
    } else if constexpr (STEP == 6) {
        gw_dsrd<2048 + 1024>(f.wl[1][0], lw);
        gw_dsrd<4096 + 2048 + 1024>(f.wl[1][1], lw);
    } else {
        gw_dsrd<2048 + 1024>(f.al[1][0], la);
        gw_dsrd<4096 + 2048 + 1024>(f.al[1][1], la);
    }
}
__device__ __forceinline__ void gm_read(GmFrags &f, unsigned la, unsigned lw) {
    gm_read2<0>(f, la, lw);
    gm_read2<1>(f, la, lw);
    gm_read2<2>(f, la, lw);
    gm_read2<3>(f, la, lw);
    gm_read2<4>(f, la, lw);
    gm_read2<5>(f, la, lw);
    gm_read2<6>(f, la, lw);
    gm_read2<7>(f, la, lw);
}

// this wave's 6 DMA requests of one slab: its column tile of W (4 KiB contiguous: two k16-blocks x {hi, lo}) and one k16-block
// (2 KiB: hi, lo) of A (wave w: k16-block w & 1 of row tile w >> 1)
__device__ __forceinline__ void gm_issue(char *slot_w, char *slot_a, const char *&pw, const char *&pa, unsigned voff) {
    gw_glds(pw + voff, slot_w);
    gw_glds(pw + voff + 1024, slot_w + 1024);
    gw_glds(pw + voff + 2048, slot_w + 2048);
    gw_glds(pw + voff + 3072, slot_w + 3072);
    gw_glds(pa + voff, slot_a);
    gw_glds(pa + voff + 1024, slot_a + 1024);
    pw += 4096;
    pa += 4096;
}

// (where in a phase the six requests go out: gm_req_at, tile_order.h)
// request I (0..3: W, 4..5: A) of the same slab on its own (GM_SPREAD_ISSUE: one request between two MFMAs of the phase's second half)
template <int I>
__device__ __forceinline__ void gm_issue1(char *slot_w, char *slot_a, const char *&pw, const char *&pa, unsigned voff) {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I < 4) gw_glds(pw + voff + I * 1024, slot_w + I * 1024);
    else gw_glds(pa + voff + (I - 4) * 1024, slot_a + (I - 4) * 1024);
    if constexpr (I == 5) {
        pw += 4096;
        pa += 4096;
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int N>
__device__ __forceinline__ void gm_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One phase (tile_order.h: gm_* -- the ring's schedule arithmetic).  PH = s mod GM_U: ring slot of slab s = PH % 3, of slab s + 1 =
// (PH + 1) % 3; register set of slab s = PH & 1.  24 MFMAs per wave and barrier.
template <int PH, bool STEADY>
__device__ __forceinline__ void gm_phase(f32x16 (&acc)[2][2], GmFrags (&fr)[2], const unsigned (&la)[GM_NS], const unsigned (&lw)[GM_NS], char *ring_w,
                                         char *ring_a, const char *&pw, const char *&pa, unsigned voff, bool issue, bool next, int vmc) {
    constexpr int SL = PH % GM_NS, SN = (PH + 1) % GM_NS;
    GmFrags &c = fr[PH & 1], &n = fr[(PH + 1) & 1];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // my DMA of slab s + 1 has landed: `vmc` newer requests may stay in flight (steady: one slab = GM_REQ)
    if (STEADY || vmc >= GM_REQ) gm_wait_vm<(GM_NS - 2) * GM_REQ>();
    else gm_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#if GM_SPREAD_ISSUE == 0
    if (STEADY || issue) gm_issue(ring_w + SL * GM_SLOT, ring_a + SL * GM_SLOT, pw, pa, voff);
#endif
    // request gm_req_at(k) goes out behind MFMA k of the phase
#define GM_AT(K)                                                                                                  \
    if constexpr (gm_req_at(K) >= 0) {                                                                            \
        if (STEADY || issue) gm_issue1<(gm_req_at(K) >= 0 ? gm_req_at(K) : 0)>(ring_w + SL * GM_SLOT, ring_a + SL * GM_SLOT, pw, pa, voff); \
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool rd = STEADY || next;
    // per accumulator and k16-block: w_hi a_hi, then w_lo a_hi, then w_hi a_lo -- the order of g_gemm_s / g_gemm_w (acc[column tile][row tile])
    GW_MFMA(acc[0][0], c.wh[0][0], c.ah[0][0]);
    if (rd) gm_read2<0>(n, la[SN], lw[SN]);
    GM_AT(0)
    GW_MFMA(acc[1][0], c.wh[0][1], c.ah[0][0]);
    if (rd) gm_read2<1>(n, la[SN], lw[SN]);
    GM_AT(1)
    GW_MFMA(acc[0][1], c.wh[0][0], c.ah[0][1]);
    if (rd) gm_read2<2>(n, la[SN], lw[SN]);
    GM_AT(2)
    GW_MFMA(acc[1][1], c.wh[0][1], c.ah[0][1]);
    if (rd) gm_read2<3>(n, la[SN], lw[SN]);
    GM_AT(3)
    GW_MFMA(acc[0][0], c.wl[0][0], c.ah[0][0]);
    if (rd) gm_read2<4>(n, la[SN], lw[SN]);
    GM_AT(4)
    GW_MFMA(acc[1][0], c.wl[0][1], c.ah[0][0]);
    if (rd) gm_read2<5>(n, la[SN], lw[SN]);
    GM_AT(5)
    GW_MFMA(acc[0][1], c.wl[0][0], c.ah[0][1]);
    if (rd) gm_read2<6>(n, la[SN], lw[SN]);
    GM_AT(6)
    GW_MFMA(acc[1][1], c.wl[0][1], c.ah[0][1]);
    if (rd) gm_read2<7>(n, la[SN], lw[SN]);
    GM_AT(7)
    GW_MFMA(acc[0][0], c.wh[0][0], c.al[0][0]);
    GM_AT(8)
    GW_MFMA(acc[1][0], c.wh[0][1], c.al[0][0]);
    GM_AT(9)
    GW_MFMA(acc[0][1], c.wh[0][0], c.al[0][1]);
    GM_AT(10)
    GW_MFMA(acc[1][1], c.wh[0][1], c.al[0][1]);
    GM_AT(11)
    GW_MFMA(acc[0][0], c.wh[1][0], c.ah[1][0]);
    GM_AT(12)
    GW_MFMA(acc[1][0], c.wh[1][1], c.ah[1][0]);
    GM_AT(13)
    GW_MFMA(acc[0][1], c.wh[1][0], c.ah[1][1]);
    GM_AT(14)
    GW_MFMA(acc[1][1], c.wh[1][1], c.ah[1][1]);
    GM_AT(15)
    GW_MFMA(acc[0][0], c.wl[1][0], c.ah[1][0]);
    GM_AT(16)
    GW_MFMA(acc[1][0], c.wl[1][1], c.ah[1][0]);
    GM_AT(17)
    GW_MFMA(acc[0][1], c.wl[1][0], c.ah[1][1]);
    GM_AT(18)
    GW_MFMA(acc[1][1], c.wl[1][1], c.ah[1][1]);
    GM_AT(19)
    GW_MFMA(acc[0][0], c.wh[1][0], c.al[1][0]);
    GM_AT(20)
    GW_MFMA(acc[1][0], c.wh[1][1], c.al[1][0]);
    GM_AT(21)
    GW_MFMA(acc[0][1], c.wh[1][0], c.al[1][1]);
    GM_AT(22)
    GW_MFMA(acc[1][1], c.wh[1][1], c.al[1][1]);
#undef GM_AT
}

// The NormedLinear epilogue of a whole-K tile (GemmMProb::epi): g_gemm_w's, on 128 rows -- see layered_wide.cuh / layered_split.cuh
// for the protocol and for why the combination order is what it is.
__device__ __forceinline__ void gm_epilogue_ln(const GemmMProb &p, f32x16 (&acc)[2][2], float *lds, int rb, int cb, int row0, int sel, int wr, int wc,
                                               float vb, float vg, float vbe) {
    constexpr int TM = GM_TM;
    const int tid = threadIdx.x, lane = tid & 63, i32 = lane & 31, hh = lane >> 5;
    const int ct0 = cb * 8 + wc * 2;
    const float osc = p.oscale[(size_t)sel * p.osc_sel_stride];
    __syncthreads();  // every wave is done with the ring
    float *red = lds;                 // [8 column tiles of the block][TM][2]
    float *rs = red + 8 * TM * 2;     // [TM][2]
    float *vecs = rs + TM * 2;        // [bias | g | b][256]
    if (tid < 256) {
        vecs[tid] = vb;
        vecs[256 + tid] = vg;
        vecs[512 + tid] = vbe;
    }
    __syncthreads();
    // (1) v = acc * oscale + bias, in place
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(vecs + (wc * 2 + n) * 32 + 8 * j + 4 * hh);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[n][i][4 * j + r] = fmaf(acc[n][i][4 * j + r], osc, b4[r]);
        }
    // (2) (mean, M2) of every row over each 32-column tile: 16 thread-local values + the lane ^ 32 half
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float s1 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) s1 += acc[n][i][e];
            s1 += __shfl_xor(s1, 32);
            const float mw = s1 * (1.f / 32.f);
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d = acc[n][i][e] - mw;
                q = fmaf(d, d, q);
            }
            q += __shfl_xor(q, 32);
            if (hh == 0) {
                red[((wc * 2 + n) * TM + (2 * wr + i) * 32 + i32) * 2 + 0] = mw;
                red[((wc * 2 + n) * TM + (2 * wr + i) * 32 + i32) * 2 + 1] = q;
            }
        }
    __syncthreads();
    // (3) fold the tiles of each 128-column group left to right -> stats[row block][group][row]: 2 groups x 128 rows
    const int NG = (p.CT + 3) / 4;
    if (tid < 2 * TM) {
        const int g = tid / TM, r = tid - g * TM, G = cb * 2 + g;
        if (G < NG) {
            float n_acc = 0.f, m_acc = 0.f, q_acc = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (G * 4 + t >= p.CT) break;
                chan_fold(n_acc, m_acc, q_acc, 32.f, red[((g * 4 + t) * TM + r) * 2], red[((g * 4 + t) * TM + r) * 2 + 1]);
            }
            float *slot = p.stats + (((size_t)rb * NG + G) * TM + r) * 2;
            __hip_atomic_store(slot, m_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(slot + 1, q_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // (4) arrive (stores acknowledged first), wait for the row block's other column blocks -- bounded
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (!(p.fault && rb == 0 && cb == 0)) __hip_atomic_fetch_add(p.arrive + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        WaitClock wck;
        while (__hip_atomic_load(p.arrive + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.ncolblk) {
            if (wck.expired(p.err)) {
                if (p.err) raise_fault(p.err, 3u | ((unsigned)rb << 4) | ((unsigned)cb << 16) |
                                                  (__hip_atomic_load(p.arrive + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 24));
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    // (5) the row's statistics: the groups folded left to right
    if (tid < TM) {
        float n_acc = 0.f, m_acc = 0.f, q_acc = 0.f;
        const float *all = p.stats + ((size_t)rb * NG * TM + tid) * 2;
        for (int g0 = 0; g0 < NG; g0 += 8) {
            float mb[8], qb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int G = g0 + u < NG ? g0 + u : NG - 1;
                mb[u] = __hip_atomic_load(all + (size_t)G * TM * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                qb[u] = __hip_atomic_load(all + (size_t)G * TM * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (g0 + u >= NG) break;
                int nt4 = p.CT - (g0 + u) * 4;
                nt4 = nt4 > 4 ? 4 : nt4;
                chan_fold(n_acc, m_acc, q_acc, 32.f * (float)nt4, mb[u], qb[u]);
            }
        }
        rs[2 * tid] = m_acc;
        rs[2 * tid + 1] = 1.0f / sqrtf(q_acc / n_acc + LN_EPS);
    }
    __syncthreads();
    // (6) normalise, activate, split -> the fragment-packed output: v_permlane32_swap pairs the two k-halves of a row so that one
    // 16-byte store per lane writes a whole 1 KiB fragment plane (see g_gemm_w)
    const bool mish = p.epi == 1;
    const float oscl = mish ? p.ascale[(size_t)sel * p.asc_sel_stride] : ACT_SCALE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rloc = (2 * wr + i) * 32 + i32;
        const float rmean = rs[2 * rloc], rrstd = rs[2 * rloc + 1];
        char *otile = reinterpret_cast<char *>(p.out) + (size_t)((row0 >> 5) + 2 * wr + i) * p.KBo * 2048 + lane * 16;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (ct0 + n >= p.CT) continue;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                unsigned hw[2][2], lw2[2][2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * jp + jj;
                    const f32x4 g4 = *reinterpret_cast<const f32x4 *>(vecs + 256 + (wc * 2 + n) * 32 + 8 * j + 4 * hh);
                    const f32x4 be4 = *reinterpret_cast<const f32x4 *>(vecs + 512 + (wc * 2 + n) * 32 + 8 * j + 4 * hh);
                    f32x4 y;
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = fmaf((acc[n][i][4 * j + r] - rmean) * rrstd, g4[r], be4[r]);
                    if (mish) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[r] = mish_fast(y[r]);
                    } else {  // SimNorm: groups of 8 consecutive features = this lane's 4 + lane ^ 32's 4
                        float m = fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3]));
                        m = fmaxf(m, __shfl_xor(m, 32));
                        float es = 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            y[r] = __expf(y[r] - m);
                            es += y[r];
                        }
                        es += __shfl_xor(es, 32);
                        const float inv = 1.0f / es;
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[r] *= inv;
                    }
                    f16x4 hi, lo;
                    split4(y, hi, lo, oscl);
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 hb = __builtin_bit_cast(u32x2, hi), lb = __builtin_bit_cast(u32x2, lo);
                    hw[jj][0] = hb[0]; hw[jj][1] = hb[1];
                    lw2[jj][0] = lb[0]; lw2[jj][1] = lb[1];
                }
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 ho, lo4;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const auto sh = __builtin_amdgcn_permlane32_swap(hw[0][d], hw[1][d], false, false);
                    const auto sl = __builtin_amdgcn_permlane32_swap(lw2[0][d], lw2[1][d], false, false);
                    ho[d] = sh[0]; ho[2 + d] = sh[1];
                    lo4[d] = sl[0]; lo4[2 + d] = sl[1];
                }
                char *o = otile + (size_t)((ct0 + n) * 2 + jp) * 2048;
                *reinterpret_cast<u32x4 *>(o) = ho;
                *reinterpret_cast<u32x4 *>(o + 1024) = lo4;
            }
        }
    }
}

__global__ __launch_bounds__(512) void g_gemm_m(GemmMParams P) {
    __shared__ __attribute__((aligned(1024))) char ring[GM_NS * GM_SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    // workgroup -> (problem, row block, column job = column block x K-part).  Block b runs on XCD b % 8.  One problem, or two of
    // different shapes: problem 1's blocks follow problem 0's (every nblk is a multiple of 8) and XCD x runs the jobs x, x + 8, ... of
    // either for ALL row blocks -- a job's W slice comes through ONE L2; with parts | 8 an XCD sees one K-part, i.e. 1 / parts of every
    // A row.  Two problems of ONE shape (the two chains' hidden layers): problem 0 on XCDs 0-3, problem 1 on XCDs 4-7, jobs x, x + 4,
    // ... -- an XCD then streams ONE problem's A rows and four (seven) W slices instead of both problems' A rows and half as many
    // slices of each: a third less through the fabric per MFMA on the 317M model's hidden pair.
    int pi, x, lanes;
    if (P.split_xcd) {
        x = b & 7;
        pi = x >> 2;
        x &= 3;
        lanes = 4;
    } else {
        pi = (P.nprob > 1 && b >= P.pr[0].nblk) ? 1 : 0;
        if (pi) b -= P.pr[0].nblk;
        x = b & 7;
        lanes = 8;
    }
    const GemmMProb &p = P.pr[pi];
    const int t = b >> 3;
    const int jj = t / p.nrowblk, rb = t - jj * p.nrowblk, job = jj * lanes + x;
    if (job >= p.ncolblk * p.parts) return;
    const int cb = job / p.parts, part = job - cb * p.parts;
    const int row0 = rb * GM_TM;
    const int sel = p.sel ? p.sel[(size_t)(row0 / p.rows_per_env) * p.sel_stride] : 0;
    const int s0 = (int)((long)part * p.nk / p.parts), nk = (int)((long)(part + 1) * p.nk / p.parts) - s0;
    const int wr = wave >> 2, wc = wave & 3;  // accumulators: row tiles 2 wr, 2 wr + 1; column tiles 2 wc, 2 wc + 1
    // DMA role: column tile `wave` of W (a tile past the matrix re-reads the last one: its sums land in the padding of ws), and
    // k16-block (wave & 1) of row tile (wave >> 1) of A
    const int ctl = cb * 8 + wave < p.CT ? cb * 8 + wave : p.CT - 1;
    const char *pw = reinterpret_cast<const char *>(p.wp + (size_t)sel * p.w_sel_stride) + ((size_t)ctl * p.kbs + p.kb0 + 2 * s0) * 2048;
    const char *pa = reinterpret_cast<const char *>(p.A) + ((size_t)((row0 >> 5) + (wave >> 1)) * p.KBa + p.a_kb0 + 2 * s0) * 2048 + (wave & 1) * 2048;
    unsigned voff = (unsigned)lane * 16u;
    asm volatile("" : "+v"(voff));
    char *ring_w = ring + 16384 + wave * 4096;
    char *ring_a = ring + wave * 2048;
    const unsigned lbase = lds_addr_of(ring) + (unsigned)lane * 16u;
    unsigned la[GM_NS], lw[GM_NS];
#pragma unroll
    for (int i = 0; i < GM_NS; ++i) {
        la[i] = lbase + (unsigned)(i * GM_SLOT) + (unsigned)wr * 8192u;
        lw[i] = lbase + (unsigned)(i * GM_SLOT) + 16384u + (unsigned)wc * 8192u;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][i][e] = 0.f;
    GmFrags fr[2];
    // fused epilogue: the block's 256 columns of bias / LayerNorm weight / bias ride through the main loop in three registers
    float vb = 0.f, vg = 0.f, vbe = 0.f;
    if (p.epi && tid < 256) {
        const int col = cb * 256 + tid;
        if (col < p.CT * 32) {
            const float *bp = p.bias + (size_t)sel * p.bias_sel_stride;
            if (p.bias_env_stride != 0) bp += (size_t)(row0 / p.rows_per_env) * p.bias_env_stride;
            vb = bp[col];
            vg = p.ln_g[(size_t)sel * p.gb_sel_stride + col];
            vbe = p.ln_b[(size_t)sel * p.gb_sel_stride + col];
        }
    }

    // prologue: slabs 0 .. min(nk, NS) - 1 requested; slab 0 into registers
    const int npro = gm_prologue_slabs(nk, GM_NS);
#pragma unroll
    for (int d = 0; d < GM_NS; ++d)
        if (d < npro) gm_issue(ring_w + d * GM_SLOT, ring_a + d * GM_SLOT, pw, pa, voff);
    __builtin_amdgcn_sched_barrier(0);
    // slab 0 has landed: gm_prologue_vmcnt(npro) = GM_REQ (npro - 1) requests may stay in flight
    if (npro >= 3) gm_wait_vm<2 * GM_REQ>();
    else if (npro == 2) gm_wait_vm<GM_REQ>();
    else gm_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    gm_read(fr[0], la[0], lw[0]);

    int s = 0;
#define GM_STEADY(PH) gm_phase<PH, true>(acc, fr, la, lw, ring_w, ring_a, pw, pa, voff, true, true, GM_REQ);
#define GM_TAIL(PH)                                                                                        \
    if (s + PH < nk) {                                                                                     \
        const GmTailStep ts = gm_tail_step(s + PH, nk, GM_NS);                                             \
        gm_phase<PH, false>(acc, fr, la, lw, ring_w, ring_a, pw, pa, voff, ts.issue, ts.next, ts.vmc);     \
    }
#pragma unroll 1
    for (; gm_steady_trip(s, nk, GM_NS, GM_U); s += GM_U) {
        GM_STEADY(0) GM_STEADY(1) GM_STEADY(2) GM_STEADY(3) GM_STEADY(4) GM_STEADY(5)
    }
#pragma unroll 1
    for (; s < nk; s += GM_U) {
        GM_TAIL(0) GM_TAIL(1) GM_TAIL(2) GM_TAIL(3) GM_TAIL(4) GM_TAIL(5)
    }
#undef GM_TAIL
#undef GM_STEADY
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // XDL write -> VALU read of the accumulators

    if (p.epi) {
        gm_epilogue_ln(p, acc, reinterpret_cast<float *>(ring), rb, cb, row0, sel, wr, wc, vb, vg, vbe);
        return;
    }
    // ---- the tile leaves as fp32 rows: transposed through the (idle) ring so that one store instruction of a wave covers 1 KiB
    // of one row.  C = [feature][row] (the weight fragment is the MFMA's A operand, as in the NormedLinear tiles): a lane holds
    // row (lane & 31) of a row tile and features 8 j + 4 (lane >> 5) + (0..3) of a column tile.
    __syncthreads();  // every wave is done with the ring (nothing in flight: the last phase waited for vmcnt(0))
    float *tile = reinterpret_cast<float *>(ring);
    {
        const int i32 = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float *trow = tile + ((2 * wr + i) * 32 + i32) * GM_LDT + wc * 64 + 4 * hh;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[n][i][4 * j + r];
                    *reinterpret_cast<f32x4 *>(trow + n * 32 + 8 * j) = v;
                }
        }
    }
    __syncthreads();
    float *out = p.ws + (size_t)part * p.part_stride + (size_t)row0 * p.ldw + cb * 256 + lane * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wave * 16 + r;
        if (row0 + row >= p.rows) break;  // wave-uniform: padding rows of the last row block (the policy-prior rows: 24-32 of 128) -- m_rows never reads them
        const f32x4 v = *reinterpret_cast<const f32x4 *>(tile + row * GM_LDT + lane * 4);
        gw_st_sc1(out + (size_t)row * p.ldw, v);  // (write-through: plain stores, left to the end-of-kernel write-back, cost the 48M plan 2.5-3 %, r6zzb)
    }
}

// ---------------------------------------------------------------- m_rows: what follows the nn.Linear, one wavefront per row
enum { MR_LN_MISH = 0, MR_LN_SIMNORM = 1, MR_TWOHOT = 2, MR_PI = 3, MR_TERM = 4 };
struct MRowProb {
    int kind, rows, rows_per_env, nwg;  // workgroups: ceil(rows / 8) (eight rows each: MR_LN_* as 8 x 64 column groups, heads one wavefront per row)
    const float *ws;                    // the GEMM's partial sums (GemmMProb::ws)
    long part_stride;
    int ldw, parts;
    const float *oscale;                // the layer's output scale(s), + sel * osc_sel_stride
    long osc_sel_stride;
    const float *bias;                  // bias(row) = bias + env(row) * bias_env_stride + sel * bias_sel_stride
    long bias_env_stride, bias_sel_stride;
    const int *sel;
    long sel_stride;
    const int *row_env;
    // MR_LN_*: LayerNorm over `width` columns, Mish / SimNorm, operand split -> columns [0, width) of the packed buffer `out`
    int width;
    const float *g, *b;
    long gb_sel_stride;
    const float *ascale;
    long asc_sel_stride;
    char *out;
    int KBo;
    // MR_LN_SIMNORM into X, optional: the NEXT step's action columns [act_L, act_ldx) of the row <- split(actions[e, act_t, n, :])
    const float *actions;
    int act_t, act_A, act_L, act_ldx, act_N, act_H, act_nsub, act_noff;
    float *term;                        // MR_TERM: [rows] (tdmpc2.py:133-134)
    TwoHotParams th;                    // MR_TWOHOT (lg / ld unused)
    PiHeadParams pi;                    // MR_PI (lg / ld unused)
};
struct MRowParams {
    MRowProb pr[2];
    int nprob;
    int serial;  // two HEAD problems over the same rows, the second reading what the first wrote (Q heads: qtmp): each wavefront does both, in order
};

// A workgroup of 512 threads owns MR_R = 8 consecutive rows.
// LN kinds: thread t holds 8 consecutive columns -- one k-half of the packed operand -- of row (t & 7): column group (t >> 3) + 64 q,
// q < NQ.  Every load of the eight rows (the K-parts in chunks of up to 32 sixteen-byte loads per thread) is in flight before the
// first use -- the kernel is a chain of memory round trips: the partial sums come from the fabric (the GEMM stored them write-through).
// The output leaves as whole 128-byte lines: the 16 bytes of a row's k-half sit next to those of the seven neighbouring rows in the
// fragment-packed layout, i.e. lanes 8 i .. 8 i + 7 of a store cover one line.  (Measured on the way, r6c .. r6g: one wavefront or
// one 256-thread workgroup per row, 4 columns per thread: every output line assembled from eight 16-byte writes of eight different
// workgroups on whatever XCDs they ran on -- 14 us per launch on 2 x 512 rows of the 48M model, 35 us on 2 x 1 024 rows of the 317M
// model, 1.4 TB/s; the same with four rows per workgroup and a quarter of the parameter loads: 15 / 32 us.)
// SimNorm's groups of 8 columns are thread-local.  Head kinds: one wavefront per row, all K-parts of its <= 128 columns in flight.
constexpr int MR_R = 8, MR_THREADS = 512;
template <int NQ>
__device__ __forceinline__ void m_rows_ln(const MRowProb &p, int wg, float (*red)[MR_R][MR_THREADS / 64]) {
    constexpr int PCH = NQ <= 4 ? 4 : 2;  // K-parts loaded together
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int r = tid & 7, cg0 = tid >> 3;
    const int row = wg * MR_R + r;
    const bool rok = row < p.rows;
    const int ncg = p.width / 8;  // column groups of the row
    const float *wr = p.ws + (size_t)row * p.ldw;
    f32x4 v[NQ][2];
    {   // the first chunk of parts
        f32x4 u[PCH][NQ][2];
#pragma unroll
        for (int k = 0; k < PCH; ++k)
            if (k < p.parts) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int cg = cg0 + 64 * q;
                    const bool ok = rok && cg < ncg;
                    const float *src = wr + (size_t)k * p.part_stride + 8 * cg;
                    u[k][q][0] = ok ? *reinterpret_cast<const f32x4 *>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
                    u[k][q][1] = ok ? *reinterpret_cast<const f32x4 *>(src + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            v[q][0] = u[0][q][0];
            v[q][1] = u[0][q][1];
        }
#pragma unroll
        for (int k = 1; k < PCH; ++k)
            if (k < p.parts) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[q][0][e] = __fadd_rn(v[q][0][e], u[k][q][0][e]);
                        v[q][1][e] = __fadd_rn(v[q][1][e], u[k][q][1][e]);
                    }
            }
    }
    for (int p0 = PCH; p0 < p.parts; p0 += PCH) {  // further parts, in order
        f32x4 u[PCH][NQ][2];
#pragma unroll
        for (int k = 0; k < PCH; ++k)
            if (p0 + k < p.parts) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int cg = cg0 + 64 * q;
                    const bool ok = rok && cg < ncg;
                    const float *src = wr + (size_t)(p0 + k) * p.part_stride + 8 * cg;
                    u[k][q][0] = ok ? *reinterpret_cast<const f32x4 *>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
                    u[k][q][1] = ok ? *reinterpret_cast<const f32x4 *>(src + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
        for (int k = 0; k < PCH; ++k)
            if (p0 + k < p.parts) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[q][0][e] = __fadd_rn(v[q][0][e], u[k][q][0][e]);
                        v[q][1][e] = __fadd_rn(v[q][1][e], u[k][q][1][e]);
                    }
            }
    }
    // the layer's vectors: one environment / ensemble member per workgroup (rows_per_env % MR_R == 0)
    const int env = (wg * MR_R) / p.rows_per_env;
    const int sel = p.sel ? p.sel[(size_t)env * p.sel_stride] : 0;
    const float *bp = p.bias + (size_t)sel * p.bias_sel_stride;
    if (p.bias_env_stride != 0) bp += (size_t)env * p.bias_env_stride;
    const float *g = p.g + (size_t)sel * p.gb_sel_stride, *bb = p.b + (size_t)sel * p.gb_sel_stride;
    const bool mish = p.kind == MR_LN_MISH;
    const float osc = p.oscale[(size_t)sel * p.osc_sel_stride];
    const float oscl = mish ? p.ascale[(size_t)sel * p.asc_sel_stride] : ACT_SCALE;
    float s1 = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int cg = cg0 + 64 * q;
        if (cg < ncg) {
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bp + 8 * cg), b1 = *reinterpret_cast<const f32x4 *>(bp + 8 * cg + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[q][0][e] = fmaf(v[q][0][e], osc, b0[e]);
                v[q][1][e] = fmaf(v[q][1][e], osc, b1[e]);
            }
            s1 += ((v[q][0][0] + v[q][0][1]) + (v[q][0][2] + v[q][0][3])) + ((v[q][1][0] + v[q][1][1]) + (v[q][1][2] + v[q][1][3]));
        }
    }
    // row r's threads of this wavefront are the lanes r, r + 8, ..., r + 56
    s1 += __shfl_xor(s1, 8);
    s1 += __shfl_xor(s1, 16);
    s1 += __shfl_xor(s1, 32);
    if (lane < 8) red[0][lane][wv] = s1;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int w = 0; w < MR_THREADS / 64; ++w) mean += red[0][r][w];
    mean /= (float)p.width;
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if (cg0 + 64 * q < ncg) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[q][h][e] - mean;
                    ss = fmaf(d, d, ss);
                }
        }
    ss += __shfl_xor(ss, 8);
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    if (lane < 8) red[1][lane][wv] = ss;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w = 0; w < MR_THREADS / 64; ++w) var += red[1][r][w];
    const float rstd = 1.0f / sqrtf(var / (float)p.width + LN_EPS);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int cg = cg0 + 64 * q;
        if (cg >= ncg) continue;
        float y[8];
        {
            const f32x4 g0 = *reinterpret_cast<const f32x4 *>(g + 8 * cg), g1 = *reinterpret_cast<const f32x4 *>(g + 8 * cg + 4);
            const f32x4 e0 = *reinterpret_cast<const f32x4 *>(bb + 8 * cg), e1 = *reinterpret_cast<const f32x4 *>(bb + 8 * cg + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = fmaf((v[q][0][e] - mean) * rstd, g0[e], e0[e]);
                y[4 + e] = fmaf((v[q][1][e] - mean) * rstd, g1[e], e1[e]);
            }
        }
        if (mish) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = mish_fast(y[e]);
        } else {  // SimNorm over this thread's 8 columns
            float m = y[0];
#pragma unroll
            for (int e = 1; e < 8; ++e) m = fmaxf(m, y[e]);
            float es = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                y[e] = __expf(y[e] - m);
                es += y[e];
            }
            const float inv = 1.0f / es;
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] *= inv;
        }
        if (rok) {
            f16x4 h0, l0, h1, l1;
            split4(f32x4{y[0], y[1], y[2], y[3]}, h0, l0, oscl);
            split4(f32x4{y[4], y[5], y[6], y[7]}, h1, l1, oscl);
            f16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi[e] = h0[e]; hi[4 + e] = h1[e];
                lo[e] = l0[e]; lo[4 + e] = l1[e];
            }
            char *o = p.out + opnd_off((size_t)row, 8 * cg, p.KBo);
            *reinterpret_cast<f16x8 *>(o) = hi;
            *reinterpret_cast<f16x8 *>(o + 1024) = lo;
        }
    }
    if (p.actions) {  // the next step's [z | a] rows are complete when this kernel ends: thread (r, cg0) fills row r's action columns
        if (rok) {
            const size_t e = (size_t)row / p.act_nsub, n = p.act_noff + (size_t)row % p.act_nsub;
            for (int c = p.act_L + cg0; c < p.act_ldx; c += MR_THREADS / 8) {
                const int a = c - p.act_L;
                put_split(p.out, p.KBo, (size_t)row, c, a < p.act_A ? p.actions[((e * p.act_H + p.act_t) * p.act_N + n) * p.act_A + a] : 0.f);
            }
        }
    }
}

// LN kinds on FEW rows (fewer than MR_WIDE_MIN: one plan of the 48M model, the policy-prior pass): one workgroup of T = 256 threads per
// row, thread t holds columns 4 (t + T q) .. + 3 -- more workgroups in flight than eight rows per workgroup would give (128 workgroups for a
// 48M plan's two chains: 3.84 ms per plan against 3.29, r6h); the 8-byte stores of this mapping assemble every output line from
// eight rows' writes, which is what the eight-row mapping above avoids where there are rows enough.
#ifndef MR_WIDE_MIN_ROWS
#define MR_WIDE_MIN_ROWS 1024
#endif
constexpr int MR_WIDE_MIN = MR_WIDE_MIN_ROWS;
template <int QN, int T>
__device__ __forceinline__ void m_rows_ln1(const MRowProb &p, int row, float (*red)[MR_R][MR_THREADS / 64]) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int n4 = p.width / 4;
    const float *wr = p.ws + (size_t)row * p.ldw;
    f32x4 v[QN], u[3][QN];
    const int np0 = p.parts < 4 ? p.parts : 4;
#pragma unroll
    for (int q = 0; q < QN; ++q)
        if (tid + T * q < n4) {
            v[q] = *reinterpret_cast<const f32x4 *>(wr + 4 * (tid + T * q));
#pragma unroll
            for (int pt = 1; pt < 4; ++pt)
                if (pt < np0) u[pt - 1][q] = *reinterpret_cast<const f32x4 *>(wr + (size_t)pt * p.part_stride + 4 * (tid + T * q));
        }
    const int env = row / p.rows_per_env;
    const int sel = p.sel ? p.sel[(size_t)env * p.sel_stride] : 0;
    const float *bp = p.bias + (size_t)sel * p.bias_sel_stride;
    if (p.bias_env_stride != 0) bp += (size_t)env * p.bias_env_stride;
    const float *g = p.g + (size_t)sel * p.gb_sel_stride, *bb = p.b + (size_t)sel * p.gb_sel_stride;
    const bool mish = p.kind == MR_LN_MISH;
    f32x4 b4[QN], gg[QN], be[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q)
        if (tid + T * q < n4) {
            b4[q] = *reinterpret_cast<const f32x4 *>(bp + 4 * (tid + T * q));
            gg[q] = *reinterpret_cast<const f32x4 *>(g + 4 * (tid + T * q));
            be[q] = *reinterpret_cast<const f32x4 *>(bb + 4 * (tid + T * q));
        }
    const float osc = p.oscale[(size_t)sel * p.osc_sel_stride];
    const float oscl = mish ? p.ascale[(size_t)sel * p.asc_sel_stride] : ACT_SCALE;
#pragma unroll
    for (int q = 0; q < QN; ++q)
        if (tid + T * q < n4) {
#pragma unroll
            for (int pt = 1; pt < 4; ++pt)
                if (pt < np0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[q][e] = __fadd_rn(v[q][e], u[pt - 1][q][e]);
                }
        }
    for (int p0 = 4; p0 < p.parts; p0 += 4) {  // more than four parts: four at a time, in order
        f32x4 w[4][QN];
#pragma unroll
        for (int q = 0; q < QN; ++q)
            if (tid + T * q < n4) {
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
                    if (p0 + pt < p.parts) w[pt][q] = *reinterpret_cast<const f32x4 *>(wr + (size_t)(p0 + pt) * p.part_stride + 4 * (tid + T * q));
            }
#pragma unroll
        for (int q = 0; q < QN; ++q)
            if (tid + T * q < n4) {
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
                    if (p0 + pt < p.parts) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[q][e] = __fadd_rn(v[q][e], w[pt][q][e]);
                    }
            }
    }
    float s1 = 0.f;
#pragma unroll
    for (int q = 0; q < QN; ++q)
        if (tid + T * q < n4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[q][e] = fmaf(v[q][e], osc, b4[q][e]);
            s1 += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
        }
    s1 = group_sum<64>(s1);
    if (lane == 0) red[0][0][wv] = s1;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) mean += red[0][0][w];
    mean /= (float)p.width;
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < QN; ++q)
        if (tid + T * q < n4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[q][e] - mean;
                ss = fmaf(d, d, ss);
            }
        }
    ss = group_sum<64>(ss);
    if (lane == 0) red[1][0][wv] = ss;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) var += red[1][0][w];
    const float rstd = 1.0f / sqrtf(var / (float)p.width + LN_EPS);
#pragma unroll
    for (int q = 0; q < QN; ++q) {
        const int c4 = tid + T * q;
        const bool ok = c4 < n4;
        if (64 * wv + T * q >= n4) continue;  // this wavefront is past the row (uniform)
        f32x4 y;
        if (ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaf((v[q][e] - mean) * rstd, gg[q][e], be[q][e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = -INFINITY;
        }
        if (mish) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = mish_fast(y[e]);
        } else {  // SimNorm: groups of 8 columns = this lane's 4 and lane ^ 1's
            float m = fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3]));
            m = fmaxf(m, __shfl_xor(m, 1));
            float es = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = ok ? __expf(y[e] - m) : 0.f;
                es += y[e];
            }
            es += __shfl_xor(es, 1);
            const float inv = 1.0f / es;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] *= inv;
        }
        if (ok) {
            f16x4 hi, lo;
            split4(y, hi, lo, oscl);
            char *o = p.out + opnd_off((size_t)row, 4 * c4, p.KBo);
            *reinterpret_cast<f16x4 *>(o) = hi;
            *reinterpret_cast<f16x4 *>(o + 1024) = lo;
        }
    }
    if (p.actions) {
        const size_t e = (size_t)row / p.act_nsub, n = p.act_noff + (size_t)row % p.act_nsub;
        for (int c = p.act_L + tid; c < p.act_ldx; c += T) {
            const int a = c - p.act_L;
            put_split(p.out, p.KBo, (size_t)row, c, a < p.act_A ? p.actions[((e * p.act_H + p.act_t) * p.act_N + n) * p.act_A + a] : 0.f);
        }
    }
}

// ---- narrow heads: at most 128 columns, one wavefront per row, lane holds columns lane and lane + 64
template <int T>
__device__ __forceinline__ void m_rows_head(const MRowProb &p, int wg, float (*lgs)[128]) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int row = wg * (T / 64) + wv;
    if (row >= p.rows) return;
    const float *wr = p.ws + (size_t)row * p.ldw;
    float xa[16], xb[16];
#pragma unroll
    for (int pt = 0; pt < 16; ++pt)
        if (pt < p.parts) {
            xa[pt] = wr[(size_t)pt * p.part_stride + lane];
            xb[pt] = wr[(size_t)pt * p.part_stride + lane + 64];
        }
    const int env = row / p.rows_per_env;
    const int sel = p.sel ? p.sel[(size_t)env * p.sel_stride] : 0;
    const float osc = p.oscale[(size_t)sel * p.osc_sel_stride];
    const float *bp = p.bias + (size_t)sel * p.bias_sel_stride;
    if (p.bias_env_stride != 0) bp += (size_t)(p.row_env ? p.row_env[row] : env) * p.bias_env_stride;
    float x0 = xa[0], x1 = xb[0];
#pragma unroll
    for (int pt = 1; pt < 16; ++pt)
        if (pt < p.parts) {
            x0 = __fadd_rn(x0, xa[pt]);
            x1 = __fadd_rn(x1, xb[pt]);
        }
    for (int pt = 16; pt < p.parts; ++pt) {
        x0 = __fadd_rn(x0, wr[(size_t)pt * p.part_stride + lane]);
        x1 = __fadd_rn(x1, wr[(size_t)pt * p.part_stride + lane + 64]);
    }
    if (p.kind == MR_TWOHOT) {
        const int nb = p.th.num_bins > 0 ? p.th.num_bins : 1;
        lgs[wv][lane] = lane < nb ? fmaf(x0, osc, bp[lane]) : 0.f;
        lgs[wv][lane + 64] = lane + 64 < nb ? fmaf(x1, osc, bp[lane + 64]) : 0.f;
        __builtin_amdgcn_wave_barrier();
        const float r = twohot_wave(lgs[wv], p.th.bins, p.th.num_bins, lane);
        if (lane == 0) twohot_apply(p.th, row, r);
        return;
    }
    if (p.kind == MR_TERM) {  // term <- clip(term + (sigmoid(logit) > 0.5), max = 1)
        if (lane == 0) {
            const float pr = 1.f / (1.f + expf(-fmaf(x0, osc, bp[0])));
            p.term[row] = fminf(p.term[row] + (pr > 0.5f ? 1.f : 0.f), 1.f);
        }
        return;
    }
    // MR_PI (world_model.py:152-174): lane a < A holds mu (column a); log_std (column A + a) comes from lane A + a (2 A <= 64) or x1
    const PiHeadParams &q = p.pi;
    const int A = q.A;
    const float v0 = lane < 2 * A && lane < 64 ? fmaf(x0, osc, bp[lane]) : 0.f;
    const float v1 = lane + 64 < 2 * A ? fmaf(x1, osc, bp[lane + 64]) : 0.f;
    lgs[wv][lane] = v0;
    lgs[wv][lane + 64] = v1;
    __builtin_amdgcn_wave_barrier();
    if (lane < A) {
        const int a = lane, n = row % q.rows_per_env;
        float mu = lgs[wv][a];
        float ls = q.lsmin + 0.5f * q.lsdif * (tanhf(lgs[wv][A + a]) + 1.f);
        float eps = 0.f;
        if (n < q.nvalid) {
            const unsigned ridx = (unsigned)((size_t)(n + q.n_off) * A + a);
            eps = q.eps ? q.eps[(size_t)env * q.eps_estride + ridx] : rng_normal(q.seed, q.call, q.site, q.iter, env, ridx);
        }
        if (q.mask) {
            const float mk = q.mask[(size_t)(q.row_env ? q.row_env[row] : env) * A + a];
            mu *= mk;
            ls *= mk;
            eps *= mk;
        }
        const float act = tanhf(mu + eps * expf(ls));
        put_split(reinterpret_cast<char *>(q.X), q.ldx / 16, (size_t)row, q.L + a, act);
        if (q.actions && n < q.nvalid) q.actions[(((size_t)env * q.H + q.t) * q.N + n) * A + a] = act;
        if (q.trace) q.trace[(size_t)row * (q.H + 2 + A) + q.H + 2 + a] = act;
    }
}

// T = 512: calls with MR_WIDE_MIN rows or more (eight rows per workgroup); T = 256: fewer rows -- one row per workgroup (LN kinds) or
// four (heads): a single 48M plan's launches are 1 024 small workgroups, and 512-thread workgroups cost them 4 us per launch (r6i)
template <int T>
__global__ __launch_bounds__(T) void m_rows(MRowParams P) {
    __shared__ float lgs[T / 64][128];
    __shared__ float red[2][MR_R][MR_THREADS / 64];
    int wg = blockIdx.x;
    const int pidx = (P.nprob > 1 && wg >= P.pr[0].nwg) ? 1 : 0;
    if (pidx) wg -= P.pr[0].nwg;
    const MRowProb &p = P.pr[pidx];
    if (p.kind == MR_LN_MISH || p.kind == MR_LN_SIMNORM) {
        if constexpr (T == 256) {
            if (p.width <= 1024) m_rows_ln1<1, T>(p, wg, red);
            else if (p.width <= 2048) m_rows_ln1<2, T>(p, wg, red);
            else m_rows_ln1<4, T>(p, wg, red);
        } else {
            if (p.width <= 2048) m_rows_ln<4>(p, wg, red);
            else m_rows_ln<8>(p, wg, red);
        }
        return;
    }
    if (P.serial) {  // (pidx == 0: the grid covers problem 0's rows)
        m_rows_head<T>(P.pr[0], wg, lgs);
        __builtin_amdgcn_wave_barrier();
        m_rows_head<T>(P.pr[1], wg, lgs);
        return;
    }
    m_rows_head<T>(p, wg, lgs);
}
