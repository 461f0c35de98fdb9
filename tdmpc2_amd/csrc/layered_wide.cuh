// g_gemm_w: the throughput GEMM of the layer-at-a-time family (hidden NormedLinear layers of the 48M / 317M world models when a
// call fills the chip: c3 / c4 / c5; reference: NormedLinear, tdmpc2/common/layers.py:94-118, inside the mlp() of
// layers.py:121-133 that WorldModel.next / reward / pi / Q run, world_model.py:114-216).
//
// One 512-thread workgroup (8 waves, one per CU: 128 KiB of LDS) computes a 256-row x 256-column output tile of
//     out = ACT(LayerNorm(A W^T * oscale + bias))            (f16x2-split products, fp32 accumulate: fused_kernels.cuh)
// Both operands are fragment-packed in HBM (layered_split.cuh): a k16-slab of the tile is 8 row tiles x 2 KiB of A and
// 8 column tiles x 2 KiB of W -- 32 planes of 1 KiB, each ONE wavefront-wide LDS-DMA (global_load_lds_dwordx4: 1 KiB of
// contiguous HBM bytes -> 1 KiB of contiguous LDS, no VGPRs, no address arithmetic), 4 per wave and slab.  The LDS holds a ring
// of NS = 4 slabs (32 KiB each).  Per slab a wave reads its 12 fragment planes (4 row tiles x {hi, lo} of A, 2 column tiles x
// {hi, lo} of W; ds_read_b128 at lane * 16: conflict-free by construction) and issues 24 MFMAs (8 accumulator tiles x 3 products):
// 21 operand bytes per MFMA against 32 for the 128 x 256 tile of g_gemm_s, and nothing but the slab's 4 DMA requests goes through
// the vector memory path.
//
// Pipeline (phase s = slab s; fragments double-buffered in registers, 2 x 48 VGPRs):
//     top of phase s      s_waitcnt lgkmcnt(0)        my fragments of slab s (read during phase s - 1) are in registers
//                         s_waitcnt vmcnt(4 (NS - 2)) my DMA requests of slab s + 1 have landed (two newer slabs stay in flight)
//                         s_barrier                   ... and everybody else's; everybody has slab s in registers
//                         4 x LDS-DMA                 slab s + NS -> the ring slot of slab s (free: see the barrier)
//                         24 MFMAs on slab s, the 12 ds_read_b128 of slab s + 1 issued between the first of them
// One barrier per 24 MFMAs; a slab has NS - 1 = 3 phases (2-3 us) to arrive; no instruction of the loop waits for anything
// issued in the same phase.  Every LDS read, MFMA and wait is an `asm volatile` (source order = issue order; hipcc neither
// counts nor moves them), the DMA requests are compiler builtins fenced by sched_barrier.
// The accumulation order per output element is that of g_gemm_s (slab by slab: w_hi a_hi, w_lo a_hi, w_hi a_lo), so the sums --
// and with the canonical LayerNorm combination order the whole layer -- are bit-identical to the other tiles': a plan computes
// the same bits alone (small tiles) and inside a batch that takes this one.
// Epilogue: g_gemm_s's NormedLinear epilogue (statistics exchange between the column blocks of a row block, bounded wait),
// writing the fragment-packed output with one 16-byte store per lane: v_permlane32_swap pairs the two k-halves of a row so
// that a wavefront's store covers one whole 1 KiB fragment plane.
// Included by k_layered.hip after layered_split.cuh.
#pragma once

template <int OFF>
__device__ __forceinline__ void gw_dsrd(f16x8 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void gw_glds(const char *g, char *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}
#define KS_ST "sc1"
#define KS_LD "sc1"
#define GW_MFMA(ACC, WF, AF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(WF), "v"(AF))
// partial accumulators of a K-split tile: write-through (sc1) 16-byte stores, L1-bypassing (sc1) 16-byte loads -- the guide's
// "16-B sc1 stores + drained vmcnt + flag, sc1 loads on the reader" hand-off (MI355X_MICROARCH.md, valid forms; publish-large)
// s_nop 1 BEHIND THE STORE, IN THE SAME STATEMENT: a VMEM store of more than 64 bits reads its data registers after it has
// issued, and a VALU write of one of them needs 2 wait states behind it on gfx940+.  The compiler's hazard recognizer inserts
// them for its own stores but does not look inside inline asm: it reused the just-stored accumulator registers for the next
// store's address one s_mov later, and -- depending on how the CU's waves interleaved -- lanes 12..15 of a quarter wave stored
// address bits (r5z: 1 plan in 2 of the 317M model wrong with two chains in flight, never with one).
__device__ __forceinline__ void gw_st_sc1(float *p, const f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off " KS_ST "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// In-kernel phase clocks of wave 0 (profiling builds only: -DGW_TIMING; TDMPC2_GW_TIMING=1 makes the host allocate and print
// them): cycles from kernel start to [1] end of the main loop, [2] statistics stored, [3] peers arrived, [4] row statistics,
// [5] end; [6] counts workgroups.
#ifdef GW_TIMING
#define GW_T0 unsigned long long gw_t[6]; gw_t[0] = __builtin_amdgcn_s_memtime();
#define GW_T(i) { __builtin_amdgcn_sched_barrier(0); gw_t[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define GW_TFLUSH if (p.timing && tid == 0) { for (int i_ = 1; i_ < 6; ++i_) atomicAdd(p.timing + i_, gw_t[i_] - gw_t[0]); atomicAdd(p.timing + 6, 1ull); }
#else
#define GW_T0
#define GW_T(i)
#define GW_TFLUSH
#endif

struct GwFrags {
    f16x8 ah[4], al[4], wh[2], wl[2];
};

constexpr int GW_SLOT = 32768;  // bytes of one k16-slab in the ring: [A: 8 row tiles x (hi, lo)][W: 8 column tiles x (hi, lo)]
constexpr int GW_NSLOT = 4;     // ring slots: 128 KiB (a fifth -- all 160 KiB of the CU, one more slab in flight -- bought nothing: profiles/README.md r4c)
constexpr int GW_U = GW_NSLOT % 2 ? 2 * GW_NSLOT : GW_NSLOT;  // phases per unrolled trip: whole turns of the ring and of the register sets
constexpr int GW_NB = (GW_NSLOT + 1) / 2;                      // read-address registers per operand (16-bit ds_read offsets: two slots each)

// Fragment planes of the slab in ring slot SLOT (la / lw: this wave's A / W read addresses for slots 0-1, + 65536 for 2-3),
// two per step so that the phase can issue them between its first MFMAs.
template <int SLOT, int STEP>
__device__ __forceinline__ void gw_read2(GwFrags &f, const unsigned (&la)[GW_NB], const unsigned (&lw)[GW_NB]) {
    constexpr int O = (SLOT & 1) * GW_SLOT;
    const unsigned a = la[SLOT >> 1], w = lw[SLOT >> 1];
    if constexpr (STEP == 0) {
        gw_dsrd<O + 0>(f.wh[0], w);
        gw_dsrd<O + 0 * 2048>(f.ah[0], a);
    } else if constexpr (STEP == 1) {
        gw_dsrd<O + 2048>(f.wh[1], w);
        gw_dsrd<O + 1 * 2048>(f.ah[1], a);
    } else if constexpr (STEP == 2) {
        gw_dsrd<O + 2 * 2048>(f.ah[2], a);
        gw_dsrd<O + 3 * 2048>(f.ah[3], a);
    } else if constexpr (STEP == 3) {
        gw_dsrd<O + 1024>(f.wl[0], w);
        gw_dsrd<O + 2048 + 1024>(f.wl[1], w);
    } else if constexpr (STEP == 4) {
        gw_dsrd<O + 0 * 2048 + 1024>(f.al[0], a);
        gw_dsrd<O + 1 * 2048 + 1024>(f.al[1], a);
    } else {
        gw_dsrd<O + 2 * 2048 + 1024>(f.al[2], a);
        gw_dsrd<O + 3 * 2048 + 1024>(f.al[3], a);
    }
}
template <int SLOT>
__device__ __forceinline__ void gw_read(GwFrags &f, const unsigned (&la)[GW_NB], const unsigned (&lw)[GW_NB]) {
    gw_read2<SLOT, 0>(f, la, lw);
    gw_read2<SLOT, 1>(f, la, lw);
    gw_read2<SLOT, 2>(f, la, lw);
    gw_read2<SLOT, 3>(f, la, lw);
    gw_read2<SLOT, 4>(f, la, lw);
    gw_read2<SLOT, 5>(f, la, lw);
}

// One phase.  PH = s mod GW_U (ring slot of slab s = PH % NS, of slab s + 1 = (PH + 1) % NS; register set of slab s = PH & 1).
// STEADY: slab s + NS exists (its DMA is issued here) and so does slab s + 1; otherwise the flags say (vmc: requests that may
// stay in flight at the top of the phase).
// (where in a phase the four requests go out: gw_req_at, tile_order.h)
template <int PH, bool STEADY>
__device__ __forceinline__ void gw_phase(f32x16 (&acc)[2][4], GwFrags (&fr)[2], const unsigned (&la)[GW_NB], const unsigned (&lw)[GW_NB],
                                         char *ring_w, const char *&pa, const char *&pw, unsigned voff, bool issue, bool next, int vmc) {
    constexpr int NS = GW_NSLOT, SL = PH % NS, SN = (PH + 1) % NS;
    GwFrags &c = fr[PH & 1], &n = fr[(PH + 1) & 1];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int wv = gw_phase_vmcnt(STEADY, vmc, NS);  // tile_order.h: the ring's schedule arithmetic (tests/test_ring_schedule.py)
    if (wv == gw_steady_vmcnt(NS)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NS - 2)) : "memory");
    else if (wv == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (wv == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#if GW_SPREAD_ISSUE == 0
    if (STEADY || issue) {  // slab s + NS -> the slot slab s has just left
        char *slot = ring_w + SL * GW_SLOT;
        gw_glds(pa + voff, slot);
        gw_glds(pa + voff + 1024, slot + 1024);
        gw_glds(pw + voff, slot + 16384);
        gw_glds(pw + voff + 1024, slot + 16384 + 1024);
        pa += 2048;
        pw += 2048;
    }
#endif
    // request gw_req_at(k) goes out behind MFMA k of the phase
#define GW_AT(K)                                                                       \
    if constexpr (gw_req_at(K) >= 0) {                                                 \
        if (STEADY || issue) {                                                         \
            constexpr int R_ = gw_req_at(K) >= 0 ? gw_req_at(K) : 0;                   \
            char *slot = ring_w + SL * GW_SLOT;                                        \
            __builtin_amdgcn_sched_barrier(0);                                         \
            if constexpr (R_ < 2) gw_glds(pa + voff + R_ * 1024, slot + R_ * 1024);    \
            else gw_glds(pw + voff + (R_ - 2) * 1024, slot + 16384 + (R_ - 2) * 1024); \
            if constexpr (R_ == 3) {                                                   \
                pa += 2048;                                                            \
                pw += 2048;                                                            \
            }                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                         \
        }                                                                              \
    }
    __builtin_amdgcn_sched_barrier(0);
    // product 1 of 3: w_hi a_hi, the reads of the next slab (two per MFMA) between the first six
    const bool rd = STEADY || next;
    GW_MFMA(acc[0][0], c.wh[0], c.ah[0]);
    if (rd) gw_read2<SN, 0>(n, la, lw);
    GW_AT(0)
    GW_MFMA(acc[1][0], c.wh[1], c.ah[0]);
    if (rd) gw_read2<SN, 1>(n, la, lw);
    GW_AT(1)
    GW_MFMA(acc[0][1], c.wh[0], c.ah[1]);
    if (rd) gw_read2<SN, 2>(n, la, lw);
    GW_AT(2)
    GW_MFMA(acc[1][1], c.wh[1], c.ah[1]);
    if (rd) gw_read2<SN, 3>(n, la, lw);
    GW_AT(3)
    GW_MFMA(acc[0][2], c.wh[0], c.ah[2]);
    if (rd) gw_read2<SN, 4>(n, la, lw);
    GW_AT(4)
    GW_MFMA(acc[1][2], c.wh[1], c.ah[2]);
    if (rd) gw_read2<SN, 5>(n, la, lw);
    GW_AT(5)
    GW_MFMA(acc[0][3], c.wh[0], c.ah[3]);
    GW_AT(6)
    GW_MFMA(acc[1][3], c.wh[1], c.ah[3]);
    GW_AT(7)
    // product 2: w_lo a_hi
    GW_MFMA(acc[0][0], c.wl[0], c.ah[0]);
    GW_AT(8)
    GW_MFMA(acc[1][0], c.wl[1], c.ah[0]);
    GW_AT(9)
    GW_MFMA(acc[0][1], c.wl[0], c.ah[1]);
    GW_AT(10)
    GW_MFMA(acc[1][1], c.wl[1], c.ah[1]);
    GW_AT(11)
    GW_MFMA(acc[0][2], c.wl[0], c.ah[2]);
    GW_AT(12)
    GW_MFMA(acc[1][2], c.wl[1], c.ah[2]);
    GW_AT(13)
    GW_MFMA(acc[0][3], c.wl[0], c.ah[3]);
    GW_AT(14)
    GW_MFMA(acc[1][3], c.wl[1], c.ah[3]);
    GW_AT(15)
    // product 3: w_hi a_lo
    GW_MFMA(acc[0][0], c.wh[0], c.al[0]);
    GW_AT(16)
    GW_MFMA(acc[1][0], c.wh[1], c.al[0]);
    GW_AT(17)
    GW_MFMA(acc[0][1], c.wh[0], c.al[1]);
    GW_AT(18)
    GW_MFMA(acc[1][1], c.wh[1], c.al[1]);
    GW_AT(19)
    GW_MFMA(acc[0][2], c.wh[0], c.al[2]);
    GW_AT(20)
    GW_MFMA(acc[1][2], c.wh[1], c.al[2]);
    GW_AT(21)
    GW_MFMA(acc[0][3], c.wh[0], c.al[3]);
    GW_AT(22)
    GW_MFMA(acc[1][3], c.wh[1], c.al[3]);
#undef GW_AT
}

// The last arriver of a K-split tile: acc <- partial 0 + partial 1 + ... in PART ORDER (its own partial, index PART, from the
// registers it is in; the others from the workspace, `ws` = the tile's slot + tid * 4 floats).  P and PART are template
// parameters so that every register index and every branch is static (a run-time part index costs selects, zero-initialised
// load destinations under uniform branches -- and, measured, 527 spilled registers).
// THE LOADS OF A BATCH AND THEIR WAIT ARE ONE asm STATEMENT: the compiler does not know that an inline-asm load completes
// later; a destination register it moved or spilled between a load statement and a separate wait statement would be read
// before the data has landed (the first version did that under register pressure: wrong sums, r5a).
// Addresses: ONE wave-uniform base per part (an SGPR pair) + four per-lane byte offsets (tid * 16 + j * 8192) shared by every
// part and accumulator tile -- 64-bit per-load VGPR addresses (24 registers for 12 loads) pushed the routine into scratch, and a
// kernel that uses scratch ran into the bounded waits on the 317M model (16 peers per row block; profiles/README.md r5c).
// s_nop 4: a VALU-written SGPR needs 5 wait states before a VMEM instruction reads it (the hazard recognizer does not look
// inside inline asm).
#define GW_LD_ "global_load_dwordx4 %"
__device__ __forceinline__ void gw_ld4(f32x4 (&a)[4], const unsigned (&o)[4], const char *pa) {
    asm volatile("s_nop 4\n\t" GW_LD_ "0, %4, %8 " KS_LD "\n\t" GW_LD_ "1, %5, %8 " KS_LD "\n\t" GW_LD_ "2, %6, %8 " KS_LD "\n\t" GW_LD_ "3, %7, %8 " KS_LD "\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3])
                 : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "s"(pa)
                 : "memory");
}
__device__ __forceinline__ void gw_ld8(f32x4 (&a)[4], f32x4 (&b)[4], const unsigned (&o)[4], const char *pa, const char *pb) {
    asm volatile("s_nop 4\n\t" GW_LD_ "0, %8, %12 " KS_LD "\n\t" GW_LD_ "1, %9, %12 " KS_LD "\n\t" GW_LD_ "2, %10, %12 " KS_LD "\n\t" GW_LD_ "3, %11, %12 " KS_LD "\n\t"
                 GW_LD_ "4, %8, %13 " KS_LD "\n\t" GW_LD_ "5, %9, %13 " KS_LD "\n\t" GW_LD_ "6, %10, %13 " KS_LD "\n\t" GW_LD_ "7, %11, %13 " KS_LD "\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
                 : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "s"(pa), "s"(pb)
                 : "memory");
}
__device__ __forceinline__ void gw_ld12(f32x4 (&a)[4], f32x4 (&b)[4], f32x4 (&c)[4], const unsigned (&o)[4], const char *pa, const char *pb,
                                        const char *pc) {
    asm volatile("s_nop 4\n\t" GW_LD_ "0, %12, %16 " KS_LD "\n\t" GW_LD_ "1, %13, %16 " KS_LD "\n\t" GW_LD_ "2, %14, %16 " KS_LD "\n\t" GW_LD_ "3, %15, %16 " KS_LD "\n\t"
                 GW_LD_ "4, %12, %17 " KS_LD "\n\t" GW_LD_ "5, %13, %17 " KS_LD "\n\t" GW_LD_ "6, %14, %17 " KS_LD "\n\t" GW_LD_ "7, %15, %17 " KS_LD "\n\t"
                 GW_LD_ "8, %12, %18 " KS_LD "\n\t" GW_LD_ "9, %13, %18 " KS_LD "\n\t" GW_LD_ "10, %14, %18 " KS_LD "\n\t" GW_LD_ "11, %15, %18 " KS_LD "\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]),
                   "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                 : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "s"(pa), "s"(pb), "s"(pc)
                 : "memory");
}
#undef GW_LD_
// ws: the tile's workspace slot (wave-uniform); off: tid * 16 + {0, 1, 2, 3} * 8192 bytes.  EVERY partial comes from the
// workspace -- the last arriver's own too (it was stored like the others): the accumulators are then dead across the ticket and
// the routine needs no scratch (keeping the own partial in registers saved a 256 KiB read per tile and cost 380 spilled
// registers in nine (P, part) variants).
template <int P>
__device__ __forceinline__ void gw_reduce(f32x16 (&acc)[2][4], const char *ws, const unsigned (&off)[4]) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const char *b = ws + (size_t)((n * 4 + rt) * 4) * 8192;
            f32x4 A[4], L[3][4];  // part 0; parts 1 .. P - 1
            gw_ld4(A, off, b);
            if constexpr (P == 2) gw_ld4(L[0], off, b + 262144);
            else if constexpr (P == 3) gw_ld8(L[0], L[1], off, b + 262144, b + 2 * 262144);
            else gw_ld12(L[0], L[1], L[2], off, b + 262144, b + 2 * 262144, b + 3 * 262144);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sum = A[j][r];
#pragma unroll
                    for (int q = 1; q < P; ++q) sum = __fadd_rn(sum, L[q - 1][j][r]);  // in part order, whoever arrived last
                    acc[n][rt][4 * j + r] = sum;
                }
        }
}
__device__ __forceinline__ void gw_reduce_any(f32x16 (&acc)[2][4], const char *ws, const unsigned (&off)[4], int P) {
    if (P == 2) gw_reduce<2>(acc, ws, off);
    else if (P == 3) gw_reduce<3>(acc, ws, off);
    else gw_reduce<4>(acc, ws, off);
}

// EPI = 1: LayerNorm + Mish, 2: LayerNorm + SimNorm(8).  Grid / tile order: tile_order.h with 256-row blocks.
// KS = 1: the instantiation with the K-split tail (TDMPC2_TUNE_KSPLIT = 1, not the default: see the measurement in DESIGN 10);
// KS = 0 carries none of its code (no scratch, the round-4 machine code).
template <int EPI, int KS = 0>
__global__ __launch_bounds__(512) void g_gemm_w(GemmSParams p) {
    static_assert(EPI == 1 || EPI == 2, "the wide tile exists for the NormedLinear layers");
    constexpr int NS = GW_NSLOT, TM = 256;
    __shared__ __attribute__((aligned(1024))) char ring[NS * GW_SLOT];
    GW_T0
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;  // this wave's accumulators: row tiles 4 wr .. + 3, column tiles 2 wc, 2 wc + 1
    int rb, cb, part = 0, slot = -1;
    if (KS && p.ks_parts > 1) {
        if (!gemm_w_tile(blockIdx.x, p.nrowblk, p.ncolblk, p.ks_full, p.ks_parts, p.ks_max_tail, rb, cb, part, slot)) return;
    } else if (!gemm_s_tile(blockIdx.x, p.nrowblk, p.ncolblk, p.xcd_rows, p.ncol_grid, rb, cb)) {
        return;
    }
    const int row0 = rb * TM;
    const int sel = p.sel ? p.sel[(size_t)(row0 / p.rows_per_env) * p.sel_stride] : 0;
    // this workgroup's k16-slabs: all of them, or -- a tile of the launch's last, partly filled round -- part `part` of ks_parts
    const int nk_all = p.K / 16;
    const int ks0 = slot >= 0 ? part * nk_all / p.ks_parts : 0;
    const int nk = slot >= 0 ? (part + 1) * nk_all / p.ks_parts - ks0 : nk_all;
    // the epilogue's vectors: in flight behind the prologue's DMA requests (one env per 256-row block: rows_per_env % 256 == 0)
    float vb = 0.f, vg = 0.f, vbe = 0.f;
    if (tid < 256) {
        const int col = cb * 256 + tid;
        if (col < p.CT * 32) {
            const float *bp = p.bias + (size_t)sel * p.bias_sel_stride;
            if (p.bias_env_stride != 0) bp += (size_t)(row0 / p.rows_per_env) * p.bias_env_stride;
            vb = bp[col];
            vg = p.ln_g[(size_t)sel * p.gb_sel_stride + col];
            vbe = p.ln_b[(size_t)sel * p.gb_sel_stride + col];
        }
    }
    // DMA role of this wave: row tile `wave` of A and column tile `wave` of W (a column tile past the matrix re-reads the last one)
    const int ctl = cb * 8 + wave < p.CT ? cb * 8 + wave : p.CT - 1;
    const char *pa = reinterpret_cast<const char *>(p.A) + ((size_t)((row0 >> 5) + wave) * p.KBa + p.a_kb0 + ks0) * 2048;
    const char *pw = reinterpret_cast<const char *>(p.wp + (size_t)sel * p.w_sel_stride) + ((size_t)ctl * (p.kbs ? p.kbs : nk_all) + p.kb0 + ks0) * 2048;
    unsigned voff = (unsigned)lane * 16u;
    asm volatile("" : "+v"(voff));
    char *ring_w = ring + wave * 2048;
    const unsigned lbase = lds_addr_of(ring) + (unsigned)lane * 16u;
    unsigned la[GW_NB], lw[GW_NB];
#pragma unroll
    for (int b = 0; b < GW_NB; ++b) {
        la[b] = lbase + (unsigned)wr * 8192u + 65536u * b;
        lw[b] = lbase + 16384u + (unsigned)wc * 4096u + 65536u * b;
    }

    f32x16 acc[2][4];  // [column tile][row tile]: C = [feature][row] (weight fragment = the MFMA's A operand, as in g_gemm_s<EPI>)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][rt][e] = 0.f;
    GwFrags fr[2];

    // prologue: slabs 0 .. NS - 1 requested; slab 0 into registers
    const int npro = gw_prologue_slabs(nk, NS);
#pragma unroll
    for (int d = 0; d < NS; ++d)
        if (d < npro) {
            char *slot = ring_w + d * GW_SLOT;
            gw_glds(pa + voff, slot);
            gw_glds(pa + voff + 1024, slot + 1024);
            gw_glds(pw + voff, slot + 16384);
            gw_glds(pw + voff + 1024, slot + 16384 + 1024);
            pa += 2048;
            pw += 2048;
        }
    __builtin_amdgcn_sched_barrier(0);
    // slab 0 has landed: gw_prologue_vmcnt(npro) = 4 (npro - 1) requests may stay in flight (the ladder is over npro: s_waitcnt takes constants)
    if (npro >= 5) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (npro == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (npro == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (npro == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    gw_read<0>(fr[0], la, lw);

    int s = 0;
#define GW_STEADY(PH) gw_phase<PH, true>(acc, fr, la, lw, ring_w, pa, pw, voff, true, true, gw_steady_vmcnt(NS));
    // a phase with wave-uniform flags (tile_order.h: gw_tail_step -- the newest slab requested so far is min(nk - 1, ss + NS - 1);
    // slab ss + 1 must have landed)
#define GW_TAIL(PH)                                                                                         \
    if (s + PH < nk) {                                                                                      \
        const GwTailStep ts = gw_tail_step(s + PH, nk, NS);                                                 \
        gw_phase<PH, false>(acc, fr, la, lw, ring_w, pa, pw, voff, ts.issue, ts.next, ts.vmc);              \
    }
#pragma unroll 1
    for (; gw_steady_trip(s, nk, NS, GW_U); s += GW_U) {  // steady state: every phase of the trip has a slab s' + NS <= nk - 1 to request
        GW_STEADY(0) GW_STEADY(1) GW_STEADY(2) GW_STEADY(3)
    }
#pragma unroll 1
    for (; s < nk; s += GW_U) {  // the last phases (and short contractions)
        GW_TAIL(0) GW_TAIL(1) GW_TAIL(2) GW_TAIL(3)
    }
#undef GW_TAIL
#undef GW_STEADY
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // XDL write -> VALU read of the accumulators
    GW_T(1)

    // ---------------------------------------------------------------- K-split tile: partial sums meet in the workspace
    // Every part stores its accumulators in register order (chunk c = ((n 4 + rt) 4 + j): 512 threads x 16 bytes = 8 KiB,
    // 1 KiB contiguous per wave instruction) with write-through stores, drains them, and takes a ticket; the part that draws the
    // last one adds the partials IN PART ORDER and carries on into the epilogue as if it had run the
    // whole contraction.  Nobody waits: the others leave.  The sum differs from the unsplit tile's in the last bits (fp32
    // association), by less than the f16x2 split's own error; TDMPC2_TUNE_KSPLIT = 0 keeps every tile whole.
    if (KS && slot >= 0) {
        const int P = p.ks_parts;
        float *ws = p.ks_ws + (size_t)slot * P * 65536 + tid * 4;
        {
            float *mine = ws + (size_t)part * 65536;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[n][rt][4 * j + r];
                        gw_st_sc1(mine + ((n * 4 + rt) * 4 + j) * 2048, v);
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave's stores are acknowledged (and every wave is done with the ring)
        int *flag = reinterpret_cast<int *>(ring);
        if (tid == 0) *flag = (int)__hip_atomic_fetch_add(p.ks_cnt + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != P - 1) return;
        {
            const char *wsu = reinterpret_cast<const char *>(p.ks_ws + (size_t)slot * P * 65536);  // wave-uniform
            unsigned off[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) off[j] = (unsigned)tid * 16u + (unsigned)j * 8192u;
            gw_reduce_any(acc, wsu, off, P);
        }
    }

    // ---------------------------------------------------------------- NormedLinear epilogue (the protocol of g_gemm_s<.., EPI>)
    const int i32 = lane & 31, hh = lane >> 5;
    const int ct0 = cb * 8 + wc * 2;  // this wave's column tiles
    const float osc = p.oscale[(size_t)sel * p.osc_sel_stride];
    __syncthreads();  // every wave is done with the ring
    float *red = reinterpret_cast<float *>(ring);  // [8 column tiles of the block][TM][2]
    float *rs = red + 8 * TM * 2;                   // [TM][2]
    // the block's 256 columns of bias / LayerNorm weight / LayerNorm bias: requested before the main loop (three registers ride
    // through it), read by the epilogue from LDS instead of 96 dependent 16-byte global loads per lane
    float *vecs = rs + TM * 2;                      // [bias | g | b][256]
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
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[n][rt][4 * j + r] = fmaf(acc[n][rt][4 * j + r], osc, b4[r]);
        }
    // (2) (mean, M2) of every row over each 32-column tile: 16 thread-local values + the lane ^ 32 half
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float s1 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) s1 += acc[n][rt][e];
            s1 += __shfl_xor(s1, 32);
            const float mw = s1 * (1.f / 32.f);
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d = acc[n][rt][e] - mw;
                q = fmaf(d, d, q);
            }
            q += __shfl_xor(q, 32);
            if (hh == 0) {
                red[((wc * 2 + n) * TM + wr * 128 + rt * 32 + i32) * 2 + 0] = mw;
                red[((wc * 2 + n) * TM + wr * 128 + rt * 32 + i32) * 2 + 1] = q;
            }
        }
    __syncthreads();
    // (3) fold the tiles of each 128-column group left to right -> stats[row block][group][row]: 2 groups x 256 rows = one per thread
    const int NG = (p.CT + 3) / 4;  // groups of the whole row
    {
        const int g = tid >> 8, r = tid & 255, G = cb * 2 + g;
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
    GW_T(2)
    // (4) arrive (stores acknowledged first), wait for the row block's other column blocks -- bounded
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (!(p.fault && rb == 0 && cb == 0)) __hip_atomic_fetch_add(p.arrive + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        WaitClock wc;
        while (__hip_atomic_load(p.arrive + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.ncolblk) {
            if (wc.expired(p.err)) {
                // code: 2 | row block << 4 | column block << 16 | arrivals seen << 24 (TDMPC2_DEBUG_FAULT=1 prints it)
                if (p.err) raise_fault(p.err, 2u | ((unsigned)rb << 4) | ((unsigned)cb << 16) |
                                                  (__hip_atomic_load(p.arrive + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 24));
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    GW_T(3)
    // (5) the row's statistics: the groups folded left to right (loads eight at a time, the fold in order)
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
    GW_T(4)
    // (6) normalise, activate, split -> the fragment-packed output.  A lane holds features 8 j + 4 hh + (0..3) of row i32 of a
    // column tile.  For the pair (j, j + 1) = the two k-halves of k16-block 2 ct + (j >> 1), v_permlane32_swap hands the lower
    // half-wave the upper one's 4 features of k-half 0 and the upper half-wave the lower one's of k-half 1: every lane then
    // holds 8 consecutive features (16 bytes per plane) of ITS k-half, lane' = 32 hh + i32 = lane: one store = one 1 KiB plane.
    const float oscl = EPI == 1 ? p.ascale[(size_t)sel * p.asc_sel_stride] : ACT_SCALE;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const float rmean = rs[2 * (wr * 128 + rt * 32 + i32)], rrstd = rs[2 * (wr * 128 + rt * 32 + i32) + 1];
        char *otile = reinterpret_cast<char *>(p.out) + (size_t)((row0 >> 5) + wr * 4 + rt) * p.KBo * 2048 + lane * 16;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (ct0 + n >= p.CT) continue;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                unsigned hw[2][2], lw2[2][2];  // [j - 2 jp][dword]: hi / lo planes of this lane's 4 features
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * jp + jj;
                    const f32x4 g4 = *reinterpret_cast<const f32x4 *>(vecs + 256 + (wc * 2 + n) * 32 + 8 * j + 4 * hh);
                    const f32x4 be4 = *reinterpret_cast<const f32x4 *>(vecs + 512 + (wc * 2 + n) * 32 + 8 * j + 4 * hh);
                    f32x4 y;
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = fmaf((acc[n][rt][4 * j + r] - rmean) * rrstd, g4[r], be4[r]);
                    if (EPI == 1) {
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
                // swap: lower half-wave <- upper's k-half-0 features, upper half-wave <- lower's k-half-1 features
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 ho, lo4;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const auto sh = __builtin_amdgcn_permlane32_swap(hw[0][d], hw[1][d], false, false);
                    const auto sl = __builtin_amdgcn_permlane32_swap(lw2[0][d], lw2[1][d], false, false);
                    // after the swap: element 0 = features 0..3 of this lane's k-half, element 1 = features 4..7
                    ho[d] = sh[0]; ho[2 + d] = sh[1];
                    lo4[d] = sl[0]; lo4[2 + d] = sl[1];
                }
                char *o = otile + (size_t)((ct0 + n) * 2 + jp) * 2048;
                *reinterpret_cast<u32x4 *>(o) = ho;
                *reinterpret_cast<u32x4 *>(o + 1024) = lo4;
            }
        }
    }
    GW_T(5)
    GW_TFLUSH
}

