// Fused 512-wide planner kernels: the f16 matrix pipe with fp32-class accuracy ("f16x2 split", ARITH 0, the default)
// and exact fp32 MFMA (ARITH 1) share every phase; only the contraction loops and the LDS tile form differ.
//
// The f16x2 split:
// Every fp32 operand x is carried as two f16 pieces, x ~ hi + lo with hi = f16(x), lo = f16(x - hi): 22 significand
// bits.  A product a.b is formed from THREE v_mfma_f32_32x32x16_f16 instructions
//         a_hi.b_hi + a_hi.b_lo + a_lo.b_hi                       (the dropped a_lo.b_lo term is ~2^-22 relative)
// accumulated in fp32.  The matrix pipe issues f16 MFMAs at 16x the rate of v_mfma_f32_32x32x2_f32, so the contraction
// runs at up to 16/3 of the exact-fp32 kernels' roof while the error against an fp64 evaluation of the same network
// stays in the fp32 round-off class (measured: tests/test_gpu_planner.py::test_error_attribution_against_fp64; CPU emulation: oracle/split_probe.py).
// Operands are pre-scaled by powers of two so that the lo pieces stay in the f16 normal range: activations by 2^5,
// each weight matrix by 2^kw with max|W| 2^kw in [2^13, 2^14) (k_wscale); the fp32 accumulator is multiplied back by
// the exact 2^-(kw+5) in the epilogue.
//
// Structure (reference: TDMPC2._plan / _estimate_value, tdmpc2/tdmpc2.py:122-206; WorldModel.next / reward / pi / Q,
// tdmpc2/common/world_model.py:114-216; NormedLinear / SimNorm, tdmpc2/common/layers.py:74-118; math.py:12-94):
// the LDS tile is in operand form (per row [hi plane | lo plane], f16, compile-time strides); the MFMAs
// are issued with the weight fragment as the A operand so that each lane's accumulators belong to its own two sample
// rows, which makes the LayerNorm / activation / hi-lo-split epilogue register resident (no fp32 staging pass, two
// barriers per layer); weight fragments are prefetched through a register ring whose issue order is pinned with
// sched_barrier; 32- or 64-row workgroups (ST).  DESIGN.md section 3.2 has the measured breakdown.
// Included by tdmpc2_plan.hip inside its anonymous namespace.
#pragma once

#include "kloop_schedule.h"  // kloop_asm's schedule arithmetic (shared with tests/test_ring_schedule.py)

// In-kernel phase timers (profiling builds only: -DSPLIT_TIMING, see tools/ablate.sh): wave 0 of every workgroup sums
// the shader-clock cycles it spends in each phase class into p.timing[class].
#ifdef SPLIT_TIMING
#define TIMER_FIELDS mutable unsigned long long t_acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; mutable unsigned long long t_last = 0, t_begin = 0;
#define TIMER_START(c) { (c).t_last = (c).t_begin = __builtin_amdgcn_s_memtime(); }
#define TIMER_MARK(c, cls) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_now = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); (c).t_acc[cls] += t_now - (c).t_last; (c).t_last = t_now; }
// the wave whose clock is read: 0-3 are the first-dispatched (older) wave of their SIMD, 4-7 the younger one, which loses
// the MFMA arbitration and is the critical path of every k-loop
#ifndef SPLIT_TIMING_WAVE
#define SPLIT_TIMING_WAVE 4
#endif
#define TIMER_FLUSH(c, ptr) if ((ptr) && threadIdx.x == 64 * SPLIT_TIMING_WAVE) { for (int i_ = 0; i_ < 14; ++i_) atomicAdd((ptr) + i_, (c).t_acc[i_]); atomicAdd((ptr) + 14, (c).t_last - (c).t_begin); atomicAdd((ptr) + 15, 1ull); }
#else
#define TIMER_FIELDS
#define TIMER_START(c)
#define TIMER_MARK(c, cls)
#define TIMER_FLUSH(c, ptr)
#endif
enum { T_KLOOP = 0, T_EPI = 1, T_HEAD = 2, T_ACT = 3, T_PARK = 4, T_TILE = 5, T_EPI_PRE = 6, T_EPI_SYNC = 7, T_EPI_BIAS = 8,
       T_EPI_COMB = 9, T_EPI_MATH = 10, T_CL_WAIT = 11, T_HEAD_K = 12, T_HEAD_ST = 13 };  // (head = what is left: the row routines)



// The action padding (16, 32, 48 or 64 columns) is a template parameter: with compile-time row strides every LDS
// address of the unrolled epilogues is base + immediate; with a run-time stride the compiler pre-computes hundreds of
// addresses, hoists them out of the step loop and spills them.
// ST = 32-row sample tiles per workgroup: 2 (64 rows, throughput) or 1 (32 rows: twice the workgroups for the same plans,
// used when a call has too few plans to fill the chip -- single-environment latency).
// NW = wavefronts per workgroup: 8 (each owns 64 output features = 2 feature tiles); the code is written for 4 as well
// (128 features = 4 tiles per wave), a geometry that was measured and lost both as two 32-row workgroups per CU and
// as one 64-row workgroup per CU (profiles/README.md).
// ARITH: 0 = f16x2 split (operand tile = hi / lo f16 planes), 1 = exact fp32 (v_mfma_f32_32x32x2_f32; operand tile =
// fp32 rows of WIDTH + APAD + 4 floats, stride / 4 odd).  Everything but the contraction loops and the tile writes is
// shared between the two.
template <int APAD, int ST = 2, int NW = 8, int AR = 0>
struct CtxT {
    static constexpr int NST = ST;
    static constexpr int NWAVES = NW;
    static constexpr int ARITH = AR;
    static constexpr int FT = 16 / NW;       // 32-wide output feature tiles per wave
    static constexpr int NTHR = 64 * NW;     // threads per workgroup
    static constexpr int TROWS = 32 * ST;    // sample rows per workgroup
    static constexpr int SH = WIDTH + APAD;  // split: plane length in halfs
    static constexpr int RSH = 2 * SH + 8;   // split: row stride in halfs (row stride in dwords = SH + 4 = 4 x odd)
    static constexpr int RSF_ = AR == 0 ? RSH / 2 : WIDTH + APAD + 4;  // row stride in floats of either form
    static constexpr int KBLK = AR == 0 ? 16 : 8;                      // contraction block
    static constexpr int ZKB = WIDTH / KBLK;                           // blocks covering the latent columns
    _Float16 *act;  // LDS tile.  split: row r at act + r * RSH: [hi: SH halfs | lo: SH halfs | 8 pad]; fp32: actf() rows
    float *stats;   // LDS [NW waves][TROWS][2]: per-wave LayerNorm partials
    float *gb;      // LDS [2][WIDTH]: LayerNorm weight / bias of the layer whose epilogue comes next (see gb_prefetch)
    int tid, wave, lane;
    mutable float pg[(WIDTH + NTHR - 1) / NTHR], pb[(WIDTH + NTHR - 1) / NTHR];  // the successor layer's values in flight
    TIMER_FIELDS
    __device__ __forceinline__ float *f32() const { return reinterpret_cast<float *>(act); }  // fp32 view [TROWS][RSF]
    static constexpr __device__ __forceinline__ int RSF() { return RSF_; }
};

__device__ __forceinline__ float mish_fast(float x) {
    // x * tanh(softplus(x)) = x * n / (n + 2), n = e^x (e^x + 2); for x > 20 the ratio rounds to 1 in fp32, so the
    // exponent is clamped instead of branching (tdmpc2/common/layers.py:103)
    const float e = __expf(fminf(x, 20.f));
    const float n = e * (e + 2.f);
    return x * (n * __builtin_amdgcn_rcpf(n + 2.f));  // v_rcp_f32: 1 ulp; __fdividef expands to the full division
}

// exact-arithmetic flavours of the transcendental pieces (ARITH 1 keeps libm-accurate expf and IEEE division)
template <class CT>
__device__ __forceinline__ float exp_a(float x) {
    if constexpr (CT::ARITH == 1) return expf(x);
    else return __expf(x);
}
template <class CT>
__device__ __forceinline__ float mish_a(float x) {
    if constexpr (CT::ARITH == 1) {
        const float e = expf(fminf(x, 20.f));
        const float n = e * (e + 2.f);
        return x * (n / (n + 2.f));
    } else {
        return mish_fast(x);
    }
}

__device__ __forceinline__ void split4(const f32x4 y, f16x4 &hi, f16x4 &lo, const float scale = ACT_SCALE) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ys = y[e] * scale;
        const _Float16 h = (_Float16)ys;
        hi[e] = h;
        lo[e] = (_Float16)(ys - (float)h);
    }
}


// ---------------------------------------------------------------- contraction loops
// v_mfma_f32_32x32x16_f16: lane l supplies A[i = l & 31][k = 8 (l >> 5) .. +7] and B[k = 8 (l >> 5) .. +7][j = l & 31].
// Wave w of 8 owns output columns [64 w, 64 w + 64) for both 32-row tiles: acc[set][row tile][col tile].
// B fragments are prefetched PF k-blocks ahead (PF x 16 VGPRs) through a register ring.
constexpr int PF = 2;  // measured: 2, 3 and 4 k-blocks of prefetch run within 1 % of each other; 6 spills
template <int FT>
struct BFragT {
    f16x8 h[FT], l[FT];
};
// B-fragment addressing: wave-uniform byte pointers (SGPR pairs, advanced by scalar arithmetic) + ONE 32-bit lane
// offset, made opaque so that the compiler cannot fully unroll the k-loop into per-k-block 64-bit VGPR addresses and
// hoist them out of the step loop (that cost ~1000 spilled registers).
// (The ablation builds of rounds 1-4 -- no MFMA, no weight loads, no epilogue math, ring depths 4 / 6, the compiler-scheduled
// loop -- live in tools/variants/ as patches; the shipped kernels carry no experiment switches.)
__device__ __forceinline__ f16x8 ldw(const char *ubase, unsigned voff, int imm) {
    return *reinterpret_cast<const f16x8 *>(ubase + voff + imm);
}
#define SPLIT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
template <int FT>
__device__ __forceinline__ void load_b(BFragT<FT> &b, const char *u0, size_t ct_stride, unsigned voff) {
#pragma unroll
    for (int cc = 0; cc < FT; ++cc) {
        b.h[cc] = ldw(u0 + cc * ct_stride, voff, 0);
        b.l[cc] = ldw(u0 + cc * ct_stride, voff, 1024);
    }
}
template <int ST, int FT>
__device__ __forceinline__ void zero_acc(f32x16 (&a)[ST][FT]) {
#pragma unroll
    for (int r = 0; r < ST; ++r)
#pragma unroll
        for (int cc = 0; cc < FT; ++cc)
#pragma unroll
            for (int e = 0; e < 16; ++e) a[r][cc][e] = 0.f;
}
template <int ST>
struct AFragT {
    f16x8 h[ST], l[ST];  // hi / lo pieces of the 32-row sample tiles
};
template <class CT>
__device__ __forceinline__ void load_a(AFragT<CT::NST> &a, const _Float16 *a0p, int kk) {
#pragma unroll
    for (int st = 0; st < CT::NST; ++st) {
        a.h[st] = *reinterpret_cast<const f16x8 *>(a0p + st * 32 * CT::RSH + kk * 16);
        a.l[st] = *reinterpret_cast<const f16x8 *>(a0p + st * 32 * CT::RSH + CT::SH + kk * 16);
    }
}
// Software pipeline, per k-block: [read the NEXT block's activation fragments from LDS] [12 MFMAs on the current
// block] [issue the weight loads for block + PF into the ring slot just consumed] [sched_barrier].  The barrier pins
// the order: without it the machine scheduler sinks the prefetch loads to their first use (the next outer iteration)
// and the loop runs load -> wait -> compute with no overlap.
// exact-fp32 form of the same loop: v_mfma_f32_32x32x2_f32, weight fragment (one 16-byte read = 4 MFMAs' operands, k pairs
// {8 kb + r, 8 kb + 4 + r}) as the A operand, activation fragment from the fp32 tile as the B operand.
template <class CT>
__device__ __forceinline__ void kloop_f32(const CT &c, const LayerS &ly, int kb0, int kb1, f32x16 (&acc)[CT::NST][CT::FT]) {
    constexpr int FT = CT::FT, NST = CT::NST;
    constexpr int PFD = 4;
    const int i = c.lane & 31, hh = c.lane >> 5;
    const float *a0p = c.f32() + i * CT::RSF() + 4 * hh + kb0 * 8;
    const char *u0 = reinterpret_cast<const char *>(ly.wp) + ((size_t)(FT * c.wave) * ly.KB + kb0) * 1024;  // 64 lanes x 16 B
    const size_t cts = (size_t)ly.KB * 1024;
    unsigned voff = (unsigned)c.lane * 16u;
    asm volatile("" : "+v"(voff));
    const int nk = kb1 - kb0;
    f32x4 ring[PFD][FT];
#pragma unroll
    for (int d = 0; d < PFD; ++d) {
        const int kd = d < nk ? d : nk - 1;
#pragma unroll
        for (int cc = 0; cc < FT; ++cc) ring[d][cc] = *reinterpret_cast<const f32x4 *>(u0 + (size_t)kd * 1024 + cc * cts + voff);
    }
    f32x4 an[NST];
#pragma unroll
    for (int st = 0; st < NST; ++st) an[st] = *reinterpret_cast<const f32x4 *>(a0p + st * 32 * CT::RSF());
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int k = 0; k < nk; k += PFD) {  // (a branch-free steady-state trip as in kloop_tile_s / g_gemm_s was measured here: -0.5 %)
#pragma unroll
        for (int d = 0; d < PFD; ++d) {
            const int kk = k + d;
            if (kk < nk) {  // wave-uniform
                f32x4 a[NST];
                const int kx = kk + 1 < nk ? kk + 1 : kk;
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    a[st] = an[st];
                    an[st] = *reinterpret_cast<const f32x4 *>(a0p + st * 32 * CT::RSF() + kx * 8);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int cc = 0; cc < FT; ++cc)
#pragma unroll
                        for (int st = 0; st < NST; ++st)
                            acc[st][cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[d][cc][r], a[st][r], acc[st][cc], 0, 0, 0);
                const int kn = kk + PFD < nk ? kk + PFD : nk - 1;
#pragma unroll
                for (int cc = 0; cc < FT; ++cc)
                    ring[d][cc] = *reinterpret_cast<const f32x4 *>(u0 + (size_t)kn * 1024 + cc * cts + voff);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// ---------------------------------------------------------------- hand-ordered contraction loop (split arithmetic)
// hipcc schedules the C++ loop above as [12 MFMA] [4 weight loads] [4 LDS reads of the NEXT block] -> the LDS reads sit
// right in front of their first use (the register coalescer merges "next" into "current" fragments; sched_group_barrier
// does not move them).  tools/probes/order_probe.hip: with the same traffic, issuing the LDS reads FIRST is 3 % faster, and
// the probe's loop -- no per-step VALU address arithmetic, no partial waits -- runs at 0.89 of the matrix pipe (at the
// power-managed clock) against 0.71 for the compiler's loop inside the kernel.  So the loop is written out: every LDS
// read, weight load and MFMA of a step is an `asm volatile` (source order = issue order), weights through SGPR base +
// one VGPR lane offset (no 64-bit VALU adds), activations from two fragment sets used alternately, explicit s_waitcnt:
//     top of step kk:  lgkmcnt(0)            this block's activation fragments (issued a whole step earlier)
//                      4 ds_read_b128        block kk + 1 -> the other fragment set
//                      vmcnt(4 | 0)          this block's weight fragments (all but the 4 newer loads of block kk + 1)
//                      12 MFMA
//                      4 global_load_dwordx4 block kk + 2 -> this block's ring slot
// Counter hygiene: only "all but the newest N of MY OWN loads" waits are used for vmcnt (loads return in order, anything
// the compiler still has in flight is older: the wait is merely conservative), lgkmcnt is only ever waited to 0 (scalar
// loads share that counter and return out of order), no load is in flight when the loop ends (the compiler reuses the
// destination registers), and 16 wait states separate the last MFMA from the compiler's first read of an accumulator.
#define A_MFMA(ACC, WF, AF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(WF), "v"(AF))
template <int OFF>
__device__ __forceinline__ void a_dsrd(f16x8 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ unsigned lds_addr_of(const void *p) {  // byte address inside the workgroup's LDS allocation
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)(p);
}
// one block's activation fragments (hi / lo planes of the NST row tiles) from LDS byte address `la` (+ 72 KB for tile 1)
template <class CT>
__device__ __forceinline__ void a_load_act(AFragT<CT::NST> &a, unsigned la0, unsigned la1) {
    a_dsrd<0>(a.h[0], la0);
    a_dsrd<2 * CT::SH>(a.l[0], la0);
    if constexpr (CT::NST == 2) {
        a_dsrd<0>(a.h[1], la1);
        a_dsrd<2 * CT::SH>(a.l[1], la1);
    }
}
// one k-block of two column tiles: hi plane, lo plane (+1024 B) of tile 0 at `s0`, of tile 1 at `s1`.  ONE asm statement,
// opened by s_nop 4: the base pointers are often restored from spill lanes (v_readlane / v_readfirstlane: VALU writes of
// an SGPR) right in front of the loads, and a VMEM instruction that reads such an SGPR needs 5 wait states -- the
// compiler's hazard recognizer does not look inside inline asm (found the hard way: memory faults, tools/probes/saddr_test.hip).
__device__ __forceinline__ void a_load_w(BFragT<2> &b, unsigned voff, const char *s0, const char *s1) {
    asm volatile("s_nop 4\n\t"
                 "global_load_dwordx4 %0, %4, %5\n\t"
                 "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                 "global_load_dwordx4 %2, %4, %6\n\t"
                 "global_load_dwordx4 %3, %4, %6 offset:1024"
                 : "=&v"(b.h[0]), "=&v"(b.l[0]), "=&v"(b.h[1]), "=&v"(b.l[1])
                 : "v"(voff), "s"(s0), "s"(s1));
}
template <class CT>
__device__ __forceinline__ void a_mfma12(f32x16 (&acc)[CT::NST][2], const BFragT<2> &w, const AFragT<CT::NST> &a) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int st = 0; st < CT::NST; ++st) A_MFMA(acc[st][cc], w.h[cc], a.h[st]);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int st = 0; st < CT::NST; ++st) A_MFMA(acc[st][cc], w.l[cc], a.h[st]);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int st = 0; st < CT::NST; ++st) A_MFMA(acc[st][cc], w.h[cc], a.l[st]);
}

template <class CT>
__device__ __forceinline__ void kloop_asm(const CT &c, const LayerS &ly, int kb0, int kb1, f32x16 (&acc)[CT::NST][2]) {
    static_assert(CT::FT == 2 && CT::ARITH == 0, "hand-ordered loop: 8-wave split-arithmetic geometry");
    const int i = c.lane & 31, hh = c.lane >> 5;
    unsigned la0 = lds_addr_of(c.act) + (unsigned)((i * CT::RSH + 8 * hh + kb0 * 16) * 2);
    unsigned la1 = la0 + 32 * CT::RSH * 2;
    const char *u0 = reinterpret_cast<const char *>(ly.wp) + ((size_t)(2 * c.wave) * ly.KB + kb0) * 2048;  // wave-uniform
    const size_t cts = (size_t)ly.KB * 2048;
    const unsigned voff = (unsigned)c.lane * 16u;
    const int nk = kb1 - kb0;
    // weight ring of RD = 2 k-blocks (16 VGPRs each): at step k block k + 1 is in flight behind the 12 MFMAs of block k (768 cycles
    // with the partner wave's).  (A ring of 4 lost 3.3 %: the kernel fills its VGPR budget and the loop does not wait for L2
    // latency -- profiles/r4w_kloop_ring4_ab.txt.)
    constexpr int RD = KL_RD;  // kloop_schedule.h: the schedule arithmetic, replayed on the CPU by tests/test_ring_schedule.py
    BFragT<2> ring[RD];
    AFragT<CT::NST> a2[2];
#pragma unroll
    for (int d = 0; d < RD; ++d)
        if (d < nk) a_load_w(ring[d], voff, u0 + d * 2048, u0 + cts + d * 2048);
    a_load_act<CT>(a2[0], la0, la1);
    int k = 0;
    const char *pn = u0 + RD * 2048;  // block k + RD
    // steady state, RD steps per trip (ring slot = step % RD, fragment set = step parity), no conditionals
#pragma unroll 1
    for (; kl_steady_trip(k, nk, RD); k += RD) {
#pragma unroll
        for (int d = 0; d < RD; ++d) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            la0 += 32;
            la1 += 32;
            a_load_act<CT>(a2[(d & 1) ^ 1], la0, la1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            a_mfma12<CT>(acc, ring[d], a2[d & 1]);
            a_load_w(ring[d], voff, pn, pn + cts);
            pn += 2048;
        }
    }
    // the last RD .. 2 RD - 1 steps
#pragma unroll 1
    for (; k < nk; k += RD) {
#pragma unroll
        for (int d = 0; d < RD; ++d) {
            const int kk = k + d;
            if (kk < nk) {  // wave-uniform
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (kl_next_act(kk, nk)) {
                    la0 += 32;
                    la1 += 32;
                    a_load_act<CT>(a2[(d & 1) ^ 1], la0, la1);
                }
                // blocks behind block kk that are still in flight: those issued and not yet consumed
                const int behind = kl_behind(kk, nk, RD);
                if (behind >= 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                else if (behind == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (behind == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (behind == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                a_mfma12<CT>(acc, ring[d], a2[d & 1]);
                if (kl_issue(kk, nk, RD)) {
                    a_load_w(ring[d], voff, pn, pn + cts);
                    pn += 2048;
                }
            }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // XDL write -> VALU read of the accumulators
}

template <class CT>
__device__ __forceinline__ void kloop_s(const CT &c, const LayerS &ly, int kb0, int kb1, f32x16 (&acc)[CT::NST][CT::FT]) {
    if constexpr (CT::ARITH == 1) {
        kloop_f32(c, ly, kb0, kb1, acc);
        return;
    }
    if constexpr (CT::ARITH == 0 && CT::FT == 2) {
        kloop_asm(c, ly, kb0, kb1, acc);
        return;
    }
    constexpr int FT = CT::FT;
    constexpr int PFD = FT == 2 ? PF : (PF > 1 ? PF / 2 : 1);  // ring depth in k-blocks: FT x 8 VGPRs per block
    const int i = c.lane & 31, hh = c.lane >> 5;
    const _Float16 *a0p = c.act + i * c.RSH + 8 * hh + kb0 * 16;
    // one k-block of one column tile = 2 planes x 64 lanes x 16 B = 2048 B
    const char *u0 = reinterpret_cast<const char *>(ly.wp) + ((size_t)(FT * c.wave) * ly.KB + kb0) * 2048;
    const size_t cts = (size_t)ly.KB * 2048;
    unsigned voff = (unsigned)c.lane * 16u;
    asm volatile("" : "+v"(voff));
    const int nk = kb1 - kb0;
    BFragT<FT> ring[PFD];
#pragma unroll
    for (int d = 0; d < PFD; ++d) {
        const int kd = d < nk ? d : nk - 1;
        load_b(ring[d], u0 + (size_t)kd * 2048, cts, voff);
    }
    AFragT<CT::NST> an;
    load_a<CT>(an, a0p, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int k = 0; k < nk; k += PFD) {
#pragma unroll
        for (int d = 0; d < PFD; ++d) {
            const int kk = k + d;
            if (kk < nk) {  // wave-uniform
                const AFragT<CT::NST> a = an;
                load_a<CT>(an, a0p, kk + 1 < nk ? kk + 1 : kk);
#pragma unroll
                for (int cc = 0; cc < FT; ++cc)
#pragma unroll
                    for (int st = 0; st < CT::NST; ++st) acc[st][cc] = SPLIT_MFMA(ring[d].h[cc], a.h[st], acc[st][cc]);
#pragma unroll
                for (int cc = 0; cc < FT; ++cc)
#pragma unroll
                    for (int st = 0; st < CT::NST; ++st) acc[st][cc] = SPLIT_MFMA(ring[d].l[cc], a.h[st], acc[st][cc]);
#pragma unroll
                for (int cc = 0; cc < FT; ++cc)
#pragma unroll
                    for (int st = 0; st < CT::NST; ++st) acc[st][cc] = SPLIT_MFMA(ring[d].h[cc], a.l[st], acc[st][cc]);
                const int kn = kk + PFD < nk ? kk + PFD : nk - 1;
                load_b(ring[d], u0 + (size_t)kn * 2048, cts, voff);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// Narrow output layers (two-hot / policy heads): wave `ct` < CT computes column tile ct for BOTH 32-row tiles, so every
// weight fragment is fetched once per workgroup and feeds 6 MFMAs; six independent accumulators (row tile x product
// kind) keep the matrix pipe issuing back to back.  Activations are the A operand here: C[sample][logit].
template <class CT>
__device__ __forceinline__ void kloop_tile_s(const CT &c, const LayerS &ly, int ct, int kb0, int kb1, f32x16 (&out)[CT::NST]) {
    if constexpr (CT::ARITH == 1) {  // exact fp32: activations as the A operand, one accumulator per row tile
        constexpr int NST = CT::NST, PFD = 4;
        const int i = c.lane & 31, hh = c.lane >> 5;
        const float *a0p = c.f32() + i * CT::RSF() + 4 * hh + kb0 * 8;
        const char *u = reinterpret_cast<const char *>(ly.wp) + ((size_t)ct * ly.KB + kb0) * 1024;
        unsigned voff = (unsigned)c.lane * 16u;
        asm volatile("" : "+v"(voff));
        const int nk = kb1 - kb0;
#pragma unroll
        for (int r = 0; r < NST; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) out[r][e] = 0.f;
        f32x4 ring[PFD];
#pragma unroll
        for (int d = 0; d < PFD; ++d) ring[d] = *reinterpret_cast<const f32x4 *>(u + (size_t)(d < nk ? d : nk - 1) * 1024 + voff);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int k = 0; k < nk; k += PFD) {
#pragma unroll
            for (int d = 0; d < PFD; ++d) {
                const int kk = k + d;
                if (kk < nk) {
                    f32x4 a[NST];
#pragma unroll
                    for (int st = 0; st < NST; ++st) a[st] = *reinterpret_cast<const f32x4 *>(a0p + st * 32 * CT::RSF() + kk * 8);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int st = 0; st < NST; ++st)
                            out[st] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st][r], ring[d][r], out[st], 0, 0, 0);
                    const int kn = kk + PFD < nk ? kk + PFD : nk - 1;
                    ring[d] = *reinterpret_cast<const f32x4 *>(u + (size_t)kn * 1024 + voff);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        return;
    }
    constexpr int PFT = 4;  // weight blocks in flight (2 / 4 / 8 measured equal: profiles/r4zb_head_prefetch_depth_ab.txt)
    const int i = c.lane & 31, hh = c.lane >> 5;
    const _Float16 *a0p = c.act + i * c.RSH + 8 * hh + kb0 * 16;
    const char *u = reinterpret_cast<const char *>(ly.wp) + ((size_t)ct * ly.KB + kb0) * 2048;  // uniform
    unsigned voff = (unsigned)c.lane * 16u;
    asm volatile("" : "+v"(voff));
    const int nk = kb1 - kb0;
    f32x16 acc[CT::NST][3];
#pragma unroll
    for (int r = 0; r < CT::NST; ++r)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][t][e] = 0.f;
    f16x8 rh[PFT], rl[PFT];
#pragma unroll
    for (int d = 0; d < PFT; ++d) {
        const int kd = d < nk ? d : nk - 1;
        rh[d] = ldw(u + (size_t)kd * 2048, voff, 0);
        rl[d] = ldw(u + (size_t)kd * 2048, voff, 1024);
    }
    AFragT<CT::NST> an;
    load_a<CT>(an, a0p, 0);
    __builtin_amdgcn_sched_barrier(0);
    // one k-block; `steady`: compile-time true in the main loop, whose trips contain no conditional (with the range tests
    // inside, hipcc's wait-count insertion loses track of the queue at every join and waits for nearly all loads: the
    // lesson of g_gemm_s, profiles/README.md r03h)
    auto step = [&](const int kk, const int d, const bool steady) __attribute__((always_inline)) {
        const AFragT<CT::NST> a = an;
        load_a<CT>(an, a0p, (steady || kk + 1 < nk) ? kk + 1 : kk);
#pragma unroll
        for (int st = 0; st < CT::NST; ++st) acc[st][0] = SPLIT_MFMA(a.h[st], rh[d], acc[st][0]);
#pragma unroll
        for (int st = 0; st < CT::NST; ++st) acc[st][1] = SPLIT_MFMA(a.h[st], rl[d], acc[st][1]);
#pragma unroll
        for (int st = 0; st < CT::NST; ++st) acc[st][2] = SPLIT_MFMA(a.l[st], rh[d], acc[st][2]);
        const int kn = (steady || kk + PFT < nk) ? kk + PFT : nk - 1;
        rh[d] = ldw(u + (size_t)kn * 2048, voff, 0);
        rl[d] = ldw(u + (size_t)kn * 2048, voff, 1024);
        __builtin_amdgcn_sched_barrier(0);
    };
    int k = 0;
#pragma unroll 1
    for (; k + 2 * PFT <= nk; k += PFT) {  // every block has a successor and a block PFT ahead
#pragma unroll
        for (int d = 0; d < PFT; ++d) step(k + d, d, true);
    }
#pragma unroll 1
    for (; k < nk; k += PFT) {
#pragma unroll
        for (int d = 0; d < PFT; ++d)
            if (k + d < nk) step(k + d, d, false);
    }
#pragma unroll
    for (int r = 0; r < CT::NST; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) out[r][e] = acc[r][0][e] + (acc[r][1][e] + acc[r][2][e]);
}

// ---------------------------------------------------------------- register-resident epilogue of the 512-wide layers
// kloop_s issues the MFMAs with the WEIGHT fragment as the A operand, so the 32x32 C tile is C[feature][sample]:
// lane (j = l & 31, hh = l >> 5) holds, for sample rows 32 st + j (st = 0, 1), the features
//     F(ft, m, r) = 64 wave + 32 ft + 8 m + 4 hh + r        (ft = 0, 1; m = reg >> 2; r = reg & 3)
// i.e. half of the wave's 64 output features of that row; lane l ^ 32 holds the other half.  LayerNorm statistics are
// therefore thread-local sums plus ONE cross-lane exchange, combined over the 8 waves through a 4 KB LDS table with
// Chan's parallel-variance formula; normalisation, activation and the hi/lo split run on the accumulators in place
// and the operand form is written straight back to the LDS tile: no fp32 staging pass, 2 barriers per layer.
// LayerNorm: biased variance, eps 1e-5 (layers.py:101).  ACT 0 Mish, 1 SimNorm over 8 consecutive features
// (layers.py:84-88) = this lane's 4 + the partner lane's 4.

// raw accumulators <-> a dense global tile in register order (coalesced 1 KiB per wave instruction)
template <class CT>
__device__ __forceinline__ void park(const CT &c, const f32x16 (&acc)[CT::NST][CT::FT], float *dst) {
#pragma unroll
    for (int st = 0; st < CT::NST; ++st)
#pragma unroll
        for (int ft = 0; ft < CT::FT; ++ft)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[st][ft][4 * m + r];
                const int idx4 = ((c.wave * CT::NST + st) * CT::FT + ft) * 4 + m;
                *reinterpret_cast<f32x4 *>(dst + ((size_t)idx4 * 64 + c.lane) * 4) = v;
            }
}
template <class CT>
__device__ __forceinline__ void unpark(const CT &c, f32x16 (&acc)[CT::NST][CT::FT], const float *src) {
#pragma unroll
    for (int st = 0; st < CT::NST; ++st)
#pragma unroll
        for (int ft = 0; ft < CT::FT; ++ft)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int idx4 = ((c.wave * CT::NST + st) * CT::FT + ft) * 4 + m;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(src + ((size_t)idx4 * 64 + c.lane) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[st][ft][4 * m + r] = v[r];
            }
}

// values in register order -> operand form in the LDS tile (hi / lo planes), scaled by ACT_SCALE
template <class CT>
__device__ __forceinline__ void regs_to_tile(const CT &c, const f32x16 (&y)[CT::NST][CT::FT], const float scale = ACT_SCALE) {
    const int j = c.lane & 31, hh = c.lane >> 5;
    if constexpr (CT::ARITH == 1) {
#pragma unroll
        for (int st = 0; st < CT::NST; ++st) {
            float *fp = c.f32() + (32 * st + j) * CT::RSF() + 32 * CT::FT * c.wave + 4 * hh;
#pragma unroll
            for (int ft = 0; ft < CT::FT; ++ft)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = y[st][ft][4 * m + r];
                    *reinterpret_cast<f32x4 *>(fp + 32 * ft + 8 * m) = v;
                }
        }
        return;
    }
#pragma unroll
    for (int st = 0; st < CT::NST; ++st) {
        _Float16 *hp = c.act + (32 * st + j) * c.RSH + 32 * CT::FT * c.wave + 4 * hh;
#pragma unroll
        for (int ft = 0; ft < CT::FT; ++ft)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = y[st][ft][4 * m + r];
                f16x4 hi, lo;
                split4(v, hi, lo, scale);
                *reinterpret_cast<f16x4 *>(hp + 32 * ft + 8 * m) = hi;
                *reinterpret_cast<f16x4 *>(hp + c.SH + 32 * ft + 8 * m) = lo;
            }
    }
}

// LayerNorm affine parameters travel through LDS.  A lane needs the values of its 16 features per feature tile, the
// same for all 32 sample columns: read from global that is 16 broadcast `global_load_dwordx4` per wave and layer, each
// costing the L1 its full 1 KiB / 16 cycles of address processing behind a weight stream that leaves nothing cached
// (measured: 62 k of 1 150 k cycles per launch).  Instead every thread fetches ONE value of the NEXT epilogue's g and b
// at the top of the current epilogue (gb_prefetch: latency hidden behind the epilogue), stores it after the barrier
// that ends the epilogue (gb_commit: every wave is past its reads of the previous values), and the next epilogue reads
// float4s from LDS after its own statistics barrier.  Protocol: every epi_t names its successor's parameters and is
// followed by epi_barrier().
struct GB {  // LayerNorm weight / bias pointers of one layer, by value (no address of a kernel argument is taken)
    const float *g = nullptr, *b = nullptr;
};
__device__ __forceinline__ GB gb_of(const LayerS &ly) { return GB{ly.g, ly.b}; }

template <class CT>
__device__ __forceinline__ void gb_prefetch(const CT &c, const float *g, const float *b) {
#pragma unroll
    for (int u = 0; u < (WIDTH + CT::NTHR - 1) / CT::NTHR; ++u) {
        const int f = c.tid + u * CT::NTHR;
        c.pg[u] = f < WIDTH ? g[f] : 0.f;
        c.pb[u] = f < WIDTH ? b[f] : 0.f;
    }
}
template <class CT>
__device__ __forceinline__ void gb_commit(const CT &c) {
#pragma unroll
    for (int u = 0; u < (WIDTH + CT::NTHR - 1) / CT::NTHR; ++u) {
        const int f = c.tid + u * CT::NTHR;
        if (f < WIDTH) {
            c.gb[f] = c.pg[u];
            c.gb[WIDTH + f] = c.pb[u];
        }
    }
}
template <class CT>
__device__ __forceinline__ void epi_barrier(const CT &c) {
    __syncthreads();
    gb_commit(c);
}

// acc (raw MFMA sums) -> ACT(LayerNorm(acc * osc + bias)) -> operand form in the LDS tile (+ optional register-order
// fp32 copy `zcopy` in global).  The bias values of this lane's features are read from global as float4 at
// 32 FT wave + 32 ft + 8 m + 4 hh (two distinct addresses per wave instruction), the LayerNorm affine values from
// c.gb at the same offsets; `next` = the layer whose epilogue follows this one in program order (null: none).  One barrier inside (the statistics exchange, which also orders every wave's last read of the operand tile
// before the first write of the new one); the caller adds the one before the next contraction.
template <int ACT, class CT, bool HASBIAS = true>
__device__ __forceinline__ void epi_t(const CT &c, f32x16 (&acc)[CT::NST][CT::FT], float osc, float asc, const float *bias,
                                      GB next, float *zcopy) {
    constexpr int FT = CT::FT, NW = CT::NWAVES;
    constexpr float CNT = 32.f * FT;  // features of a row held by one wave
    const int j = c.lane & 31, hh = c.lane >> 5;
    const int poff = 32 * FT * c.wave + 4 * hh;
    if constexpr (HASBIAS) {  // else: the caller applied scale and bias already (add_row_bias)
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
            f32x4 b4[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) b4[m] = *reinterpret_cast<const f32x4 *>(bias + poff + 32 * ft + 8 * m);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int st = 0; st < CT::NST; ++st)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[st][ft][4 * m + r] = fmaf(acc[st][ft][4 * m + r], osc, b4[m][r]);
        }
    }
    TIMER_MARK(c, T_EPI_BIAS)
    // per-wave partial statistics of the sample rows this lane works on
#pragma unroll
    for (int st = 0; st < CT::NST; ++st) {
        float s = 0.f;
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[st][ft][e];
        s += __shfl_xor(s, 32);
        const float mw = s * (1.0f / CNT);
        float m2 = 0.f;
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d = acc[st][ft][e] - mw;
                m2 = fmaf(d, d, m2);
            }
        m2 += __shfl_xor(m2, 32);
        if (hh == 0) {
            c.stats[(c.wave * CT::TROWS + 32 * st + j) * 2 + 0] = mw;
            c.stats[(c.wave * CT::TROWS + 32 * st + j) * 2 + 1] = m2;
        }
    }
    TIMER_MARK(c, T_EPI_PRE)
    __syncthreads();
    TIMER_MARK(c, T_EPI_SYNC)
    float rstd[CT::NST], shift[CT::NST];
#pragma unroll
    for (int st = 0; st < CT::NST; ++st) {
        float pm[NW], mean = 0.f, msum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            pm[w] = c.stats[(w * CT::TROWS + 32 * st + j) * 2 + 0];
            msum += c.stats[(w * CT::TROWS + 32 * st + j) * 2 + 1];
            mean += pm[w];
        }
        mean *= 1.0f / NW;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float d = pm[w] - mean;
            msum = fmaf(CNT * d, d, msum);  // Chan: M2 = sum M2_w + sum n_w (mean_w - mean)^2
        }
        rstd[st] = 1.0f / sqrtf(msum * (1.0f / WIDTH) + LN_EPS);
        shift[st] = -mean * rstd[st];
    }
    TIMER_MARK(c, T_EPI_COMB)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
        f32x4 gq[4], bq[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            gq[m] = *reinterpret_cast<const f32x4 *>(c.gb + poff + 32 * ft + 8 * m);
            bq[m] = *reinterpret_cast<const f32x4 *>(c.gb + WIDTH + poff + 32 * ft + 8 * m);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 g4 = gq[m], b4 = bq[m];
#pragma unroll
            for (int st = 0; st < CT::NST; ++st) {
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = fmaf(fmaf(acc[st][ft][4 * m + r], rstd[st], shift[st]), g4[r], b4[r]);
                if (ACT == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = mish_a<CT>(y[r]);
                } else {
                    float mx = fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3]));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    float es = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        y[r] = exp_a<CT>(y[r] - mx);
                        es += y[r];
                    }
                    es += __shfl_xor(es, 32);
                    if constexpr (CT::ARITH == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[r] = y[r] / es;
                    } else {
                        const float inv = 1.0f / es;
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[r] *= inv;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[st][ft][4 * m + r] = y[r];
            }
        }
    }
    regs_to_tile(c, acc, ACT == 0 ? asc : ACT_SCALE);  // SimNorm outputs lie in [0, 1]: fixed scale
    TIMER_MARK(c, T_EPI_MATH)  // LayerNorm affine + activation + split + tile store (the compiler merges them)
    if (zcopy) park(c, acc, zcopy);
}

// acc * osc + bias of the lane's OWN sample rows: brow[st] = the first-layer bias vector of row 32 st + (lane & 31)
// (training batches carry one task per row: world_model.py:95-97).  16-byte loads, two rows per lane.
template <class CT>
__device__ __forceinline__ void add_row_bias(const CT &c, f32x16 (&acc)[CT::NST][CT::FT], float osc, const float *const (&brow)[CT::NST]) {
    const int hh = c.lane >> 5;
    const int poff = 32 * CT::FT * c.wave + 4 * hh;
#pragma unroll
    for (int st = 0; st < CT::NST; ++st)
#pragma unroll
        for (int ft = 0; ft < CT::FT; ++ft)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(brow[st] + poff + 32 * ft + 8 * m);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[st][ft][4 * m + r] = fmaf(acc[st][ft][4 * m + r], osc, b4[r]);
            }
}

// Narrow output layers (two-hot / policy heads): acc * oscale + bias -> fp32 logits in the staging view of the tile.
// C/D fragment of kloop_tile_s (activations as the A operand): lane holds column (l & 31), rows (reg&3) + 8 (reg>>2) + 4 (l>>5).
template <class CT>
__device__ __forceinline__ void store_tile_s(const CT &c, const f32x16 &acc, float osc, const float *bias, int ct, int rt) {
    float *f = c.f32();
    constexpr int RSF = CT::RSF();
    const int j = c.lane & 31, hh = c.lane >> 5;
    const int col = ct * 32 + j;
    const float bv = bias[col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        f[row * RSF + col] = fmaf(acc[reg], osc, bv);
    }
}

// two_hot_inv (math.py:74-83) on fp32 logits in the staging view.
// softmax(logits) . bins -> symexp; the softmax normalisation is applied once to the weighted sum.
template <class CT>
__device__ __forceinline__ float twohot_rows_s(const CT &c, const float *bins, int num_bins) {
    const int part = c.tid & 7;
    const bool live = (c.tid >> 3) < CT::TROWS;  // ST = 1: the upper half of the workgroup has no row
    const int row = live ? c.tid >> 3 : 0;
    const float *rp = c.f32() + row * c.RSF();
    if (num_bins <= 1) {  // regression head (one output column): two_hot_inv is the identity (0) or symexp (1), math.py:76-79
        const float x1 = rp[0];
        return num_bins == 0 ? x1 : symexp_f(x1);
    }
    float v[16];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int jj = part + 8 * q;
        v[q] = (jj < num_bins) ? rp[jj] : -INFINITY;
        m = fmaxf(m, v[q]);
    }
    m = group_max<8>(m);
    float es = 0.f, x = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int jj = part + 8 * q;
        const float ev = (jj < num_bins) ? expf(v[q] - m) : 0.f;  // libm-accurate in both arithmetics: feeds symexp
        es += ev;
        x = fmaf(ev, (jj < num_bins) ? bins[jj] : 0.f, x);
    }
    es = group_sum<8>(es);
    x = group_sum<8>(x);
    return symexp_f(x / es);
}

template <int ACT, class CT>
__device__ __forceinline__ void layer_full_s(const CT &c, const LayerS &ly, const float *bias, int kb0, int kb1,
                                             GB next, float *zcopy = nullptr) {
    f32x16 acc[CT::NST][CT::FT];
    zero_acc(acc);
    if (next.g) gb_prefetch(c, next.g, next.b);  // in flight behind the whole contraction
    kloop_s(c, ly, kb0, kb1, acc);
    TIMER_MARK(c, T_KLOOP)
    epi_t<ACT>(c, acc, *ly.oscale, *ly.ascale, bias, next, zcopy);
    epi_barrier(c);
    TIMER_MARK(c, T_EPI)
}

// logits of a narrow head (<= 4 column tiles: one wave per column tile, both row tiles) -> the staging view.  (The contraction
// split over wave pairs -- all 8 waves, half of K each -- was built and measured: -0.2 %, profiles/README.md r4e: the head GEMM is
// MFMA work on one wave per SIMD that two waves per SIMD do not finish sooner.)
template <class CT>
__device__ __forceinline__ void head_logits_s(const CT &c, const LayerS &ly) {
    f32x16 acc[CT::NST];
    const float osc = *ly.oscale;
    const int ct = c.wave;
    if (ct < ly.CT) kloop_tile_s(c, ly, ct, 0, CT::ZKB, acc);
    __syncthreads();
    TIMER_MARK(c, T_HEAD_K)
    if (ct < ly.CT) {
#pragma unroll
        for (int rt = 0; rt < CT::NST; ++rt) store_tile_s(c, acc[rt], osc, ly.bias, ct, rt);
    }
    __syncthreads();
    TIMER_MARK(c, T_HEAD_ST)
}

template <class CT>
__device__ __forceinline__ float head_twohot_s(const CT &c, const LayerS &ly, const float *bins, int num_bins) {
    head_logits_s(c, ly);
    const float r = twohot_rows_s(c, bins, num_bins);
    __syncthreads();
    return r;
}

// write one action value into the operand-form action columns of a row
template <class CT>
__device__ __forceinline__ void put_action(const CT &c, int row, int a, float v) {
    if constexpr (CT::ARITH == 1) {
        c.f32()[row * CT::RSF() + WIDTH + a] = v;
        return;
    }
    const float vs = v * ACT_SCALE;
    const _Float16 h = (_Float16)vs;
    _Float16 *rp = c.act + row * c.RSH + WIDTH + a;
    rp[0] = h;
    rp[c.SH] = (_Float16)(vs - (float)h);
}

template <class CT, typename EpsFn>
__device__ __forceinline__ void head_pi_rows_s(const CT &c, int A, int Apad, float lsmin, float lsdif, const float *mask_wg, EpsFn eps,
                                               float *gdst, int nvalid, float *tsc, const float *mask_tab = nullptr,
                                               const int *row_task = nullptr, int put_rows = CT::TROWS, bool agent_store = false);

// Policy prior output layer + squashed Gaussian sample (world_model.py:152-173); action -> operand-form action columns
// (zero for padded columns) and optionally gdst[row * A + a] for rows < nvalid.
template <class CT, typename EpsFn>
__device__ __forceinline__ void head_pi_s(const CT &c, const LayerS &ly, int A, int Apad, float lsmin, float lsdif,
                                          const float *mask_wg, EpsFn eps, float *gdst, int nvalid, float *tsc,
                                          const float *mask_tab = nullptr, const int *row_task = nullptr) {
    head_logits_s(c, ly);
    head_pi_rows_s(c, A, Apad, lsmin, lsdif, mask_wg, eps, gdst, nvalid, tsc, mask_tab, row_task, CT::TROWS, false);
}

// the policy head's logits (mean | log_std, staging view) -> squashed Gaussian sample -> operand-form action columns
template <class CT, typename EpsFn>
__device__ __forceinline__ void head_pi_rows_s(const CT &c, int A, int Apad, float lsmin, float lsdif, const float *mask_wg, EpsFn eps,
                                               float *gdst, int nvalid, float *tsc, const float *mask_tab, const int *row_task,
                                               int put_rows, bool agent_store) {
    // put_rows: only rows < put_rows take the action into their operand columns (the cluster path's in-launch policy prior:
    // the other rows of the tile keep their sampled actions); agent_store: gdst is read by another workgroup of this launch
    const int row = c.tid >> 3, part = c.tid & 7;
    const float *rp = c.f32() + row * c.RSF();
    // action mask: one per workgroup (planning: the plan's task) or one per row (training batches: mask_tab[task of row])
    const float *mask = mask_wg;
    if (mask_tab && row < CT::TROWS) mask = mask_tab + (size_t)row_task[row] * A;
    // the logits (staging columns < 2A <= 128) do not alias the action columns (hi: floats 256.., lo: floats >= 512)
    if (row < CT::TROWS)
    for (int a = part; a < Apad; a += 8) {
        float out = 0.f;
        if (a < A) {
            float mu = rp[a], lsr = rp[A + a];
            float ls = lsmin + 0.5f * lsdif * (tanhf(lsr) + 1.f);  // math.log_std, math.py:12-13
            float e = eps(row, a);
            if (mask) {
                const float mk = mask[a];
                mu *= mk;
                ls *= mk;
                e *= mk;
            }
            out = tanhf(mu + e * expf(ls));
            if (gdst && row < nvalid) {
                if (agent_store) __hip_atomic_store(gdst + row * A + a, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else gdst[row * A + a] = out;
            }
            if (tsc) tsc[a] = out;
        }
        if (row < put_rows) put_action(c, row, a, out);
    }
    __syncthreads();
}

// register-order fp32 tile in global (written by park / epi_t's zcopy) -> operand-form z columns
template <class CT>
__device__ __forceinline__ void tile_from_global_s(const CT &c, const float *src) {
    f32x16 y[CT::NST][CT::FT];
    unpark(c, y, src);
    regs_to_tile(c, y);
}
template <class CT>
__device__ __forceinline__ void tile_broadcast_row_s(const CT &c, const float *src_row) {
    for (int idx = c.tid; idx < CT::TROWS * (WIDTH / 4); idx += CT::NTHR) {
        const int row = idx / (WIDTH / 4), c4 = idx % (WIDTH / 4);
        const f32x4 y = *reinterpret_cast<const f32x4 *>(src_row + 4 * c4);
        if constexpr (CT::ARITH == 1) {
            *reinterpret_cast<f32x4 *>(c.f32() + row * CT::RSF() + 4 * c4) = y;
        } else {
            f16x4 hi, lo;
            split4(y, hi, lo);
            _Float16 *hp = c.act + row * c.RSH + 4 * c4;
            *reinterpret_cast<f16x4 *>(hp) = hi;
            *reinterpret_cast<f16x4 *>(hp + c.SH) = lo;
        }
    }
}
// row-major fp32 rows in global -> operand-form z columns (rows >= nvalid are zero)
template <class CT>
__device__ __forceinline__ void tile_from_rows_s(const CT &c, const float *src, int nvalid) {
    for (int idx = c.tid; idx < CT::TROWS * (WIDTH / 4); idx += CT::NTHR) {
        const int row = idx / (WIDTH / 4), c4 = idx % (WIDTH / 4);
        f32x4 y = {0.f, 0.f, 0.f, 0.f};
        if (row < nvalid) y = *reinterpret_cast<const f32x4 *>(src + (size_t)row * WIDTH + 4 * c4);
        if constexpr (CT::ARITH == 1) {
            *reinterpret_cast<f32x4 *>(c.f32() + row * CT::RSF() + 4 * c4) = y;
        } else {
            f16x4 hi, lo;
            split4(y, hi, lo);
            _Float16 *hp = c.act + row * c.RSH + 4 * c4;
            *reinterpret_cast<f16x4 *>(hp) = hi;
            *reinterpret_cast<f16x4 *>(hp + c.SH) = lo;
        }
    }
}
// operand form -> fp32 trace dump (hi + lo, unscaled)
template <class CT>
__device__ __forceinline__ void dump_tile_s(const CT &c, float *trace, int nslot, int slot, const float *scale_ptr = nullptr) {
    if (!trace) return;
    const float inv_scale = 1.0f / (scale_ptr ? *scale_ptr : ACT_SCALE);
    float *dst = trace + ((size_t)blockIdx.x * nslot + slot) * CT::TROWS * WIDTH;
    for (int idx = c.tid; idx < CT::TROWS * WIDTH; idx += CT::NTHR) {
        const int row = idx / WIDTH, col = idx % WIDTH;
        if constexpr (CT::ARITH == 1) {
            dst[idx] = c.f32()[row * CT::RSF() + col];
        } else {
            const _Float16 *hp = c.act + row * c.RSH + col;
            dst[idx] = ((float)hp[0] + (float)hp[c.SH]) * inv_scale;
        }
    }
}

// ================================================================ kernel: per-plan setup
template <int APAD, int AR>
__global__ __launch_bounds__(NTHREADS, 2) void ks_setup(SetupParamsT<NetS> p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int e = blockIdx.x, tid = threadIdx.x;
    typedef CtxT<APAD, 2, 8, AR> CT;  // 64 rows, 8 waves
    CT c{reinterpret_cast<_Float16 *>(smem), smem + ROWS * CT::RSF(), smem + ROWS * CT::RSF() + 1024, tid,
         __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63};
    if (p.cl_flags)  // cluster path (cluster_kernels.cuh): this plan's arrival words start at phase 0
        for (int idx = tid; idx < p.cl_flag_words; idx += NTHREADS) p.cl_flags[(size_t)e * p.cl_flag_words + idx] = 0u;
    if (p.cl2_flags && e == 0)
        for (int idx = tid; idx < p.cl2_flag_words; idx += NTHREADS) p.cl2_flags[idx] = 0u;
    if (p.err_clear && e == 0 && tid == 0) __hip_atomic_store(p.err_clear, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (p.multitask) {
        const float *emb = p.task_emb + (size_t)e * p.T;
        for (int net = 0; net < p.nnets; ++net) {
            const LayerS &l1 = net == BE_DYN ? p.dyn.l[0] : net == BE_REW ? p.rew.l[0] : net == BE_PI ? p.pi.l[0]
                                                                                                    : p.q[net - BE_Q0].l[0];
            const float *w = p.wemb[net] + (size_t)tid * p.T;
            float s = 0.f;
            for (int k = 0; k < p.T; ++k) s = fmaf(w[k], emb[k], s);
            p.beff[((size_t)e * p.nnets + net) * WIDTH + tid] = l1.bias[tid] + s;
        }
    }
    for (int idx = tid; idx < p.H * p.A; idx += NTHREADS) {
        const int t = idx / p.A;
        float m = 0.f;
        if (!p.t0[e] && t < p.H - 1) m = p.prev_mean[(size_t)e * p.H * p.A + idx + p.A];
        p.mean[(size_t)e * p.H * p.A + idx] = m;
        p.std[(size_t)e * p.H * p.A + idx] = p.max_std;
    }
    if (p.skip_cvec) return;  // cluster path: step 0 contracts the full [z | a] range itself
    tile_broadcast_row_s(c, p.z0 + (size_t)e * WIDTH);
    __syncthreads();
    f32x16 acc[2][2][2];
    zero_acc(acc[0]);
    zero_acc(acc[1]);
    kloop_s(c, p.rew.l[0], 0, CT::ZKB, acc[0]);
    kloop_s(c, p.dyn.l[0], 0, CT::ZKB, acc[1]);
    const float *b_rew = p.multitask ? p.beff + ((size_t)e * p.nnets + BE_REW) * WIDTH : p.rew.l[0].bias;
    const float *b_dyn = p.multitask ? p.beff + ((size_t)e * p.nnets + BE_DYN) * WIDTH : p.dyn.l[0].bias;
    const float o_rew = *p.rew.l[0].oscale, o_dyn = *p.dyn.l[0].oscale;
    if ((c.lane & 31) == 0) {  // sample row 0 of the tile: lanes 0 (hh = 0) and 32 (hh = 1), sample tile 0
        const int hh = c.lane >> 5;
#pragma unroll
        for (int ft = 0; ft < CT::FT; ++ft)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int col = 64 * c.wave + 32 * ft + 8 * (reg >> 2) + 4 * hh + (reg & 3);
                p.cvec[((size_t)e * 2 + 0) * WIDTH + col] = fmaf(acc[0][0][ft][reg], o_rew, b_rew[col]);
                p.cvec[((size_t)e * 2 + 1) * WIDTH + col] = fmaf(acc[1][0][ft][reg], o_dyn, b_dyn[col]);
            }
    }
}

// ================================================================ kernel: policy-prior trajectories
// ST = 1 (one 32-row tile) when num_pi_trajs <= 32 -- the reference's 24 -- else 2.
template <int APAD, int ST, int AR>
__global__ __launch_bounds__(NTHREADS, 2) void ks_pitraj(PiTrajParamsT<NetS> p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int e = blockIdx.x, tid = threadIdx.x;
    typedef CtxT<APAD, ST, 8, AR> CT;
    constexpr int ZKB16 = CT::ZKB;  // k-blocks of this arithmetic covering the latent columns
    CT c{reinterpret_cast<_Float16 *>(smem), smem + CT::TROWS * CT::RSF(), smem + CT::TROWS * CT::RSF() + 1024, tid,
         __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63};
    const float *mask = p.act_mask ? p.act_mask + (size_t)e * p.A : nullptr;
    const float *b_pi = p.multitask ? p.beff + ((size_t)e * p.nnets + BE_PI) * WIDTH : p.pi.l[0].bias;
    const float *b_dyn = p.multitask ? p.beff + ((size_t)e * p.nnets + BE_DYN) * WIDTH : p.dyn.l[0].bias;
    float *zs = p.zscratch + (size_t)e * p.zscratch_estride;
    const int KBA = ZKB16 + p.Apad / CT::KBLK;
    tile_broadcast_row_s(c, p.z0 + (size_t)e * WIDTH);
    {  // zs <- z0 for every sample row, in register order (what tile_from_global_s reads back)
        f32x16 y[ST][2];
        const int hh = c.lane >> 5;
#pragma unroll
        for (int ft = 0; ft < CT::FT; ++ft)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 z = *reinterpret_cast<const f32x4 *>(p.z0 + (size_t)e * WIDTH + 64 * c.wave + 32 * ft + 8 * m + 4 * hh);
#pragma unroll
                for (int st = 0; st < ST; ++st)
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[st][ft][4 * m + r] = z[r];
            }
        park(c, y, zs);
    }
    gb_prefetch(c, p.pi.l[0].g, p.pi.l[0].b);
    epi_barrier(c);
    for (int t = 0; t < p.H; ++t) {
        layer_full_s<0>(c, p.pi.l[0], b_pi, 0, ZKB16, gb_of(p.pi.l[1]));
        layer_full_s<0>(c, p.pi.l[1], p.pi.l[1].bias, 0, ZKB16, t == p.H - 1 ? GB{} : gb_of(p.dyn.l[0]));
        const float *tape = p.pi_traj_eps ? p.pi_traj_eps + ((size_t)e * p.H + t) * p.P * p.A : nullptr;
        auto eps = [&](int row, int a) -> float {
            if (row >= p.P) return 0.f;
            if (tape) return tape[row * p.A + a];
            return rng_normal(p.seed, p.call, SITE_PITRAJ, t, e, (unsigned)(row * p.A + a));
        };
        head_pi_s(c, p.pi.l[2], p.A, p.Apad, p.log_std_min, p.log_std_dif, mask, eps,
                  p.actions + ((size_t)e * p.H + t) * p.N * p.A, p.P, nullptr);
        if (t == p.H - 1) break;
        tile_from_global_s(c, zs);
        __syncthreads();
        layer_full_s<0>(c, p.dyn.l[0], b_dyn, 0, KBA, gb_of(p.dyn.l[1]));
        layer_full_s<0>(c, p.dyn.l[1], p.dyn.l[1].bias, 0, ZKB16, gb_of(p.dyn.l[2]));
        layer_full_s<1>(c, p.dyn.l[2], p.dyn.l[2].bias, 0, ZKB16, gb_of(p.pi.l[0]), zs);
    }
}

// Termination head output (world_model.py:132-141; tdmpc2.py:133-134): 1 logit per row -> (sigmoid(logit) > 0.5) in
// every lane of the row's 8-lane group.  Column tile 0 only: one wave runs the contraction.
template <class CT>
__device__ __forceinline__ float head_term_s(const CT &c, const LayerS &ly) {
    f32x16 acc[CT::NST];
    if (c.wave == 0) kloop_tile_s(c, ly, 0, 0, CT::ZKB, acc);
    const float osc = *ly.oscale;
    __syncthreads();
    if (c.wave == 0) {
#pragma unroll
        for (int rt = 0; rt < CT::NST; ++rt) store_tile_s(c, acc[rt], osc, ly.bias, 0, rt);
    }
    __syncthreads();
    const bool live = (c.tid >> 3) < CT::TROWS;
    const float x = c.f32()[(live ? c.tid >> 3 : 0) * c.RSF()];
    const float pr = 1.f / (1.f + expf(-x));
    __syncthreads();
    return pr > 0.5f ? 1.f : 0.f;
}

// (Round 2 contracted the two first layers over the same [z | a] tile in ONE compiler-scheduled pass -- fragments read from LDS
// once, +1.5 ... 2.6 % --; the hand-ordered loop (kloop_asm) run twice beats it: profiles/README.md r02c and "What was tried".)
// first layers of two nets over the same tile: acca <- la, accb <- lb (raw sums)
template <class CT>
__device__ __forceinline__ void first_layers_s(const CT &c, const LayerS &la, const LayerS &lb, int kb0, int kb1,
                                               f32x16 (&acca)[CT::NST][CT::FT], f32x16 (&accb)[CT::NST][CT::FT]) {
    zero_acc(acca);
    zero_acc(accb);
    kloop_s(c, la, kb0, kb1, acca);
    kloop_s(c, lb, kb0, kb1, accb);
}

// ================================================================ kernel: one CEM iteration's rollouts
// EP = 1: episodic planning -- the termination head (world_model.py:132-141) is evaluated on every new latent and masks
// the later rewards and the terminal value (tdmpc2.py:128-136).  Its first layer reads the same z_t columns as the step's
// reward / dynamics first layers, so it runs at the top of step t (t >= 1; after the loop next to the policy prior's):
// the raw dynamics sums wait in the workgroup's L2 scratch tile, the raw termination sums in the held registers.
// TR = 1: the instantiation behind tdmpc2_plan_estimate_value_trace (activation dumps after every phase); the planning
// instantiations (TR = 0) carry no trace code -- 66 fewer spilled SGPRs, 7 fewer VGPRs, +0.6 % (A/B r03h).  Episodic kernels
// keep the dumps inside the one instantiation (TRACE = TR || EP).
#define DUMP_TILE(...) do { if constexpr (TRACE) dump_tile_s(__VA_ARGS__); } while (0)
// (Registers: the 8-wave instantiations fill the 256-VGPR budget of two waves per SIMD -- 65 spilled, none inside a k-loop or
// an epilogue; dropping the second launch-bounds argument changes nothing: tools/kmeta.py, profiles/README.md r4w.)
template <int APAD, int ST, int NW, int AR, int EP, int TR = 0>
__global__ __launch_bounds__(64 * NW, 2) void ks_rollout(RolloutParamsT<NetS> p) {
    constexpr bool TRACE = TR || EP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int s_is_last;
    const int e = blockIdx.x / p.tiles, tile = p.tile_off + blockIdx.x % p.tiles;  // tile_off: a row range of the plan (shard_values)
    const int tid = threadIdx.x;
    typedef CtxT<APAD, ST, NW, AR> CT;
    constexpr int TROWS = CT::TROWS, NTHR = CT::NTHR, FT = CT::FT;
    constexpr int ZKB16 = CT::ZKB;  // k-blocks of this arithmetic covering the latent columns
    CT c{reinterpret_cast<_Float16 *>(smem), smem + TROWS * CT::RSF(), smem + TROWS * CT::RSF() + 1024, tid,
         __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63};
    float *sm_mean = smem + TROWS * CT::RSF() + 2048;  // [H*A] after the tile, the LayerNorm partials and c.gb
    float *sm_std = sm_mean + p.H * p.A;
    const int row0 = tile * TROWS;
    const float *mask = p.act_mask ? p.act_mask + (size_t)e * p.A : nullptr;
    const float *disc = p.disc_pow + (size_t)e * (p.H + 1);
    const int KBA = ZKB16 + p.Apad / CT::KBLK;
    float *zs = p.zscratch + (size_t)blockIdx.x * TROWS * WIDTH;
    const int NSLOT = 5 * p.H + 7;
    const bool live = (tid >> 3) < TROWS;  // ST = 1: threads 256..511 own no sample row in the row-per-8-lanes phases
    float *tsc = nullptr;
    if constexpr (TRACE) tsc = p.trace_scalars && live ? p.trace_scalars + ((size_t)e * p.N + row0 + (tid >> 3)) * (p.H + 2 + p.A) : nullptr;

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
    gb_prefetch(c, p.rew.l[0].g, p.rew.l[0].b);  // the first epilogue's LayerNorm parameters
    epi_barrier(c);

    float G = 0.f;
    float termv = 0.f;  // EP: 1 once any latent of this row's trajectory was classified terminal (tdmpc2.py:133-134)
    TIMER_START(c)
    for (int t = 0; t < p.H; ++t) {
        // ---- actions of step t (tdmpc2.py:176-181) -> operand-form action columns; a thread handles PAIRS of action
        // columns so that one Philox call feeds two samples
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
                            v[u] = ag[(size_t)n * p.A + a];
                        } else {
                            float r = z[u];
                            if (p.sample_eps)
                                r = p.sample_eps[(size_t)e * p.sample_eps_estride +
                                                 (unsigned)(((size_t)t * (p.N - p.P) + (n - p.P)) * p.A + a)];
                            v[u] = sample_action(sm_mean[t * p.A + a], sm_std[t * p.A + a], r);
                        }
                        if (mask && !p.given_actions) v[u] *= mask[a];
                        if (!p.given_actions) ag[(size_t)n * p.A + a] = v[u];
                    }
                    put_action(c, row, a, v[u]);
                }
            }
        }
        __syncthreads();
        TIMER_MARK(c, T_ACT)
        // ---- first layers of dynamics and reward over the same [z_t | a_t] tile; the raw dynamics accumulators wait for
        // the reward chain in 64 held VGPRs (parking them in L2 instead measured 4.6 % slower)
        const bool term_step = EP && t > 0;
        f32x16 accd[ST][FT];
        {
            f32x16 acc[ST][FT];
            gb_prefetch(c, p.rew.l[1].g, p.rew.l[1].b);
            first_layers_s(c, p.dyn.l[0], p.rew.l[0], t == 0 ? ZKB16 : 0, KBA, accd, acc);
            TIMER_MARK(c, T_KLOOP)
            if constexpr (EP) {
                if (term_step) {  // the held registers now take the termination head's first layer on z_t
                    park(c, accd, zs);
                    zero_acc(accd);
                    kloop_s(c, p.term.l[0], 0, ZKB16, accd);
                }
            }
            epi_t<0>(c, acc, *p.rew.l[0].oscale, *p.rew.l[0].ascale, t == 0 ? p.cvec + ((size_t)e * 2 + 0) * WIDTH : b_rew,
                     gb_of(p.rew.l[1]), nullptr);
        }
        epi_barrier(c);
        TIMER_MARK(c, T_EPI)
        DUMP_TILE(c, p.trace_tiles, NSLOT, 5 * t + 0, p.rew.l[0].ascale);
        // ---- reward: layer 2, two-hot head
        layer_full_s<0>(c, p.rew.l[1], p.rew.l[1].bias, 0, ZKB16, term_step ? gb_of(p.term.l[0]) : gb_of(p.dyn.l[0]));
        DUMP_TILE(c, p.trace_tiles, NSLOT, 5 * t + 1, p.rew.l[1].ascale);
        // for the held epilogue below; in flight behind the head
        if (term_step) gb_prefetch(c, p.term.l[1].g, p.term.l[1].b);
        else gb_prefetch(c, p.dyn.l[1].g, p.dyn.l[1].b);
        const float r = head_twohot_s(c, p.rew.l[2], p.bins, p.num_bins);
        TIMER_MARK(c, T_HEAD)
        if (tsc && (tid & 7) == 0) tsc[t] = r;
        if constexpr (EP) {
            if (term_step) {  // termination(z_t): layers 1 (held), 2, logit
                epi_t<0>(c, accd, *p.term.l[0].oscale, *p.term.l[0].ascale, p.term.l[0].bias, gb_of(p.term.l[1]), nullptr);
                epi_barrier(c);
                layer_full_s<0>(c, p.term.l[1], p.term.l[1].bias, 0, ZKB16, gb_of(p.dyn.l[0]));
                gb_prefetch(c, p.dyn.l[1].g, p.dyn.l[1].b);
                termv = fminf(termv + head_term_s(c, p.term.l[2]), 1.f);
                unpark(c, accd, zs);
            }
        }
        G += disc[t] * (1.f - termv) * r;
        // ---- dynamics: release the held first layer, layers 2 and 3 (SimNorm)
        epi_t<0>(c, accd, *p.dyn.l[0].oscale, *p.dyn.l[0].ascale, t == 0 ? p.cvec + ((size_t)e * 2 + 1) * WIDTH : b_dyn,
                 gb_of(p.dyn.l[1]), nullptr);
        epi_barrier(c);
        TIMER_MARK(c, T_EPI)
        DUMP_TILE(c, p.trace_tiles, NSLOT, 5 * t + 2, p.dyn.l[0].ascale);
        layer_full_s<0>(c, p.dyn.l[1], p.dyn.l[1].bias, 0, ZKB16, gb_of(p.dyn.l[2]));
        DUMP_TILE(c, p.trace_tiles, NSLOT, 5 * t + 3, p.dyn.l[1].ascale);
        layer_full_s<1>(c, p.dyn.l[2], p.dyn.l[2].bias, 0, ZKB16, t == p.H - 1 ? gb_of(p.pi.l[0]) : gb_of(p.rew.l[0]),
                        t == p.H - 1 ? zs : nullptr);
        DUMP_TILE(c, p.trace_tiles, NSLOT, 5 * t + 4);
    }
    // ---- a_H = pi(z_H) (tdmpc2.py:135); z_H was also saved to zs.  EP: termination(z_H) first layer from the same tile.
    f32x16 acch[ST][FT];  // held raw sums: termination first layer (EP), then the second Q head's
    if constexpr (EP) {
        zero_acc(acch);
        kloop_s(c, p.term.l[0], 0, ZKB16, acch);
    }
    layer_full_s<0>(c, p.pi.l[0], b_pi, 0, ZKB16, gb_of(p.pi.l[1]));
    DUMP_TILE(c, p.trace_tiles, NSLOT, 5 * p.H + 0, p.pi.l[0].ascale);
    layer_full_s<0>(c, p.pi.l[1], p.pi.l[1].bias, 0, ZKB16, EP ? gb_of(p.term.l[0]) : gb_of(p.q[q0].l[0]));
    DUMP_TILE(c, p.trace_tiles, NSLOT, 5 * p.H + 1, p.pi.l[1].ascale);
    if constexpr (EP) gb_prefetch(c, p.term.l[1].g, p.term.l[1].b);
    {
        auto eps = [&](int row, int a) -> float {
            const unsigned ridx = (unsigned)((size_t)(row0 + row) * p.A + a);
            if (p.pi_eps) return p.pi_eps[(size_t)e * p.pi_eps_estride + ridx];
            return rng_normal(p.seed, p.call, SITE_PI, p.iter, e, ridx);
        };
        head_pi_s(c, p.pi.l[2], p.A, p.Apad, p.log_std_min, p.log_std_dif, mask, eps, nullptr, 0,
                  tsc ? tsc + p.H + 2 : nullptr);
    }
    TIMER_MARK(c, T_HEAD)
    if constexpr (EP) {  // termination(z_H) (tdmpc2.py:133-134, last loop iteration): the hidden layers overwrite z columns only
        epi_t<0>(c, acch, *p.term.l[0].oscale, *p.term.l[0].ascale, p.term.l[0].bias, gb_of(p.term.l[1]), nullptr);
        epi_barrier(c);
        layer_full_s<0>(c, p.term.l[1], p.term.l[1].bias, 0, ZKB16, gb_of(p.q[q0].l[0]));
        termv = fminf(termv + head_term_s(c, p.term.l[2]), 1.f);
    }
    tile_from_global_s(c, zs);
    __syncthreads();
    TIMER_MARK(c, T_TILE)
    DUMP_TILE(c, p.trace_tiles, NSLOT, 5 * p.H + 2);
    // ---- Q(z_H, a_H): the two selected heads, first layers in one pass (world_model.py:186-216)
    {
        f32x16 acc[ST][FT];
        gb_prefetch(c, p.q[q0].l[1].g, p.q[q0].l[1].b);
        first_layers_s(c, p.q[q1].l[0], p.q[q0].l[0], 0, KBA, acch, acc);
        TIMER_MARK(c, T_KLOOP)
        epi_t<0>(c, acc, *p.q[q0].l[0].oscale, *p.q[q0].l[0].ascale, b_q0, gb_of(p.q[q0].l[1]), nullptr);
    }
    epi_barrier(c);
    TIMER_MARK(c, T_EPI)
    DUMP_TILE(c, p.trace_tiles, NSLOT, 5 * p.H + 3, p.q[q0].l[0].ascale);
    layer_full_s<0>(c, p.q[q0].l[1], p.q[q0].l[1].bias, 0, ZKB16, gb_of(p.q[q1].l[0]));
    DUMP_TILE(c, p.trace_tiles, NSLOT, 5 * p.H + 4, p.q[q0].l[1].ascale);
    gb_prefetch(c, p.q[q1].l[1].g, p.q[q1].l[1].b);  // for the held second-head epilogue below
    const float qa = head_twohot_s(c, p.q[q0].l[2], p.bins, p.num_bins);
    TIMER_MARK(c, T_HEAD)
    epi_t<0>(c, acch, *p.q[q1].l[0].oscale, *p.q[q1].l[0].ascale, b_q1, gb_of(p.q[q1].l[1]), nullptr);
    epi_barrier(c);
    TIMER_MARK(c, T_EPI)
    DUMP_TILE(c, p.trace_tiles, NSLOT, 5 * p.H + 5, p.q[q1].l[0].ascale);
    layer_full_s<0>(c, p.q[q1].l[1], p.q[q1].l[1].bias, 0, ZKB16, GB{});
    DUMP_TILE(c, p.trace_tiles, NSLOT, 5 * p.H + 6, p.q[q1].l[1].ascale);
    const float qb = head_twohot_s(c, p.q[q1].l[2], p.bins, p.num_bins);
    TIMER_MARK(c, T_HEAD)
    TIMER_FLUSH(c, p.timing)
    if (tsc && (tid & 7) == 0) {
        tsc[p.H] = qa;
        tsc[p.H + 1] = qb;
    }
    const float val = G + disc[p.H] * (1.f - termv) * ((qa + qb) / 2.f);
    if (!p.fold_refit) {
        if ((tid & 7) == 0 && live) p.value[(size_t)e * p.N + row0 + (tid >> 3)] = val;
        return;
    }
    // ---- elite selection + refit by the LAST workgroup of this plan to get here (tdmpc2.py:184-206): one launch per CEM
    // iteration.  Hand-over protocol (MI355X_MICROARCH.md, inter-workgroup visibility; a plan's workgroups sit on different
    // XCDs, whose L2s are not coherent): the only data another workgroup reads -- this workgroup's 32 / 64 values --
    // leaves as agent-scope write-through (sc1) stores; every wave drains its stores (an explicit s_waitcnt vmcnt(0): the
    // workgroup-scope release fence alone emits no wait) before the barrier; one lane takes the plan's ticket with an agent-scope atomic; the last
    // arriver reads the values with agent-scope (sc1) loads, which bypass its L1 and find lines its XCD's L2 has not held
    // in this launch (refit_plan).  No cache maintenance instruction on either side.  The elite ACTIONS are not handed
    // over at all: the refit re-derives them (RefitParams::regen).  Measured and rejected: agent-scope release / acquire
    // fences (buffer_wbl2 / buffer_inv sc1 act on the XCD's whole L2 -- the acquire alone evicts the L2-resident weights of
    // 32 neighbouring workgroups: +13 % on the launch, profiles/README.md r02b-r02d).
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
        if (last) __hip_atomic_store(p.ticket + e, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch starts at 0
        s_is_last = last;
    }
    __syncthreads();
    if (!s_is_last) return;
    refit_plan(p.rf, e, smem, tid, NTHR);
}

// ================================================================ kernel: pi + two Q heads on a batch of latent rows
// The forward halves of TDMPC2._td_target (tdmpc2/tdmpc2.py:239-254: a = pi(z'), min of two target heads, then
// r + discount (1 - terminated) Q) and of TDMPC2.update_pi (tdmpc2.py:208-225: a = pi(z), mean of two online heads) on the
// planner's layer code: one workgroup per 64 rows, the same three chains the rollout kernel ends with.  Multitask
// batches carry one task per row (world_model.py:95-97): the first-layer biases b + W[:, L:L+T] . task_emb come from a
// per-task table (`beff_tab`, built by ks_task_bias) indexed with the row's task, and so do the action mask and discount.
template <int APAD, int AR>
__global__ __launch_bounds__(NTHREADS, 2) void ks_value(ValueParamsT<NetS> p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    typedef CtxT<APAD, 2, 8, AR> CT;
    constexpr int TROWS = CT::TROWS, ZKB16 = CT::ZKB;
    CT c{reinterpret_cast<_Float16 *>(smem), smem + TROWS * CT::RSF(), smem + TROWS * CT::RSF() + 1024, tid,
         __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63};
    int *s_task = reinterpret_cast<int *>(smem + TROWS * CT::RSF() + 2048);  // [TROWS] task of each row (multitask)
    const int row0 = blockIdx.x * TROWS;
    const int nvalid = min(TROWS, p.rows - row0);
    const float *zsrc = p.z + (size_t)row0 * WIDTH;
    const int KBA = ZKB16 + p.Apad / CT::KBLK;
    int q0, q1;
    if (p.qidx) {
        q0 = p.qidx[0];
        q1 = p.qidx[1];
    } else {  // randperm(num_q)[:2] (world_model.py:212), one draw per call
        const uint4 r = rng_raw(p.seed, p.call, SITE_QIDX, 0, 0, 0);
        q0 = (int)(r.x % (unsigned)p.nq);
        q1 = (int)(r.y % (unsigned)(p.nq - 1));
        if (q1 >= q0) ++q1;
    }
    if (p.task_ids && tid < TROWS) s_task[tid] = tid < nvalid ? p.task_ids[row0 + tid] : 0;
    tile_from_rows_s(c, zsrc, nvalid);
    gb_prefetch(c, p.pi.l[0].g, p.pi.l[0].b);
    epi_barrier(c);
    // first-layer bias vectors of this lane's two sample rows
    const int j = c.lane & 31;
    auto first_layer = [&](const LayerS &ly, int slot, int kb1, GB next) {
        if (!p.task_ids) {
            layer_full_s<0>(c, ly, ly.bias, 0, kb1, next);
            return;
        }
        const float *brow[CT::NST];
#pragma unroll
        for (int st = 0; st < CT::NST; ++st) brow[st] = p.beff_tab + ((size_t)s_task[32 * st + j] * p.nnets + slot) * WIDTH;
        f32x16 acc[CT::NST][CT::FT];
        zero_acc(acc);
        if (next.g) gb_prefetch(c, next.g, next.b);
        kloop_s(c, ly, 0, kb1, acc);
        add_row_bias(c, acc, *ly.oscale, brow);
        epi_t<0, CT, false>(c, acc, 1.f, *ly.ascale, nullptr, next, nullptr);
        epi_barrier(c);
    };
    first_layer(p.pi.l[0], BE_PI, ZKB16, gb_of(p.pi.l[1]));
    layer_full_s<0>(c, p.pi.l[1], p.pi.l[1].bias, 0, ZKB16, gb_of(p.q[q0].l[0]));
    {
        auto eps = [&](int row, int a) -> float {
            const unsigned ridx = (unsigned)((size_t)(row0 + row) * p.A + a);
            if (p.pi_eps) return row < nvalid ? p.pi_eps[ridx] : 0.f;
            return rng_normal(p.seed, p.call, SITE_PI, 0, 0, ridx);
        };
        head_pi_s(c, p.pi.l[2], p.A, p.Apad, p.log_std_min, p.log_std_dif, nullptr, eps,
                  p.action ? p.action + (size_t)row0 * p.A : nullptr, nvalid, nullptr, p.task_ids ? p.mask_tab : nullptr, s_task);
    }
    tile_from_rows_s(c, zsrc, nvalid);  // the hidden layers overwrote the z columns; the action columns stay
    __syncthreads();
    first_layer(p.q[q0].l[0], BE_Q0 + q0, KBA, gb_of(p.q[q0].l[1]));
    layer_full_s<0>(c, p.q[q0].l[1], p.q[q0].l[1].bias, 0, ZKB16, gb_of(p.q[q1].l[0]));
    const float qa = head_twohot_s(c, p.q[q0].l[2], p.bins, p.num_bins);
    tile_from_rows_s(c, zsrc, nvalid);
    __syncthreads();
    first_layer(p.q[q1].l[0], BE_Q0 + q1, KBA, gb_of(p.q[q1].l[1]));
    layer_full_s<0>(c, p.q[q1].l[1], p.q[q1].l[1].bias, 0, ZKB16, GB{});
    const float qb = head_twohot_s(c, p.q[q1].l[2], p.bins, p.num_bins);
    const int row = tid >> 3;
    if ((tid & 7) == 0 && row < nvalid) {
        float v = p.reduce_min ? fminf(qa, qb) : (qa + qb) / 2.f;
        if (p.reward) {
            const float disc = p.disc_tab ? p.disc_tab[s_task[row]] : p.discount;
            v = p.reward[row0 + row] + disc * (1.f - p.terminated[row0 + row]) * v;
        }
        p.out[row0 + row] = v;
    }
}

