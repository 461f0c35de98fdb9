// The schedule arithmetic of the fused family's hand-ordered contraction loop (fused_kernels.cuh: kloop_asm) -- which k-block a
// step waits for, how many of the wave's own weight loads it may leave in flight, whether it reads / requests a further block --
// in one place: the kernel calls these, and tests/test_ring_schedule.py compiles them with g++ and replays the loop for every
// contraction length (no fragment used before its loads have landed, no ring slot reloaded before its MFMAs have issued, nothing
// in flight at the end, the counted waits tight).  Plain C++ (no HIP types).
#pragma once

#ifndef __HIPCC__
#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
#endif

constexpr int KL_RD = 2;        // weight ring depth in k-blocks (16 VGPRs each); 4 lost 3.3 % (profiles/r4w_kloop_ring4_ab.txt)
constexpr int KL_W_LOADS = 4;   // global_load_dwordx4 per k-block and wave: hi / lo plane of two column tiles

// a whole trip of `rd` steady steps from block k on: every step has a block k' + rd <= nk - 1 to request
__host__ __device__ inline bool kl_steady_trip(int k, int nk, int rd) { return k + 2 * rd - 1 < nk; }
__host__ __device__ inline int kl_steady_vmcnt(int rd) { return KL_W_LOADS * (rd - 1); }
// last steps: blocks behind block kk that are still in flight at its wait (issued and not yet consumed)
__host__ __device__ inline int kl_behind(int kk, int nk, int rd) { return nk - 1 - kk < rd - 1 ? nk - 1 - kk : rd - 1; }
__host__ __device__ inline bool kl_next_act(int kk, int nk) { return kk + 1 < nk; }       // block kk + 1's activation fragments
__host__ __device__ inline bool kl_issue(int kk, int nk, int rd) { return kk + rd < nk; }  // block kk + rd into block kk's ring slot
