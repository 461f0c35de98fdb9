// Single-plan latency, second stage: ks_rollout_cl2 -- the reward chain beside the dynamics chain.
//
// ks_rollout_cl (cluster_kernels.cuh) runs one CEM iteration of ONE plan (evaluate.py:80 -> tdmpc2.py:111) on 16 clusters of 8
// workgroups, 128 of the 256 CUs, and every cluster walks through the iteration's 17 layer + 6 head hand-overs one after the
// other (tdmpc2.py:122-136: reward and dynamics of step t, then the policy and the two Q heads).  A hand-over costs ~9.5 us
// whatever it carries, so the launch is 23 x 9.5 us.  But the reward chain of a step only CONSUMES z_t: nothing downstream of
// it feeds the dynamics.  Here every 32-row tile gets TWO clusters on the same XCD (all 256 CUs):
//     role D   per step: dyn.l0, dyn.l1, dyn.l2 -> z_{t+1} (its exchange tile Z[t] stays put for the launch);
//              then pi.l0, pi.l1, policy head -> a_H; q0.l0, q0.l1, Q head a; finally value = G + disc^H (Qa + Qb) / 2
//     role R   per step: (t > 0: z_t from D's Z[t-1] through the SimNorm epilogue -- the same registers-to-tile routine D's own
//              members run), rew.l0, rew.l1, reward head -> G; then z_H from Z[H-1], a_H from D's policy-head logits and the
//              same noise, q1.l0, q1.l1, Q head b; (G, Qb) to D's member 0 through a 256-byte mailbox
// D's critical path is 3 H + 5 layer hand-overs + 2 heads (H = 3: 16 instead of 23), R runs one step behind it and ends with
// its Q head at about the same time.  Same arithmetic as ks_rollout_cl, operation for operation: bit-identical values.
// Cross-cluster reads use D's arrival words (R polls them exactly as D's own members do; phase numbers are a function of
// (launch, step) alone) and agent-scope loads; D writes Z[t] and the policy-head tile with write-through stores whatever the
// placement.  Every wait is bounded and raises the handle's error word like the cluster path's.
// Launch 0 also computes the policy-prior trajectories (tdmpc2.py:154-160: rows < P of tile 0, a_t = pi(z_t)): tile 0's D cluster
// runs pi.l0, pi.l1 and the policy head in front of every step's dynamics (as ks_rollout_cl's cluster 0 does) and leaves each
// step's head logits in a tile of its own; tile 0's R cluster turns them into the same actions with the same noise.
// Used for single non-episodic plans.
// Included by k_cluster.hip after cluster_kernels.cuh.
#pragma once

constexpr int CL2_SLOTS = 6 + 2 * MAXH;  // exchange tiles per cluster: 0 / 1 layers, 4 head, 5 policy head, 6 + t: Z[t], 6 + MAXH + t: prior policy head of step t
constexpr int CL2_MAIL = 15;         // arrival word of R's cluster that carries the mailbox's launch tag

// poll the 8 arrival words of a PEER cluster until all have reached `phase` (bounded)
template <class CT>
__device__ __forceinline__ void cl2_wait_peer(const CT &c, ClState &x, const unsigned *pflags, unsigned phase) {
    if (c.tid < CL && !*x.dead) {
        WaitClock wc;
        while ((__hip_atomic_load(pflags + c.tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffffu) < phase) {
            if (wc.expired(x.err)) {
                raise_fault(x.err, 1u);
                *x.dead = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

template <int APAD>
__global__ __launch_bounds__(NTHREADS, 2) void ks_rollout_cl2(RolloutParamsT<NetS> p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int s_is_last, s_dead, s_fast;
    typedef CtxT<APAD, 1, 8, 0> CT;
    constexpr int TROWS = CT::TROWS, NTHR = CT::NTHR;
    constexpr int ZKB16 = CT::ZKB;
    const int tid = threadIdx.x;
    // blocks {64 g + xc + 8 r : r = 0..7} = cluster 8 g + xc on XCD xc; tile (g >> 1) * 8 + xc, role g & 1: both roles of a tile
    // share an XCD (speed only)
    const int xc = blockIdx.x & 7, bq = blockIdx.x >> 3;
    const int rank = bq & 7, g = bq >> 3;
    const int tile = (g >> 1) * 8 + xc, role = g & 1;
    if (tile >= p.tiles) return;  // (both clusters of the tile leave)
    const int cl = tile * 2 + role, peer = tile * 2 + (role ^ 1);
    const int e = 0;
    CT c{reinterpret_cast<_Float16 *>(smem), smem + TROWS * CT::RSF(), smem + TROWS * CT::RSF() + 1024, tid,
         __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63};
    float *sm_mean = smem + TROWS * CT::RSF() + 2048;
    float *sm_std = sm_mean + p.H * p.A;
    ClState x{p.cl2_xbuf + (size_t)cl * CL2_SLOTS * CL_TILE, p.cl2_flags + (size_t)cl * CL_FLAG_STRIDE, p.cl_err,
              smem + TROWS * CT::RSF() + 2048 + ((2 * p.H * p.A + 3) & ~3), &s_dead, &s_fast, rank, 0u,
              (unsigned)(p.iter * cl_phases(p.H)), (unsigned)(p.iter * cl_heads(p.H)), false, false};
    const float *peer_xbuf = p.cl2_xbuf + (size_t)peer * CL2_SLOTS * CL_TILE;
    const unsigned *peer_flags = p.cl2_flags + (size_t)peer * CL_FLAG_STRIDE;
    const unsigned base_ph = (unsigned)(p.iter * cl_phases(p.H)), base_hp = (unsigned)(p.iter * cl_heads(p.H));
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
    float *zs = p.cl2_zs + (size_t)(cl * CL + rank) * TROWS * WIDTH;
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
    const unsigned pf = pifold ? 5u : 3u;  // D's layer hand-overs per step
    if (role == 0) {
        if (pifold) gb_prefetch(c, p.pi.l[0].g, p.pi.l[0].b);
        else gb_prefetch(c, p.dyn.l[0].g, p.dyn.l[0].b);
    } else {
        gb_prefetch(c, p.rew.l[0].g, p.rew.l[0].b);
    }
    tile_broadcast_row_s(c, p.z0 + (size_t)e * WIDTH);  // z_0 in every row (tdmpc2.py:163)
    epi_barrier(c);
    if (pifold) {  // zs <- z_0 in register order (later steps: the SimNorm epilogue's copy): the policy head's staging view clobbers z
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

    // the sampled actions of step t into this member's tile (tdmpc2.py:176-181); D's member 0 also writes them out
    auto fill_actions = [&](int t) {
        float *ag = p.actions + ((size_t)e * p.H + t) * p.N * p.A;
        const int hp = p.Apad / 2;
        for (int idx = tid; idx < TROWS * hp; idx += NTHR) {
            const int row = idx / hp, a0 = 2 * (idx % hp);
            const int n = row0 + row;
            float v[2] = {0.f, 0.f};
            const bool sampled = !(n < p.P);
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
                        v[u] = pifold ? 0.f : ag[(size_t)n * p.A + a];  // policy-prior rows: written by launch 0 (there: the policy head below)
                    } else {
                        float r = z[u];
                        if (p.sample_eps)
                            r = p.sample_eps[(size_t)e * p.sample_eps_estride + (unsigned)(((size_t)t * (p.N - p.P) + (n - p.P)) * p.A + a)];
                        v[u] = sample_action(sm_mean[t * p.A + a], sm_std[t * p.A + a], r);
                    }
                    if (mask) v[u] *= mask[a];
                    if (role == 0 && rank == 0 && sampled) ag[(size_t)n * p.A + a] = v[u];
                }
                put_action(c, row, a, v[u]);
            }
        }
        __syncthreads();
    };
    // this member's copy of a peer exchange tile's epilogue (x with the peer's tiles)
    ClState xp = x;
    xp.xbuf = const_cast<float *>(peer_xbuf);
    auto eps_pi = [&](int row, int a) -> float {
        const unsigned ridx = (unsigned)((size_t)(row0 + row) * p.A + a);
        if (p.pi_eps) return p.pi_eps[(size_t)e * p.pi_eps_estride + ridx];
        return rng_normal(p.seed, p.call, SITE_PI, p.iter, e, ridx);
    };

    // D's policy-head logits (exchange tile `slot`, complete when its head arrival words reach `hphase`) -> this member's staging view
    auto peer_head_to_staging = [&](int slot, unsigned hphase) {
        const unsigned *hflags = peer_flags + 8;
        const int nct = p.pi.l[2].CT;
        if (tid < nct && !*x.dead) {
            WaitClock wc;
            while (__hip_atomic_load(hflags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < hphase) {
                if (wc.expired(x.err)) {
                    raise_fault(x.err, 1u);
                    *x.dead = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        const float *hbuf = peer_xbuf + (size_t)slot * CL_TILE;
        const int row = tid >> 4, c4 = (tid & 15) * 4;
        f32x4 v0, v1;
        cl_ld16x2(v0, v1, hbuf + row * 128 + c4, hbuf + row * 128 + 64 + c4);
        float *f = c.f32() + row * CT::RSF();  // (the staging view aliases the z columns' hi plane: z comes back from zs afterwards)
        *reinterpret_cast<f32x4 *>(f + c4) = v0;
        *reinterpret_cast<f32x4 *>(f + 64 + c4) = v1;
        __syncthreads();
    };
    // a_t of the policy-prior rows from the step's policy-head logits in the staging view (tdmpc2.py:156-160), then z_t back
    auto prior_actions = [&](int t, float *gdst) {
        const float *tape = p.pi_traj_eps ? p.pi_traj_eps + ((size_t)e * p.H + t) * p.P * p.A : nullptr;
        auto eps = [&](int row, int a) -> float {
            if (row >= p.P) return 0.f;
            if (tape) return tape[row * p.A + a];
            return rng_normal(p.seed, p.call, SITE_PITRAJ, t, e, (unsigned)(row * p.A + a));
        };
        head_pi_rows_s(c, p.A, p.Apad, p.log_std_min, p.log_std_dif, mask, eps, gdst, p.P, nullptr, nullptr, nullptr, p.P, true);
        tile_from_global_s(c, zs);
        __syncthreads();
    };

    if (role == 0) {
        // ================================================================ D: dynamics, policy, first Q head, the value
        for (int t = 0; t < p.H; ++t) {
            fill_actions(t);
            if (pifold) {
                cl_layer<0>(c, x, CL_L(p.pi.l[0]), b_pi, 0, ZKB16, 0, gb_of(p.pi.l[1]));
                cl_layer<0>(c, x, CL_L(p.pi.l[1]), p.pi.l[1].bias, 0, ZKB16, 1, gb_of(p.dyn.l[0]));
                x.pub = 1;
                cl_head_logits(c, x, p.pi.l[2], true, 6 + MAXH + t);
                x.pub = 0;
                prior_actions(t, rank == 0 ? p.actions + ((size_t)e * p.H + t) * p.N * p.A : nullptr);
            }
            cl_layer<0>(c, x, CL_L(p.dyn.l[0]), b_dyn, 0, KBA, 0, gb_of(p.dyn.l[1]));
            cl_layer<0>(c, x, CL_L(p.dyn.l[1]), p.dyn.l[1].bias, 0, ZKB16, 1, gb_of(p.dyn.l[2]));
            x.pub = 1;  // Z[t] is read by the other cluster: write-through whatever the placement
            cl_layer<1>(c, x, CL_L(p.dyn.l[2]), p.dyn.l[2].bias, 0, ZKB16, 6 + t, (t == p.H - 1 || pifold) ? gb_of(p.pi.l[0]) : gb_of(p.dyn.l[0]),
                        (t == p.H - 1 || pifold) ? zs : nullptr);
            x.pub = 0;
        }
        // a_H = pi(z_H) (tdmpc2.py:135); z_H was also saved to zs
        cl_layer<0>(c, x, CL_L(p.pi.l[0]), b_pi, 0, ZKB16, 0, gb_of(p.pi.l[1]));
        cl_layer<0>(c, x, CL_L(p.pi.l[1]), p.pi.l[1].bias, 0, ZKB16, 1, gb_of(p.q[q0].l[0]));
        x.pub = 1;
        cl_head_logits(c, x, p.pi.l[2], true, 5);
        x.pub = 0;
        head_pi_rows_s(c, p.A, p.Apad, p.log_std_min, p.log_std_dif, mask, eps_pi, nullptr, 0, nullptr);
        tile_from_global_s(c, zs);
        __syncthreads();
        // Qa(z_H, a_H)
        cl_layer<0>(c, x, CL_L(p.q[q0].l[0]), b_q0, 0, KBA, 0, gb_of(p.q[q0].l[1]));
        cl_layer<0>(c, x, CL_L(p.q[q0].l[1]), p.q[q0].l[1].bias, 0, ZKB16, 1, GB{});
        const float qa = cl_head_twohot(c, x, p.q[q0].l[2], p.bins, p.num_bins);
        if (rank != 0) return;  // member 0 owns the values
        // (G, Qb) of the tile's rows from R's member 0
        if (tid == 0 && !*x.dead) {
            WaitClock wc;
            while (__hip_atomic_load(peer_flags + CL2_MAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(p.iter + 1)) {
                if (wc.expired(x.err)) {
                    raise_fault(x.err, 1u);
                    *x.dead = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        float Gv = 0.f, qb = 0.f;
        if ((tid & 7) == 0 && live) {
            const float *mb = p.cl2_mail + ((size_t)tile * TROWS + (tid >> 3)) * 2;
            Gv = __hip_atomic_load(mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            qb = __hip_atomic_load(mb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const float val = Gv + disc[p.H] * ((qa + qb) / 2.f);
        if (!p.fold_refit) {
            if ((tid & 7) == 0 && live) p.value[(size_t)e * p.N + row0 + (tid >> 3)] = val;
            return;
        }
        // elite selection + refit by the last member-0 workgroup of the plan (the hand-over of ks_rollout / ks_rollout_cl)
        if ((tid & 7) == 0 && live)
            __hip_atomic_store(p.value + (size_t)e * p.N + row0 + (tid >> 3), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        return;
    }
    // ==================================================================== R: rewards, second Q head
    float G = 0.f;
    for (int t = 0; t < p.H; ++t) {
        if (t > 0) {  // z_t: D's Z[t - 1] through the SimNorm epilogue (c.gb holds dyn.l2's parameters: named by rew.l1 below)
            cl2_wait_peer(c, x, peer_flags, base_ph + pf * (unsigned)t);
            cl_epi<1>(c, xp, 6 + t - 1, CL_E(p.dyn.l[2]), p.dyn.l[2].bias, gb_of(p.rew.l[0]), pifold ? zs : nullptr);
        }
        fill_actions(t);
        if (pifold) {  // the prior rows' a_t: D's policy-head logits of this step, the same noise
            peer_head_to_staging(6 + MAXH + t, base_hp + (unsigned)t + 1u);
            prior_actions(t, nullptr);
        }
        cl_layer<0>(c, x, CL_L(p.rew.l[0]), b_rew, 0, KBA, 0, gb_of(p.rew.l[1]));
        cl_layer<0>(c, x, CL_L(p.rew.l[1]), p.rew.l[1].bias, 0, ZKB16, 1, gb_of(p.dyn.l[2]));
        const float r = cl_head_twohot(c, x, p.rew.l[2], p.bins, p.num_bins);
        G += disc[t] * r;
    }
    // z_H, then a_H from D's policy-head logits and the same noise
    cl2_wait_peer(c, x, peer_flags, base_ph + pf * (unsigned)p.H);
    cl_epi<1>(c, xp, 6 + p.H - 1, CL_E(p.dyn.l[2]), p.dyn.l[2].bias, gb_of(p.q[q1].l[0]), zs);
    peer_head_to_staging(5, base_hp + (pifold ? (unsigned)p.H : 0u) + 1u);
    head_pi_rows_s(c, p.A, p.Apad, p.log_std_min, p.log_std_dif, mask, eps_pi, nullptr, 0, nullptr);
    tile_from_global_s(c, zs);
    __syncthreads();
    // Qb(z_H, a_H)
    cl_layer<0>(c, x, CL_L(p.q[q1].l[0]), b_q1, 0, KBA, 0, gb_of(p.q[q1].l[1]));
    cl_layer<0>(c, x, CL_L(p.q[q1].l[1]), p.q[q1].l[1].bias, 0, ZKB16, 1, GB{});
    const float qb = cl_head_twohot(c, x, p.q[q1].l[2], p.bins, p.num_bins);
    if (rank != 0) return;
    if ((tid & 7) == 0 && live) {
        float *mb = p.cl2_mail + ((size_t)tile * TROWS + (tid >> 3)) * 2;
        __hip_atomic_store(mb, G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mb + 1, qb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(const_cast<unsigned *>(x.flags) + CL2_MAIL, (unsigned)(p.iter + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
