// f16x2-split arithmetic (see fused_kernels.cuh) for the layer-at-a-time family: the 19M / 48M / 317M world models.
//
// Activations that feed a GEMM are stored in HBM in OPERAND FORM, FRAGMENT-PACKED exactly like the weights: a buffer of
// `KB` k16-blocks per row holds, for every 32-row tile r32 and k16-block kb, 2 KiB at ((r32 * KB + kb) * 2048): the hi plane
// (1 KiB) then the lo plane, each [64 lanes][8 halfs] with lane = 32 (k >> 3 & 1) + (row & 31) -- the register image of a
// v_mfma_f32_32x32x16_f16 operand (hi = f16(s x), lo = f16(s x - hi), s the layer's power-of-two operand scale).  One
// wavefront-wide 16-byte load / LDS-DMA fetches a whole fragment plane as 1 KiB of contiguous bytes, its LDS image is read
// back with conflict-free ds_read_b128 at lane * 16, and a GEMM epilogue writes 512 (or 1024) contiguous bytes per store
// instruction.  Same bytes per row as fp32 (KB * 64).  opnd_off() below is the one definition of the layout; the row
// kernels that produce an activation (l_ln_act_s, l_init_x_s, l_set_action_s, l_pi_head_s) and the GEMM epilogues use it.
// GEMM outputs that are not operands (head logits, unfused pre-activations) stay fp32 row-major.
// g_gemm_s<NCT>: 128 x (128 NCT) output tile per 256-thread workgroup; wave w owns 32 NCT output columns for all 128 rows
// (4 row tiles x NCT column tiles): every weight fragment (2 KB per k16-block, from L2) feeds 12 MFMAs and the row
// fragments come from LDS (8 ds_read_b128 per block, shared by the NCT column tiles).
// Included by tdmpc2_plan.hip inside its anonymous namespace, after fused_kernels.cuh and layered_kernels.cuh.
#pragma once

// byte offset of the hi half of element (row, col) in a fragment-packed operand buffer with KB k16-blocks per row (lo: + 1024)
__host__ __device__ __forceinline__ size_t opnd_off(size_t row, int col, int KB) {
    return ((row >> 5) * (size_t)KB + (size_t)(col >> 4)) * 2048 + (size_t)((((col >> 3) & 1) << 5) + (int)(row & 31)) * 16 + (size_t)(col & 7) * 2;
}

struct GemmSParams {
    const _Float16 *A;  // fragment-packed operand buffer (see above) with KBa k16-blocks per row
    int KBa, a_kb0;     // blocks per row of A; the first block contracted (a k-range: the action columns of X)
    int K;              // contraction length, multiple of GBK
    int kb0, kbs;       // a k-range of the packed matrix: first k16-block and blocks per column tile (0: K / 16 = the whole matrix)
    const _Float16 *wp; // split-packed [CT][K/16][2][64][8] (k_pack_split), + sel * w_sel_stride (in halfs)
    long w_sel_stride;
    const float *oscale; // device scalar(s): 2^-(kw + 5), + sel * osc_sel_stride
    long osc_sel_stride;
    int CT, ncolblk;
    const float *bias;
    long bias_env_stride, bias_sel_stride;
    const int *sel;
    long sel_stride;
    int rows_per_env;
    const int *row_env; // optional per-row env of the bias lookup (see GemmParams)
    float *out;         // fp32 [Rp, ldo]  (EPI != 0: fragment-packed operand buffer with KBo k16-blocks per row)
    int ldo, KBo;
    // ---- EPI != 0: LayerNorm + activation + hi / lo split in the epilogue (no fp32 round trip, no row kernel)
    const float *ln_g, *ln_b;  // LayerNorm weight / bias [width], + sel * gb_sel_stride
    long gb_sel_stride;
    const float *ascale;       // operand scale of a Mish layer's output (LayerScal::ascale), + sel * asc_sel_stride
    long asc_sel_stride;
    int width;                 // LayerNorm width = out_features (a multiple of 32)
    float *stats;              // [row block][column block][rows of the tile][2]: (mean, M2) of a workgroup's columns
    unsigned int *arrive;      // [row block] arrival counters of this launch (zero on entry)
    unsigned int *err;         // the handle's host-mapped error word (bounded wait gave up)
    int fault;                 // test hook (TDMPC2_CLUSTER_FAULT at create): column block 0 of row block 0 never arrives
    // EPI != 0: the tile order (tile_order.h: gemm_s_order / gemm_s_tile)
    int xcd_rows, nrowblk, ncol_grid;
    // g_gemm_w with a K-split tail (tile_order.h: gemm_w_order / gemm_w_tile); ks_parts <= 1: off, the order above applies
    int ks_parts, ks_full, ks_max_tail;
    float *ks_ws;              // [split tile slot][part][32 chunks][512 threads][4]: partial accumulators in register order
    unsigned int *ks_cnt;      // [split tile slot] arrival tickets of this launch (zero on entry)
    unsigned long long *timing;  // g_gemm_w profiling builds (-DGW_TIMING), else null
    TwoHotParams th;             // EPI = 3: what the two-hot value of a row goes into (lg / ld unused: the logits stay in LDS)
};

// One step of Chan's parallel (mean, M2) combination: (n_acc, m_acc, q_acc) <- combined with a block of nb values of mean mb and
// M2 qb.  Every kernel that folds LayerNorm partials uses THIS function, with explicitly rounded operations (no fused
// multiply-add contraction the compiler could apply in one kernel and not in another): the row statistics -- and with them a
// plan's bits -- do not depend on which GEMM tile computed them.
__device__ __forceinline__ void chan_fold(float &n_acc, float &m_acc, float &q_acc, float nb, float mb, float qb) {
    const float nt = __fadd_rn(n_acc, nb), dl = __fsub_rn(mb, m_acc);
    m_acc = __fadd_rn(m_acc, __fmul_rn(dl, __fdiv_rn(nb, nt)));
    q_acc = __fadd_rn(q_acc, __fadd_rn(qb, __fmul_rn(__fmul_rn(dl, dl), __fdiv_rn(__fmul_rn(n_acc, nb), nt))));
    n_acc = nt;
}

// (bounded waits: WaitClock, common.cuh -- 5 ms by the wall clock; a healthy wait is microseconds -- peers of a row block are
// dispatched back to back -- or, when a row block straddles the residency limit of its XCD, one tile's run time)

// NCT = 32-wide output column tiles per wave: 1 -> 128 x 128 workgroup tile (narrow outputs: heads, small models),
// 2 -> 128 x 256 (a wave owns 64 columns x 128 rows = 8 accumulators: per k16-block 4 KB of weight fragments and 8 KB of
// row fragments feed 24 MFMAs, 512 operand bytes per MFMA against 853 with NCT = 1 -- operand delivery, not MFMA issue,
// is what bounds these loops, profiles/README.md).
// RT = 32-row tiles per workgroup: 4 (128 rows, throughput) or 2 / 1 (64 / 32 rows: calls with few rows -- single-plan
// latency of the 48M / 317M models, where 128-row tiles leave most of the chip idle: c3 at E = 1 is 4 x 14 = 56
// workgroups on 512 slots; every weight fragment then feeds fewer MFMAs, which does not matter while the chip is not full).
// SD = chunks of the row operand in flight (global -> registers -> LDS): 1 for the throughput tiles, whose neighbours on
// the CU cover the latency; 4 -- together with a weight ring of 8 / 16 k16-blocks instead of 4 -- for calls with few rows (one
// or two workgroups per CU: single-plan latency of the 48M / 317M models), where the k-loop ran at one L2 / Infinity-Cache
// round trip per two chunks: c3 single plan 6.17 -> 5.25 ms, c4 24.6 -> 23.0 ms, bit-identical sums (profiles/README.md r03c).
// (Measured and rejected there: split-K over 4 / 8 workgroups per tile with the consumer adding the slices -- slower: the
// partial sums' traffic; one accumulator per product kind on the 32-row tile -- no change.)
// EPI = 0: out <- acc * oscale + bias (fp32 pre-activation, head logits).
// EPI = 1 / 2: the NormedLinear epilogue (layers.py:94-118) -- LayerNorm over the WHOLE output row, then Mish (1) or SimNorm
// (2), then the hi / lo operand split -- inside the GEMM.  A row's columns are spread over the `ncolblk` workgroups of its
// row block, so these exchange per-row (mean, M2) of their own columns through L2 / HBM: agent-scope stores, one arrival
// counter per row block, a bounded wait, agent-scope loads, Chan's parallel combination (as exact as a two-pass variance).
// The MFMA runs with the weight fragment as the A operand here, so that a lane holds 16 NCT FEATURES of one row (thread-local
// row sums, one lane ^ 32 exchange) and writes 4 consecutive features per store.  The workgroups of a row block must be
// dispatched together: row-major tile order (never the strip order of gemm_tile_of_block), in-order dispatch per XCD; a wait
// that gives up raises the handle's error word (the plan then returns NaN, tdmpc2_plan_take_fault).
// EPI = 3 (NCT = 1, RT <= 2: the narrow two-hot heads -- reward, Q): two_hot_inv (math.py:74-83) in the epilogue.  The workgroup
// holds all 128 logit columns of its rows: they are staged in LDS as the fp32 values EPI = 0 would have written, and each
// wavefront runs l_twohot's own row routine (twohot_wave + twohot_apply) on its rows -- the same bits as GEMM + l_twohot,
// without the logits' round trip through HBM and the extra launch.
// PF = k16-blocks of the weight ring (0: the rule below).  PF = 3 on the throughput tile: three chunks per trip, weight
// fragments requested 1.5 chunks ahead instead of 1 (16 more VGPRs).
template <int NCT, int RT = 4, int SD = 1, int EPI = 0, int PF = 0>
__global__ __launch_bounds__(GTHREADS, 2) void g_gemm_s(GemmSParams p) {
    constexpr int TM = 32 * RT;  // rows of this workgroup's tile
    // LDS image of a 32-wide chunk: the chunk's RT x 2 (k16-blocks) x 2 (planes) fragment planes, 1 KiB each, as they lie in HBM
    __shared__ __attribute__((aligned(16))) _Float16 As[2][RT * 4][512];  // [buffer][(row tile, k16-block, plane)][lane * 8]
    constexpr int LGS_LD = 132;  // floats per staged logit row (EPI = 3)
    __shared__ __attribute__((aligned(16))) float lgs[EPI == 3 ? TM * LGS_LD : 4];
    static_assert(EPI != 3 || (NCT == 1 && RT <= 2), "two-hot epilogue: one 128-column block, at most 64 rows of logits in LDS");
    constexpr bool LN = EPI == 1 || EPI == 2;  // NormedLinear epilogue: transposed accumulator tile, tile order of tile_order.h
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rb, cb;
    if (!LN) {
        gemm_tile_of_block(blockIdx.x, gridDim.x, p.ncolblk, rb, cb);
    } else {  // tile_order.h: XCD-local row blocks, or (padded) row-major
        if (!gemm_s_tile(blockIdx.x, p.nrowblk, p.ncolblk, p.xcd_rows, p.ncol_grid, rb, cb)) return;
    }
    const int row0 = rb * TM;
    const int sel = p.sel ? p.sel[(size_t)(row0 / p.rows_per_env) * p.sel_stride] : 0;
    const int KB = p.K / 16;
    const int ct0 = (cb * 4 + wave) * NCT;
    // weight fragments: wave-uniform byte pointers + opaque 32-bit lane offset (see fused_kernels.cuh); a column tile past
    // the matrix re-reads the last one and is dropped in the epilogue
    const char *u[NCT];
#pragma unroll
    for (int n = 0; n < NCT; ++n) {
        const int ct = ct0 + n < p.CT ? ct0 + n : p.CT - 1;
        u[n] = reinterpret_cast<const char *>(p.wp + (size_t)sel * p.w_sel_stride) + ((size_t)ct * (p.kbs ? p.kbs : KB) + p.kb0) * 2048;
    }
    unsigned voff = (unsigned)lane * 16u;
    asm volatile("" : "+v"(voff));

    // A staging: wave w copies plane (w & 1) of k16-block (w >> 1) of the chunk for each of the RT row tiles: 1 KiB contiguous
    // per wave and instruction, LDS image = HBM image
    const char *ab = reinterpret_cast<const char *>(p.A) + ((size_t)(row0 >> 5) * p.KBa + p.a_kb0) * 2048;
    int g_off[RT], l_off[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        g_off[i] = (i * p.KBa + (wave >> 1)) * 2048 + (wave & 1) * 1024 + lane * 16;
        l_off[i] = (4 * i + wave) * 1024 + lane * 16;
    }
    const int nchunks = p.K / GBK;
    f32x4 stage[SD][RT];
#pragma unroll
    for (int d = 0; d < SD; ++d)
        if (d < nchunks) {
#pragma unroll
            for (int i = 0; i < RT; ++i) stage[d][i] = *reinterpret_cast<const f32x4 *>(ab + g_off[i] + (size_t)d * 4096);
        }
    char *lds = reinterpret_cast<char *>(&As[0][0][0]);
    constexpr int BUF_BYTES = RT * 4 * 1024;
#pragma unroll
    for (int i = 0; i < RT; ++i) *reinterpret_cast<f32x4 *>(lds + l_off[i]) = stage[0][i];
    __syncthreads();

    f32x16 acc[NCT][RT];
#pragma unroll
    for (int n = 0; n < NCT; ++n)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][r][e] = 0.f;

    const int i32 = lane & 31, hh = lane >> 5;
    // weight ring in k16-blocks (NCT x 8 VGPRs each); few-row variants (SD = 4): 8 ... 16 blocks
    constexpr int PFB = PF ? PF : SD > 2 ? (RT == 1 ? 16 : 8) : (NCT == 1 ? 4 : 2);
    f16x8 rh[PFB][NCT], rl[PFB][NCT];
#pragma unroll
    for (int d = 0; d < PFB; ++d) {
        const int kd = d < KB ? d : KB - 1;
#pragma unroll
        for (int n = 0; n < NCT; ++n) {
            rh[d][n] = ldw(u[n] + (size_t)kd * 2048, voff, 0);
            rl[d][n] = ldw(u[n] + (size_t)kd * 2048, voff, 1024);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int U = SD > 2 ? PFB / 2 : (PFB % 2 ? 2 * PFB : 2);  // chunks per trip (even: the LDS buffer of a chunk is then a compile-time choice): a whole number of turns of the weight ring and of the staging ring
    static_assert(U % SD == 0 && (2 * U) % PFB == 0, "staging depth 1, 2 or 4; ring turns per trip");
    // One chunk (32 of K = two k16-blocks).  `steady`: compile-time true in the main loop, whose trips contain NO conditional --
    // with the "is there a next chunk / a chunk to request" tests inside, hipcc's wait-count insertion loses track of the
    // vector-memory queue at every join and waits for (nearly) everything in front of the staging store: the deep weight ring
    // then bought nothing (s_waitcnt vmcnt(4) where vmcnt(15) would do; 0.39 us per chunk, profiles/README.md r03d).
    auto chunk = [&](const int ch, const int cc, const bool steady) __attribute__((always_inline)) {
        const bool more = steady || ch + 1 < nchunks;
        if (steady || ch + SD < nchunks) {  // chunk ch's slot is free (its rows went to LDS one chunk ago): request chunk ch + SD
#pragma unroll
            for (int i = 0; i < RT; ++i)
                stage[cc % SD][i] = *reinterpret_cast<const f32x4 *>(ab + g_off[i] + (size_t)(ch + SD) * 4096);
        }
        const _Float16 *af = &As[ch & 1][0][0] + lane * 8;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int d = (cc * 2 + kb) % PFB;
            f16x8 fh[RT], fl[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                fh[rt] = *reinterpret_cast<const f16x8 *>(af + ((rt * 2 + kb) * 2 + 0) * 512);
                fl[rt] = *reinterpret_cast<const f16x8 *>(af + ((rt * 2 + kb) * 2 + 1) * 512);
            }
            // EPI != 0: weight fragment as the A operand -> C[feature][row] (same sums, transposed accumulator tile)
#pragma unroll
            for (int n = 0; n < NCT; ++n)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[n][rt] = LN ? SPLIT_MFMA(rh[d][n], fh[rt], acc[n][rt]) : SPLIT_MFMA(fh[rt], rh[d][n], acc[n][rt]);
#pragma unroll
            for (int n = 0; n < NCT; ++n)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[n][rt] = LN ? SPLIT_MFMA(rl[d][n], fh[rt], acc[n][rt]) : SPLIT_MFMA(fh[rt], rl[d][n], acc[n][rt]);
#pragma unroll
            for (int n = 0; n < NCT; ++n)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[n][rt] = LN ? SPLIT_MFMA(rh[d][n], fl[rt], acc[n][rt]) : SPLIT_MFMA(fl[rt], rh[d][n], acc[n][rt]);
            const int kn = ch * 2 + kb + PFB;
            const int knc = kn < KB ? kn : KB - 1;
#pragma unroll
            for (int n = 0; n < NCT; ++n) {
                rh[d][n] = ldw(u[n] + (size_t)knc * 2048, voff, 0);
                rl[d][n] = ldw(u[n] + (size_t)knc * 2048, voff, 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
            char *dst = lds + ((ch + 1) & 1) * BUF_BYTES;
#pragma unroll
            for (int i = 0; i < RT; ++i) *reinterpret_cast<f32x4 *>(dst + l_off[i]) = stage[(cc + 1) % SD][i];
        }
        __syncthreads();
    };
    int c = 0;
#ifndef GEMM_NO_STEADY_LOOP
    for (; c + U + SD <= nchunks; c += U) {  // steady state: every chunk has a successor to store and a chunk to request
#pragma unroll
        for (int cc = 0; cc < U; ++cc) chunk(c + cc, cc, true);
    }
#endif
    for (; c < nchunks; c += U) {  // the last trips (and short contractions)
#pragma unroll
        for (int cc = 0; cc < U; ++cc)
            if (c + cc < nchunks) chunk(c + cc, cc, false);  // uniform
    }

    const float osc = p.oscale[(size_t)sel * p.osc_sel_stride];
    const float *bsel = p.bias + (size_t)sel * p.bias_sel_stride;
    if constexpr (LN) {
        // transposed C fragment: lane holds row (l & 31) of row tile rt and features (reg & 3) + 8 (reg >> 2) + 4 (l >> 5) of
        // column tile ct0 + n.  LDS (the staging buffers are idle now): per-wave partials, then the rows' (mean, rstd).
        __syncthreads();  // every wave is done with the last chunk's fragments
        // The combination ORDER is a function of the layer alone, never of the tile shape, the row count or the number of
        // plans in the call: per 32-column tile (mean, M2); tiles folded left to right inside groups of 128 columns; groups
        // folded left to right over the row.  A plan therefore computes the same bits alone, inside a batch and when its
        // rows are split over ranks, whichever g_gemm_s variant the call size selects.
        static_assert((8 * NCT + 2) * TM * 4 <= (int)sizeof(As), "the epilogue's tables fit the idle staging buffers");
        float *red = reinterpret_cast<float *>(&As[0][0][0]);         // [4 NCT tiles of the column block][TM][2]
        float *rs = red + 4 * NCT * TM * 2;                            // [TM][2]
        // (1) v = acc * oscale + bias, in place
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = row0 + rt * 32 + i32;
            const float *bp = bsel;
            if (p.bias_env_stride != 0) bp += (size_t)(p.row_env ? p.row_env[row] : row / p.rows_per_env) * p.bias_env_stride;
#pragma unroll
            for (int n = 0; n < NCT; ++n) {
                const int ct = ct0 + n < p.CT ? ct0 + n : p.CT - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bp + ct * 32 + 8 * j + 4 * hh);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[n][rt][4 * j + r] = fmaf(acc[n][rt][4 * j + r], osc, b4[r]);
                }
            }
        }
        // (2) (mean, M2) of every row over each 32-column tile: 16 thread-local values + the lane ^ 32 half
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int n = 0; n < NCT; ++n) {
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
                    red[((wave * NCT + n) * TM + rt * 32 + i32) * 2 + 0] = mw;
                    red[((wave * NCT + n) * TM + rt * 32 + i32) * 2 + 1] = q;
                }
            }
        __syncthreads();
        // (3) fold the tiles of each 128-column group left to right -> stats[row block][group][row]
        const int NG = (p.CT + 3) / 4;  // groups of the whole row
        for (int idx = tid; idx < NCT * TM; idx += GTHREADS) {
            const int g = idx / TM, r = idx - g * TM, G = cb * NCT + g;  // group g of this column block = group G of the row
            if (G >= NG) continue;
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
        // (4) arrive (stores acknowledged first), wait for the row block's other column blocks -- bounded
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (!(p.fault && rb == 0 && cb == 0)) __hip_atomic_fetch_add(p.arrive + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            WaitClock wc;
            while (__hip_atomic_load(p.arrive + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.ncolblk) {
                if (wc.expired(p.err)) {
                    if (p.err) raise_fault(p.err, 2u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
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
        // (6) normalise, activate, split -> the fragment-packed output.  A lane holds features 8 j + 4 hh + (0..3) of row i32 of
        // column tile ct: k16-block 2 ct + (j >> 1), k-half j & 1 -> 8 bytes at lane' = 32 (j & 1) + i32, + 8 hh: one store
        // instruction of the wave covers 512 contiguous bytes of a fragment plane.
        const float *gsel = p.ln_g + (size_t)sel * p.gb_sel_stride, *besel = p.ln_b + (size_t)sel * p.gb_sel_stride;
        const float oscl = EPI == 1 ? p.ascale[(size_t)sel * p.asc_sel_stride] : ACT_SCALE;
        float rmean[RT], rrstd[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            rmean[rt] = rs[2 * (rt * 32 + i32)];
            rrstd[rt] = rs[2 * (rt * 32 + i32) + 1];
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            char *otile = reinterpret_cast<char *>(p.out) + (size_t)((row0 >> 5) + rt) * p.KBo * 2048 + i32 * 16 + hh * 8;
#pragma unroll
            for (int n = 0; n < NCT; ++n) {
                if (ct0 + n >= p.CT) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 g4 = *reinterpret_cast<const f32x4 *>(gsel + (ct0 + n) * 32 + 8 * j + 4 * hh);
                    const f32x4 be4 = *reinterpret_cast<const f32x4 *>(besel + (ct0 + n) * 32 + 8 * j + 4 * hh);
                    f32x4 y;
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = fmaf((acc[n][rt][4 * j + r] - rmean[rt]) * rrstd[rt], g4[r], be4[r]);
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
                    char *o = otile + (size_t)((ct0 + n) * 2 + (j >> 1)) * 2048 + (j & 1) * 512;
                    *reinterpret_cast<f16x4 *>(o) = hi;
                    *reinterpret_cast<f16x4 *>(o + 1024) = lo;
                }
            }
        }
        return;
    }
    if constexpr (EPI == 3) {
        // logits -> LDS (C fragment: lane holds column (l & 31) of this wave's tile, rows (reg & 3) + 8 (reg >> 2) + 4 (l >> 5))
        const int colv = ct0 < p.CT ? ct0 * 32 + i32 : 0;
        const float bv = ct0 < p.CT ? bsel[colv] : 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                lgs[r * LGS_LD + wave * 32 + i32] = fmaf(acc[0][rt][reg], osc, bv);
            }
        __syncthreads();
        for (int r = wave; r < TM; r += 4) {  // one wavefront per row, as l_twohot
            const int row = row0 + r;
            if (row >= p.th.rows) continue;
            const float v = twohot_wave(lgs + r * LGS_LD, p.th.bins, p.th.num_bins, lane);
            if (lane == 0) twohot_apply(p.th, row, v);
        }
        return;
    }
    // epilogue: acc * oscale + bias -> fp32.  C fragment: lane holds column (l & 31), rows (reg&3) + 8 (reg>>2) + 4 (l>>5).
#pragma unroll
    for (int n = 0; n < NCT; ++n) {
        if (ct0 + n >= p.CT) continue;
        const int col = (ct0 + n) * 32 + i32;
        const float bshared = p.bias_env_stride == 0 ? bsel[col] : 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = row0 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                float bv = bshared;
                if (p.bias_env_stride != 0)
                    bv = bsel[(size_t)(p.row_env ? p.row_env[row] : row / p.rows_per_env) * p.bias_env_stride + col];
                p.out[(size_t)row * p.ldo + col] = fmaf(acc[n][rt][reg], osc, bv);
            }
    }
}

// ---------------------------------------------------------------- row kernels writing operand form (fragment-packed)
__device__ __forceinline__ void put_split(char *buf, int KB, size_t row, int col, float v) {  // bounded operands: latents, actions
    const float vs = v * ACT_SCALE;
    const _Float16 h = (_Float16)vs;
    _Float16 *o = reinterpret_cast<_Float16 *>(buf + opnd_off(row, col, KB));
    o[0] = h;
    o[512] = (_Float16)(vs - (float)h);
}

// One wavefront per row: fp32 pre-activation row pre[row * ldpre + (0 .. width)] (an unfused GEMM's output, width <= 4096)
// -> ACT(LayerNorm(.)) -> columns [0, width) of row `row` of the fragment-packed operand buffer `out` (KBo k16-blocks per
// row); the buffer's other columns are left alone (X: action and padding columns).
template <int ACT>
__global__ __launch_bounds__(RW_THREADS) void l_ln_act_s(LnActParams p) {
    const int row = blockIdx.x * (RW_THREADS / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= p.rows) return;
    const int sel = p.sel ? p.sel[(size_t)(row / p.rows_per_env) * p.sel_stride] : 0;
    const float *g = p.g + (size_t)sel * p.gb_sel_stride, *bb = p.b + (size_t)sel * p.gb_sel_stride;
    const float *xr = p.x + (size_t)row * p.ld;
    const int n4 = p.width / 4;
    f32x4 v[16];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c4 = lane + 64 * q;
        if (c4 < n4) {
            v[q] = *reinterpret_cast<const f32x4 *>(xr + 4 * c4);
            s += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
        }
    }
    const float mean = group_sum<64>(s) / (float)p.width;
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        if (lane + 64 * q < n4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[q][e] - mean;
                ss = fmaf(d, d, ss);
            }
        }
    }
    const float var = group_sum<64>(ss) / (float)p.width;
    const float rstd = 1.0f / sqrtf(var + LN_EPS);
    // operand scale of this layer's output: chosen at bind time for Mish layers (k_ascale), fixed for SimNorm outputs
    const float oscl = ACT == 0 ? p.ascale[(size_t)sel * p.asc_sel_stride] : ACT_SCALE;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c4 = lane + 64 * q;
        const bool ok = c4 < n4;
        if (ACT == 0 && !ok) continue;
        if (ACT == 1 && 64 * q >= n4) continue;  // whole wave out of range (uniform)
        f32x4 y;
        if (ok) {
            const f32x4 gg = *reinterpret_cast<const f32x4 *>(g + 4 * c4);
            const f32x4 be = *reinterpret_cast<const f32x4 *>(bb + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaf((v[q][e] - mean) * rstd, gg[e], be[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = -INFINITY;
        }
        if (ACT == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = mish_fast(y[e]);
        } else {
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
            char *o = p.out + opnd_off((size_t)row, 4 * c4, p.KBo);  // 4 consecutive columns never straddle an 8-column k-half
            *reinterpret_cast<f16x4 *>(o) = hi;
            *reinterpret_cast<f16x4 *>(o + 1024) = lo;
        }
    }
}

// Columns [c0, c1) (multiples of 8) of every row of the packed buffer X (KB k16-blocks per row) <- split(src(row, col)).
// One workgroup per 32-row tile; a thread writes the 8 halfs (16 bytes per plane) of one (row, 8-column group): a wavefront's
// store covers two runs of 512 contiguous bytes.
// gridDim.y > 1: the tile's (row, column group) items are dealt over the blocks of a column of the grid.
template <class SRC>
__device__ __forceinline__ void fill_packed_tile(char *X, int KB, int c0, int c1, SRC src) {
    const int ng = (c1 - c0) >> 3;  // 8-column groups
    for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < ng * 32; idx += blockDim.x * gridDim.y) {
        const int r = idx & 31, gq = idx >> 5, col = c0 + 8 * gq;
        const size_t row = (size_t)blockIdx.x * 32 + r;
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float vs = src(row, col + e) * ACT_SCALE;
            const _Float16 h = (_Float16)vs;
            hi[e] = h;
            lo[e] = (_Float16)(vs - (float)h);
        }
        char *o = X + opnd_off(row, col, KB);
        *reinterpret_cast<f16x8 *>(o) = hi;
        *reinterpret_cast<f16x8 *>(o + 1024) = lo;
    }
}

// X[row] <- operand form of [z0[env] | zeros]; G, term <- 0.  One workgroup per 32-row tile (rows past `rows`: zeros).
// (gridDim.y: column chunks -- a single plan has ONE tile, and one workgroup took 40 us over its ldx / 8 x 32 items, r6zw)
__global__ void l_init_x_s(float *X, int ldx, int L, int rows_per_env, const float *z0, float *G, float *term, int rows) {
    fill_packed_tile(reinterpret_cast<char *>(X), ldx / 16, 0, ldx, [&](size_t row, int c) {
        return (c < L && row < (size_t)rows) ? z0[(row / rows_per_env) * L + c] : 0.f;
    });
    if (threadIdx.x < 32 && blockIdx.y == 0) {
        const size_t row = (size_t)blockIdx.x * 32 + threadIdx.x;
        if (row < (size_t)rows) {
            if (G) G[row] = 0.f;
            if (term) term[row] = 0.f;
        }
    }
}

// the action columns [L, ldx) of X <- split(actions[e, t, n, :]) (zeros past A).  One workgroup per 32-row tile; L % 8 == 0.
__global__ void l_set_action_s(float *X, int ldx, int L, int A, int N, int H, int t, int rows, const float *actions, int nsub, int n_off) {
    fill_packed_tile(reinterpret_cast<char *>(X), ldx / 16, L, ldx, [&](size_t row, int c) {
        const int a = c - L;
        if (a >= A || row >= (size_t)rows) return 0.f;
        const size_t e = row / nsub, n = n_off + row % nsub;
        return actions[((e * H + t) * N + n) * A + a];
    });
}

__global__ void l_pi_head_s(PiHeadParams p) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.rows * p.A) return;
    const int row = idx / p.A, a = idx % p.A;
    const int e = row / p.rows_per_env, n = row % p.rows_per_env;
    const float *lr = p.lg + (size_t)row * p.ld;
    float mu = lr[a];
    float ls = p.lsmin + 0.5f * p.lsdif * (tanhf(lr[p.A + a]) + 1.f);
    float eps = 0.f;
    if (n < p.nvalid) {
        const unsigned ridx = (unsigned)((size_t)(n + p.n_off) * p.A + a);
        eps = p.eps ? p.eps[(size_t)e * p.eps_estride + ridx] : rng_normal(p.seed, p.call, p.site, p.iter, e, ridx);
    }
    if (p.mask) {
        const float mk = p.mask[(size_t)(p.row_env ? p.row_env[row] : e) * p.A + a];
        mu *= mk;
        ls *= mk;
        eps *= mk;
    }
    const float act = tanhf(mu + eps * expf(ls));
    put_split(reinterpret_cast<char *>(p.X), p.ldx / 16, (size_t)row, p.L + a, act);
    if (p.actions && n < p.nvalid) p.actions[(((size_t)e * p.H + p.t) * p.N + n) * p.A + a] = act;
    if (p.trace) p.trace[(size_t)row * (p.H + 2 + p.A) + p.H + 2 + a] = act;
}

// X[row] <- operand form of [z[row] | zeros] (rows >= nvalid: zeros).  One workgroup per 32-row tile.
__global__ void l_init_rows_s(float *X, int ldx, int L, const float *z, int nvalid) {
    fill_packed_tile(reinterpret_cast<char *>(X), ldx / 16, 0, ldx, [&](size_t row, int c) {
        return (c < L && row < (size_t)nvalid) ? z[row * L + c] : 0.f;
    });
}
