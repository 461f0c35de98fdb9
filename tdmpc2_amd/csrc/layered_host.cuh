// Host orchestration of the layer-at-a-time planner path (kernels: layered_kernels.cuh, layered_split.cuh).
// Included by k_layered.hip inside namespace tdk (the functions declared in launch.h are the family's interface).
#pragma once


// LayerNorm + activation fused into the GEMM's epilogue (g_gemm_s<.., EPI>, split arithmetic): which NormedLinear's
// LayerNorm parameters / output scale, and whether the activation is Mish (0) or SimNorm (1).
struct LnFuse {
    int act;
    long gb_sel_stride;
    int width;
};

// Hidden-activation / logit buffers a chain of GEMMs runs through, and the exchange buffer of its fused epilogues: the
// handle holds two sets so that two independent chains (reward || dynamics of one step, the two Q heads) can be in flight on
// two streams at once -- the second chain's workgroups fill the slots the first one's last, partly filled round leaves idle
// and the launch gaps of one chain hide behind the other's kernels.  PRE: fp32 pre-activations of a NormedLinear whose
// epilogue is NOT fused (split arithmetic: the operand buffers are fragment-packed, so the LayerNorm kernel cannot work in place).
struct LayBufs {
    float *HA, *HB, *LG, *stats, *PRE, *ksws;
};
inline LayBufs lay_bufs(const tdmpc2_plan *h, int set) {
    const Layered &L = h->lay;
    return set == 0 ? LayBufs{L.HA, L.HB, L.LG, L.stats, L.PRE, L.ksws} : LayBufs{L.HA2, L.HB2, L.LG2, L.stats2, L.PRE2, L.ksws2};
}

// The arrival counters of the fused launches of one stage: a fresh slice per launch, ALL counters handed out so far zeroed at
// the start of every stage.  The memset covers [0, high-water mark of the handle) -- not "what the previous call used": under
// hipGraph capture the extent is frozen into the graph while replays do not update the host's bookkeeping, so an extent that
// depends on the call history can leave a counter of an earlier, larger eager call at its old value (a workgroup would then
// skip its wait and read stale statistics).  The high-water mark only grows; a graph captured after any warm-up of the same
// shape covers every slice its launches use.
int lay_arrive_reset(tdmpc2_plan *h, hipStream_t st) {
    Layered &L = h->lay;
    if (!L.arrive) return 0;
    // the iteration's l_sample has zeroed them (lay_sample_iteration) and nothing has taken a counter since: no memset
    const bool clean = L.arrive_clean && L.arrive_off == 0;
    L.arrive_clean = false;
    if (clean) return 0;
    if (L.arrive_off > L.arrive_high) L.arrive_high = L.arrive_off;
    if (L.arrive_high) HIP_TRY(hipMemsetAsync(L.arrive, 0, L.arrive_high * sizeof(unsigned int), st));
    L.arrive_off = 0;
    return 0;
}

#define GEMM_S_LAUNCH_PF(NCTV, RTV, SDV, PFV)                                                                  \
    do {                                                                                                      \
        if (epi == 0) hipLaunchKernelGGL((g_gemm_s<NCTV, RTV, SDV, 0, PFV>), dim3(nblk), dim3(GTHREADS), 0, st, q); \
        else if (epi == 1) hipLaunchKernelGGL((g_gemm_s<NCTV, RTV, SDV, 1, PFV>), dim3(nblk), dim3(GTHREADS), 0, st, q); \
        else hipLaunchKernelGGL((g_gemm_s<NCTV, RTV, SDV, 2, PFV>), dim3(nblk), dim3(GTHREADS), 0, st, q);     \
    } while (0)
#define GEMM_S_LAUNCH(NCTV, RTV, SDV) GEMM_S_LAUNCH_PF(NCTV, RTV, SDV, 0)
// the narrow tiles that can run the two-hot epilogue (epi == 3)
#define GEMM_S_LAUNCH_TH(RTV, SDV)                                                                              \
    do {                                                                                                      \
        if (epi == 3) hipLaunchKernelGGL((g_gemm_s<1, RTV, SDV, 3, 0>), dim3(nblk), dim3(GTHREADS), 0, st, q);  \
        else GEMM_S_LAUNCH(1, RTV, SDV);                                                                      \
    } while (0)

// A k-range of a layer with a per-environment bias of the caller's: the action columns of a first layer at t = 0, where the
// z columns' product is one vector per plan (lay_cvec).
struct GemmRange {
    int kb0, kblocks;        // first k16-block, number of blocks (a multiple of 2)
    const float *bias_env;   // [env, Mp]
    long bias_env_stride;
    int col_off;             // A's first column (a multiple of 16)
};

int lay_ln(tdmpc2_plan *h, hipStream_t st, int act, float *x, int ld, int width, size_t rows, int rows_per_env,
           const HostLayer &ly, long gb_sel_stride, const int *sel, float *out_packed = nullptr, int ld_out = 0);

// One nn.Linear over `rows_p` (padded) rows.  `slot` = index of the net in beff (multitask first layers), -1 otherwise.
// A / out: split arithmetic -- fragment-packed operand buffers of lda / ldo columns (out: only with `ln`; head logits and
// pre-activations are fp32 [rows, ldo]); exact fp32 -- fp32 [rows, ld].
// `ln` != null: the whole NormedLinear (layers.py:94-118) -- out <- ACT(LayerNorm(A W^T + b)) -- inside the GEMM's epilogue when the
// handle's fused path is on and fits, else by the LayerNorm row kernel behind the GEMM (split: through bufs->PRE).
int lay_gemm(tdmpc2_plan *h, hipStream_t st, const float *A, int lda, size_t rows_p, int rows_per_env,
             const HostLayer &ly, long w_sel_stride, long bias_sel_stride, int slot, const int *sel, float *out, int ldo,
             const LnFuse *ln = nullptr, const LayBufs *bufs = nullptr, const GemmRange *range = nullptr, size_t rows = 0,
             const TwoHotParams *th = nullptr, bool *th_done = nullptr) {
    if (th_done) *th_done = false;
    Layered &L = h->lay;
    const LayBufs b0 = lay_bufs(h, 0);
    if (!bufs) bufs = &b0;
    if (h->split) {
        GemmSParams q{};
        q.A = reinterpret_cast<const _Float16 *>(A); q.KBa = lda / 16; q.K = ly.KB * 16; q.wp = ly.wps;
        if (range) {
            q.a_kb0 = range->col_off / 16; q.K = range->kblocks * 16; q.kb0 = range->kb0; q.kbs = ly.KB;
        }
        q.w_sel_stride = sel ? w_sel_stride : 0; q.oscale = ly.oscale;
        q.osc_sel_stride = sel ? (long)(3 * sizeof(LayerScal) / sizeof(float)) : 0;  // the net's [heads][3] scalar table
        q.row_env = L.row_env;
        if (slot >= 0 && h->cfg.multitask) {
            q.bias = L.bias_tab + (size_t)slot * L.Mp;
            q.bias_env_stride = (long)h->nnets * L.Mp;
            q.bias_sel_stride = sel ? L.Mp : 0;
        } else {
            q.bias = ly.bias;
            q.bias_env_stride = 0;
            q.bias_sel_stride = sel ? bias_sel_stride : 0;
        }
        if (range && range->bias_env) {
            q.bias = range->bias_env; q.bias_env_stride = range->bias_env_stride; q.bias_sel_stride = 0;
        }
        q.sel = sel; q.sel_stride = 2; q.rows_per_env = rows_per_env; q.CT = ly.CT;
        const long cus = h->num_cus > 0 ? h->num_cus : 256;
        // Can the NormedLinear epilogue run inside the GEMM?  (always decided the same way whatever the tile: the statistics'
        // combination order is tile-independent, so a plan's bits do not depend on the size of the call it is part of)
        const bool can_fuse = ln && L.fuse_ln && L.arrive && bufs->stats && rows_p * ((ly.CT + 3) / 4) * 2 <= L.stats_cap;
        // ---- the 256 x 256 tile (g_gemm_w): fused NormedLinear layers of calls that fill the chip with one workgroup per CU
        const long w256_min = L.knob[LK_W256_MIN];
        // (Measured and rejected, profiles/README.md r4f: 256 x 224 / 192 tiles -- 8 x 1 wave layout, one W fragment set -- for widths
        // like the 48M model's 1792 = 8 x 224, which would fill 1.875 rounds of the chip instead of 1.64: the layout reads every W
        // fragment eight times from LDS (128 KiB per slab) and lost 4.5 % in spite of the better fill.)
        constexpr int NT = 8;
        const int ncb256 = (ly.CT + NT - 1) / NT;
        const size_t stats_need = rows_p * ((ly.CT + 3) / 4) * 2;
        // K-split tail (tile_order.h: gemm_w_order): tiles of the launch's last, partly filled round are computed by `parts`
        // workgroups each.  That makes the wide tile worth taking from fewer tiles on (a 180-tile SimNorm layer: 720 workgroups)
        const int nrowblk_w = (int)(rows_p / 256);
        const long cus_x = cus >= 8 ? cus / 8 : 1;
        const long ks_min = L.knob[LK_W_SPLIT_MIN];
        const int ks_maxp = L.knob[LK_W_SPLIT_MAX];
        const int ks_ovh = L.knob[LK_W_SPLIT_OVH];
        GemmWOrder wo{};
        wo.parts = 1;
        const bool w_shape = can_fuse && ly.CT >= 8 && rows_p % 256 == 0 && rows_per_env % 256 == 0 && ncb256 <= 32 && !L.row_env &&
                             stats_need <= L.stats_cap && w256_min >= 0;
        // L.ksplit (TDMPC2_TUNE_KSPLIT): 0 never; 1 whenever the round arithmetic says so; 2 (default) only for launches that leave
        // most of the chip idle -- 16 .. 128 tiles, i.e. one or two plans of the 317M model (64 tiles -> 256 workgroups of a quarter
        // of K: single-plan latency 17.85 -> 16.2 ms, +9.7 % plans/s at E = 1; the 48M model at E = 4: +6 %; profiles/r5j_*).  On launches that fill the chip the partial sums' traffic (256 KiB per part
        // through the fabric, +210 MB per hidden GEMM of the 48M model at E = 30) costs more than the better fill buys: c3 -10 ... -13 %,
        // c4 -2.7 % (profiles/README.md r5d).
        const long tiles_w = (long)nrowblk_w * ncb256;
        const long ks_auto_lo = L.knob[LK_KSPLIT_AUTO_LO];
        const long ks_auto_min = L.knob[LK_KSPLIT_AUTO_MIN] >= 0 ? L.knob[LK_KSPLIT_AUTO_MIN] : cus / 4;
        const bool ks_want = L.ksplit == 1 || (L.ksplit == 2 && tiles_w >= ks_auto_lo && tiles_w <= cus / 2);
        if (w_shape && ks_want && bufs->ksws && ks_maxp > 1) {
            const int nk = q.K / 16;
            wo = gemm_w_order(nrowblk_w, ncb256, (int)cus_x, nk, ks_maxp, ks_ovh / (nk + 25));
            if (wo.parts > 1 && ((size_t)8 * wo.max_tail * wo.parts > L.ksws_slots || (long)wo.nblk < (L.ksplit == 2 ? ks_auto_min : ks_min))) wo.parts = 1;
        }
        const size_t n_arrive = (size_t)nrowblk_w + (wo.parts > 1 ? (size_t)8 * wo.max_tail : 0);
        if (w_shape && ((long)nrowblk_w * ncb256 >= w256_min || wo.parts > 1) &&
            L.arrive_off + n_arrive <= L.arrive_cap /* out of arrival counters: the narrow tiles below degrade to the unfused LayerNorm */) {
            const int nrowblk = nrowblk_w;
            q.ncolblk = ncb256;
            q.ln_g = ly.g; q.ln_b = ly.b; q.gb_sel_stride = sel ? ln->gb_sel_stride : 0;
            q.ascale = ly.ascale; q.asc_sel_stride = sel ? (long)(3 * sizeof(LayerScal) / sizeof(float)) : 0;
            q.width = ln->width; q.stats = bufs->stats; q.arrive = L.arrive + L.arrive_off; q.err = h->cl_err_dev; q.fault = h->cl_fault;
            L.arrive_off += n_arrive;
            q.out = out; q.KBo = ldo / 16;
            q.ks_parts = wo.parts; q.ks_full = wo.full; q.ks_max_tail = wo.max_tail; q.ks_ws = bufs->ksws; q.ks_cnt = q.arrive + nrowblk;
            // tile order: XCD-local row blocks keep the column blocks of a row block -- which wait for each other -- on consecutive
            // slots of ONE XCD (with one workgroup per CU and <= 16 column blocks two launches in flight cannot starve each other:
            // 2 x 15 waiting workgroups < 32 CUs), and read every A row through one L2
            const int xr_env = L.knob[LK_W_XCD_ROWS];
            // TDMPC2_GEMM_W_XCD_ROWS=2: XCD rectangles, a row block on 2 XCDs -- per XCD and launch A/4 + W/2 instead of A/8 + W
            // through the L2.  At 16 column blocks (317M model): c4 +1 %, fabric-side reads of the launch 881 -> 654 MB, of a stage
            // 27.3 -> 20.6 GB (profiles/README.md r4s, r4t) -- but NOT the default: with a row block on two XCDs two launches in
            // flight can wait for each other in a circle (A's XCD full of launch 1 waiting for B's, B's full of launch 2 waiting
            // for A's); the stress test lost 3 waits in 6 300 stages that way, the XCD-local order none (r4za).
            const int xr_auto = nrowblk >= 16 ? 1 : 0;
            const GemmSOrder ord = gemm_s_order(nrowblk, q.ncolblk, xr_env >= 0 ? xr_env : xr_auto, 1);
            q.xcd_rows = ord.xcd_rows; q.ncol_grid = ord.ncol_grid; q.nrowblk = nrowblk;
            q.timing = L.gw_timing ? L.gw_timing + (q.K >= 1024 ? 8 : 0) + (ln->act ? 16 : 0) : nullptr;  // [Mish K < 1024 | Mish K >= 1024 | SimNorm ...]
            const int epi = 1 + ln->act;
#define GEMM_W_LAUNCH(K) hipLaunchKernelGGL(K, dim3(wo.parts > 1 ? wo.nblk : ord.nblk), dim3(512), 0, st, q)
            if (wo.parts > 1) {
                if (epi == 1) GEMM_W_LAUNCH((g_gemm_w<1, 1>));
                else GEMM_W_LAUNCH((g_gemm_w<2, 1>));
            } else if (epi == 1) GEMM_W_LAUNCH((g_gemm_w<1>));
            else GEMM_W_LAUNCH((g_gemm_w<2>));
#undef GEMM_W_LAUNCH
            LAUNCH_CHECK();
            return 0;
        }
        // wide outputs (>= 256 columns) take the 128 x 256 tile from 128 such workgroups on: with a second chain in flight
        // (lay_estimate_value) a partly filled round is not idle (A/B r3n: threshold 512 -> 256: c3 +2.0 %, c4 +2.3 %; 128: single
        // plans of the 317M model 20.8 -> 18.7 ms; below that single plans of the 48M model lose)
        const size_t wide_min = L.knob[LK_NCT1] ? (size_t)1 << 30 : (size_t)L.knob[LK_WIDE_MIN];
        const bool wide = ly.CT >= 8 && (rows_p / GBM) * ((ly.CT + 7) / 8) >= wide_min;
        q.ncolblk = wide ? (ly.CT + 7) / 8 : (ly.CT + 3) / 4;
        // rows per workgroup tile: 128 when that fills the chip (two workgroups per CU), else 64 or 32 -- few rows mean
        // single-plan latency, where occupancy beats operand reuse (TDMPC2_GEMM_RT=4 forces the 128-row tile)
        const long slots = 2L * cus;
        int rt = 4;
        if (!wide && !L.knob[LK_RT4]) {
            const double fill = L.knob[LK_FILL_PERMILLE] / 1000.0;
            const double fill_head = L.knob[LK_FILL_HEAD_PERMILLE] / 1000.0;
            const double f = ly.CT <= 4 ? fill_head : fill;  // narrow outputs (two-hot / policy heads): one column block
            while (rt > 1 && (double)((long)(rows_p / (32 * rt)) * q.ncolblk) < (double)slots * f) rt >>= 1;
        }
        const int nrowblk = (int)(rows_p / (32 * rt));
        int nblk = nrowblk * q.ncolblk;
        // few workgroups per CU: row operand staged four chunks deep, weight ring of 8 / 16 blocks (g_gemm_s<.., .., 4>)
        const bool deep = !wide && (long)nblk < 2 * slots && !L.knob[LK_SD1];
        int epi = 0;
        const bool fuse = can_fuse && L.arrive_off + (size_t)nrowblk <= L.arrive_cap;
        if (fuse) {
            epi = 1 + ln->act;
            q.ln_g = ly.g; q.ln_b = ly.b; q.gb_sel_stride = sel ? ln->gb_sel_stride : 0;
            q.ascale = ly.ascale; q.asc_sel_stride = sel ? (long)(3 * sizeof(LayerScal) / sizeof(float)) : 0;
            q.width = ln->width; q.stats = bufs->stats; q.arrive = L.arrive + L.arrive_off; q.err = h->cl_err_dev; q.fault = h->cl_fault;
            L.arrive_off += (size_t)nrowblk;
            q.out = out; q.KBo = ldo / 16;
            // tile order (tile_order.h); TDMPC2_GEMM_XCD_ROWS = 0 / 1: never / always XCD-local row blocks, TDMPC2_GEMM_COL_PAD = 0:
            // no padding of the row-major order
            const int xcd_rows_env = L.knob[LK_XCD_ROWS];
            const int col_pad_env = L.knob[LK_COL_PAD];
            const GemmSOrder ord = gemm_s_order(nrowblk, q.ncolblk, xcd_rows_env, col_pad_env);
            q.xcd_rows = ord.xcd_rows; q.ncol_grid = ord.ncol_grid; q.nrowblk = nrowblk;
            nblk = ord.nblk;
        } else if (ln) {  // pre-activations -> PRE, the LayerNorm kernel writes the packed operand
            if (!bufs->PRE) return fail(TDMPC2_ERR_STATE, "no pre-activation buffer on this handle");
            q.out = bufs->PRE; q.ldo = L.ldpre;
        } else if (th && !wide && rt <= 2 && ly.CT <= 4 && (long)nblk >= cus && !L.knob[LK_TWOHOT_UNFUSED]) {
            // two-hot head: the row routine in the epilogue -- from one workgroup per CU on (c3 E = 30: +1.4 %; a single plan's 16-32
            // workgroups are better served by l_twohot's one wavefront per row across the chip: 3.33 vs 3.44 ms, profiles r4i)
            epi = 3;
            q.th = *th;
            if (th_done) *th_done = true;
        } else {
            q.out = out; q.ldo = ldo;
        }
        // The throughput tile stages its row operand TWO chunks ahead (g_gemm_s<2, 4, 2, ..>: same sums; c3 +0.4 ... 0.7 %,
        // c4 +0.5 % over one chunk ahead in three same-call A/Bs, profiles/README.md r3z / r3y / r3x).
        if (wide) GEMM_S_LAUNCH(2, 4, 2);
        else if (deep && rt == 4) GEMM_S_LAUNCH(1, 4, 4);
        else if (deep && rt == 2) GEMM_S_LAUNCH_TH(2, 4);
        else if (deep) GEMM_S_LAUNCH_TH(1, 4);
        else if (rt == 4) GEMM_S_LAUNCH(1, 4, 1);
        else if (rt == 2) GEMM_S_LAUNCH_TH(2, 1);
        else GEMM_S_LAUNCH_TH(1, 1);
        LAUNCH_CHECK();
        if (ln && !fuse)
            return lay_ln(h, st, ln->act, bufs->PRE, L.ldpre, ln->width, rows ? rows : rows_p, rows_per_env, ly, ln->gb_sel_stride, sel, out, ldo);
        return 0;
    }
    GemmParams p{};
    p.A = A; p.lda = lda; p.K = ly.KB * 8; p.wp = ly.wp; p.w_sel_stride = sel ? w_sel_stride : 0;
    p.CT = ly.CT; p.ncolblk = (ly.CT + 3) / 4;
    p.row_env = h->lay.row_env;
    if (slot >= 0 && h->cfg.multitask) {
        p.bias = h->lay.bias_tab + (size_t)slot * h->lay.Mp;
        p.bias_env_stride = (long)h->nnets * h->lay.Mp;
        p.bias_sel_stride = sel ? h->lay.Mp : 0;
    } else {
        p.bias = ly.bias;
        p.bias_env_stride = 0;
        p.bias_sel_stride = sel ? bias_sel_stride : 0;
    }
    p.sel = sel; p.sel_stride = 2; p.rows_per_env = rows_per_env; p.out = out; p.ldo = ldo;
    const int nblocks = (int)(rows_p / GBM) * p.ncolblk;
    hipLaunchKernelGGL(g_gemm, dim3(nblocks), dim3(GTHREADS), 0, st, p);
    LAUNCH_CHECK();
    if (ln) return lay_ln(h, st, ln->act, out, ldo, ln->width, rows ? rows : rows_p, rows_per_env, ly, ln->gb_sel_stride, sel);
    return 0;
}

// x <- ACT(LayerNorm(x)) row by row: exact fp32 -- in place on fp32 rows; split arithmetic -- fp32 rows `x` (a PRE buffer) ->
// the fragment-packed operand buffer `out_packed` of `ld_out` columns.
int lay_ln(tdmpc2_plan *h, hipStream_t st, int act, float *x, int ld, int width, size_t rows, int rows_per_env,
           const HostLayer &ly, long gb_sel_stride, const int *sel, float *out_packed, int ld_out) {
    LnActParams p{};
    p.x = x; p.ld = ld; p.width = width; p.rows = (int)rows; p.rows_per_env = rows_per_env;
    p.g = ly.g; p.b = ly.b; p.gb_sel_stride = sel ? gb_sel_stride : 0; p.sel = sel; p.sel_stride = 2;
    p.out = reinterpret_cast<char *>(out_packed); p.KBo = ld_out / 16;
    p.ascale = ly.ascale; p.asc_sel_stride = sel ? (long)(3 * sizeof(LayerScal) / sizeof(float)) : 0;
    const int grid = (int)((rows + RW_THREADS / 64 - 1) / (RW_THREADS / 64));
    if (h->split) {
        if (act == 0) hipLaunchKernelGGL(l_ln_act_s<0>, dim3(grid), dim3(RW_THREADS), 0, st, p);
        else hipLaunchKernelGGL(l_ln_act_s<1>, dim3(grid), dim3(RW_THREADS), 0, st, p);
    } else {
        if (act == 0) hipLaunchKernelGGL(l_ln_act<0>, dim3(grid), dim3(RW_THREADS), 0, st, p);
        else hipLaunchKernelGGL(l_ln_act<1>, dim3(grid), dim3(RW_THREADS), 0, st, p);
    }
    LAUNCH_CHECK();
    return 0;
}

// strides between consecutive Q heads of layer `l` (slab allocation in bind_weights)
// (h->lay.qarr: the online ensemble h->q, or the target ensemble h->tq inside lay_value)
inline long q_wstride(const tdmpc2_plan *h, int l) {
    if (h->cfg.num_q < 2) return 0;
    const HostNet *q = h->lay.qarr;
    return h->split ? (long)(q[1].l[l].wps - q[0].l[l].wps) : (long)(q[1].l[l].wp - q[0].l[l].wp);
}
inline long q_bstride(const tdmpc2_plan *h, int l) { return h->cfg.num_q > 1 ? (long)(h->lay.qarr[1].l[l].bias - h->lay.qarr[0].l[l].bias) : 0; }
inline long q_gstride(const tdmpc2_plan *h, int l) { return h->cfg.num_q > 1 ? (long)(h->lay.qarr[1].l[l].g - h->lay.qarr[0].l[l].g) : 0; }

// X -> hidden 1 (HA) -> hidden 2 (HB): the two NormedLinear(Mish) layers of a reference `mlp` (layers.py:121-133).
// `after_l0` (optional): recorded on `st` once the first layer -- the only reader of X in the chain -- has been launched.
int lay_hidden(tdmpc2_plan *h, hipStream_t st, const HostNet &net, int slot, size_t rows, size_t rows_p, int rpe,
               const int *sel, bool is_q, const LayBufs *bufs = nullptr, hipEvent_t after_l0 = nullptr,
               const GemmRange *l0_range = nullptr) {
    const Layered &L = h->lay;
    const LayBufs b = bufs ? *bufs : lay_bufs(h, 0);
    int rc;
    const LnFuse f0{0, is_q ? q_gstride(h, 0) : 0, h->cfg.mlp_dim}, f1{0, is_q ? q_gstride(h, 1) : 0, h->cfg.mlp_dim};
    if ((rc = lay_gemm(h, st, L.X, L.Kin, rows_p, rpe, net.l[0], is_q ? q_wstride(h, 0) : 0, is_q ? q_bstride(h, 0) : 0, slot,
                       sel, b.HA, L.Mp, &f0, &b, l0_range, rows))) return rc;
    if (after_l0) HIP_TRY(hipEventRecord(after_l0, st));
    return lay_gemm(h, st, b.HA, L.Mp, rows_p, rpe, net.l[1], is_q ? q_wstride(h, 1) : 0, is_q ? q_bstride(h, 1) : 0, -1,
                    sel, b.HB, L.Mp, &f1, &b, nullptr, rows);
}

// z <- next(z, a): dynamics MLP with SimNorm output written back into X[:, 0:L)  (world_model.py:114-121)
// `x_free` (optional): an event of another stream after which X may be overwritten (a concurrent chain still reading z_t).
int lay_dynamics(tdmpc2_plan *h, hipStream_t st, size_t rows, size_t rows_p, int rpe, hipEvent_t x_free = nullptr,
                 const GemmRange *l0_range = nullptr) {
    const Layered &L = h->lay;
    int rc;
    if ((rc = lay_hidden(h, st, h->dyn, BE_DYN, rows, rows_p, rpe, nullptr, false, nullptr, nullptr, l0_range))) return rc;
    if (x_free) HIP_TRY(hipStreamWaitEvent(st, x_free, 0));
    const LnFuse f2{1, 0, h->cfg.latent_dim};
    // the SimNorm latent goes straight into X's z columns in operand form (action / padding columns untouched)
    return lay_gemm(h, st, L.HB, L.Mp, rows_p, rpe, h->dyn.l[2], 0, 0, -1, nullptr, L.X, L.Kin, &f2, nullptr, nullptr, rows);
}

// a <- pi(z) into X[:, L:L+A) (+ actions[e, t, n < P] for the policy-prior trajectories)  (world_model.py:144-184)
int lay_policy(tdmpc2_plan *h, hipStream_t st, size_t rows, size_t rows_p, int rpe, int nvalid, const float *mask,
               const float *eps, long eps_estride, unsigned long long seed, unsigned call, int site, int iter,
               float *actions, int t, float *trace = nullptr, int n_off = 0) {
    const Layered &L = h->lay;
    const tdmpc2_plan_cfg &c = h->cfg;
    int rc;
    if ((rc = lay_hidden(h, st, h->pi, BE_PI, rows, rows_p, rpe, nullptr, false))) return rc;
    if ((rc = lay_gemm(h, st, L.HB, L.Mp, rows_p, rpe, h->pi.l[2], 0, 0, -1, nullptr, L.LG, L.ldl))) return rc;
    PiHeadParams p{};
    p.lg = L.LG; p.ld = L.ldl; p.rows = (int)rows; p.rows_per_env = rpe; p.nvalid = nvalid; p.A = c.action_dim;
    p.L = c.latent_dim; p.ldx = L.Kin; p.lsmin = c.log_std_min; p.lsdif = c.log_std_dif; p.mask = mask;
    p.eps = eps; p.eps_estride = eps_estride; p.seed = seed; p.call = call; p.site = site; p.iter = iter;
    p.X = L.X; p.actions = actions; p.t = t; p.H = c.horizon; p.N = c.num_samples; p.trace = trace;
    p.row_env = L.row_env; p.n_off = n_off;
    if (L.row_env) { p.H = 1; p.N = (int)rows; }  // value mode: actions is a flat [rows, A] output
    const int total = (int)rows * c.action_dim;
    if (h->split) hipLaunchKernelGGL(l_pi_head_s, dim3((total + 255) / 256), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(l_pi_head, dim3((total + 255) / 256), dim3(256), 0, st, p);
    LAUNCH_CHECK();
    return 0;
}

TwoHotParams lay_twohot_params(tdmpc2_plan *h, size_t rows, int rpe, int mode, int t, const float *disc_pow, float *value,
                               float *trace, int n_full, int n_off, const float *lg) {
    const Layered &L = h->lay;
    TwoHotParams p{};
    p.lg = lg ? lg : L.LG; p.ld = L.ldl; p.rows = (int)rows; p.rows_per_env = rpe; p.num_bins = h->cfg.num_bins; p.mode = mode;
    p.t = t; p.H = h->cfg.horizon; p.bins = h->bins; p.disc_pow = disc_pow; p.G = L.G; p.qtmp = L.QT; p.value = value;
    p.term = h->cfg.episodic ? L.TERM : nullptr; p.trace = trace; p.trace_ld = h->cfg.horizon + 2 + h->cfg.action_dim;
    p.n_full = n_full; p.n_off = n_off;
    return p;
}

int lay_twohot(tdmpc2_plan *h, hipStream_t st, size_t rows, int rpe, int mode, int t, const float *disc_pow, float *value,
               float *trace, int n_full = 0, int n_off = 0, const float *lg = nullptr) {
    const TwoHotParams p = lay_twohot_params(h, rows, rpe, mode, t, disc_pow, value, trace, n_full, n_off, lg);
    const int grid = (int)((rows + RW_THREADS / 64 - 1) / (RW_THREADS / 64));
    hipLaunchKernelGGL(l_twohot, dim3(grid), dim3(RW_THREADS), 0, st, p);
    LAUNCH_CHECK();
    return 0;
}

// A two-hot head (reward / Q: world_model.py:123-130,186-216 -> math.two_hot_inv): the output nn.Linear over the hidden
// activations `A` and the row routine -- inside the GEMM's epilogue on the narrow tiles (split arithmetic), else GEMM + l_twohot.
int lay_head_twohot(tdmpc2_plan *h, hipStream_t st, const float *A, size_t rows, size_t rows_p, int rpe, const HostLayer &ly,
                    long w_sel_stride, long bias_sel_stride, const int *sel, float *lg, int mode, int t, const float *disc_pow,
                    float *value, float *trace, int n_full = 0, int n_off = 0) {
    const Layered &L = h->lay;
    const TwoHotParams th = lay_twohot_params(h, rows, rpe, mode, t, disc_pow, value, trace, n_full, n_off, lg);
    bool done = false;
    int rc;
    if ((rc = lay_gemm(h, st, A, L.Mp, rows_p, rpe, ly, w_sel_stride, bias_sel_stride, -1, sel, lg, L.ldl, nullptr, nullptr, nullptr, 0,
                       &th, &done))) return rc;
    if (done) return 0;
    return lay_twohot(h, st, rows, rpe, mode, t, disc_pow, value, trace, n_full, n_off, lg);
}

int lay_setup(tdmpc2_plan *h, hipStream_t st, int E, const float *task_emb, const float *prev_mean, const unsigned char *t0,
              bool init_dist, float *beff_out, const HostNet *qarr) {
    const tdmpc2_plan_cfg &c = h->cfg;
    if (!c.multitask && !init_dist) return 0;
    if (!qarr) qarr = h->q;
    LSetupParams p{};
    p.E = E; p.H = c.horizon; p.A = c.action_dim; p.T = c.task_dim; p.M = c.mlp_dim; p.Mp = h->lay.Mp; p.nnets = h->nnets;
    p.multitask = c.multitask; p.max_std = c.max_std;
    p.bias[BE_DYN] = h->dyn.l[0].bias; p.wemb[BE_DYN] = h->dyn.l[0].wemb;
    p.bias[BE_REW] = h->rew.l[0].bias; p.wemb[BE_REW] = h->rew.l[0].wemb;
    p.bias[BE_PI] = h->pi.l[0].bias; p.wemb[BE_PI] = h->pi.l[0].wemb;
    for (int i = 0; i < c.num_q; ++i) { p.bias[BE_Q0 + i] = qarr[i].l[0].bias; p.wemb[BE_Q0 + i] = qarr[i].l[0].wemb; }
    p.task_emb = task_emb; p.prev_mean = prev_mean; p.t0 = t0; p.beff = beff_out ? beff_out : h->beff;
    p.mean = init_dist ? h->mean : nullptr; p.std = h->std;
    // few plans: one 256-column chunk per block (a single plan: nnets x Mp / 256 blocks instead of nnets); many plans already fill the chip
    const int chunks = c.multitask ? std::max(1, std::min((h->lay.Mp + 255) / 256, 512 / std::max(1, E * h->nnets))) : 1;
    hipLaunchKernelGGL(l_setup, dim3(E, c.multitask ? h->nnets : 1, chunks), dim3(256), 0, st, p);
    LAUNCH_CHECK();
    return 0;
}

// grid of l_init_x_s: one row of blocks per 32-row tile x column chunks, so that calls of a few tiles (single plans, the policy-prior
// rows) are not one workgroup looping over a whole tile
inline dim3 init_x_grid(size_t rows, int ldx) {
    const unsigned tiles = (unsigned)((rows + 31) / 32);
    const unsigned passes = (unsigned)((ldx / 8 * 32 + 255) / 256);
    return dim3(tiles, std::max(1u, std::min(passes, 256u / std::max(1u, tiles))));
}

// At t = 0 every sample row of a plan starts from the same latent (z.repeat(num_samples, 1), tdmpc2.py:163): the z columns'
// product of the reward / dynamics first layers is ONE vector per plan, cvec[net][e] = W[:, :L] z0_e + b_eff(e), computed here
// once per plan (E rows instead of E x N); the t = 0 GEMMs of every CEM iteration then contract the action columns only
// (K = 32 instead of L + 32).  Split arithmetic; same products, the z / action partial sums are added in fp32.
int lay_cvec(tdmpc2_plan *h, hipStream_t st, int E, const float *z0) {
    Layered &L = h->lay;
    L.cvec_ready = false;
    if (!h->split || !L.Z0X || L.knob[LK_Z0_SHARED_OFF]) return 0;
    const tdmpc2_plan_cfg &c = h->cfg;
    const size_t rows_p = round_up((size_t)E, GBM);
    hipLaunchKernelGGL(l_init_x_s, init_x_grid((size_t)E, L.Kin), dim3(256), 0, st, L.Z0X, L.Kin, c.latent_dim, 1, z0, (float *)nullptr, (float *)nullptr, E);
    LAUNCH_CHECK();
    const HostNet *nets[2] = {&h->rew, &h->dyn};
    const int slots[2] = {BE_REW, BE_DYN};
    int rc;
    for (int i = 0; i < 2; ++i) {
        GemmRange r{0, c.latent_dim / 16, nullptr, 0, 0};
        if ((rc = lay_gemm(h, st, L.Z0X, L.Kin, rows_p, 1, nets[i]->l[0], 0, 0, slots[i], nullptr, L.cvec + (size_t)i * L.cvec_rows * L.Mp, L.Mp,
                           nullptr, nullptr, &r))) return rc;
    }
    L.cvec_ready = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The few-row path (layered_mid.cuh): every nn.Linear = one g_gemm_m problem (K-parts of 64 x 256 tiles, raw fp32 partial sums) +
// one m_rows problem (sum the parts, scale, bias, then LayerNorm + Mish / SimNorm + split, two-hot, policy head or termination);
// a launch of either kernel carries up to two problems -- the two chains of a step, the two Q heads -- on ONE stream.
struct MidOp {
    // the nn.Linear
    const float *A;          // fragment-packed operand buffer
    int lda;
    const HostLayer *ly;
    int slot;                // index of the net in beff (multitask first layers), -1 otherwise
    const int *sel;          // Q head of each plan, or null
    bool is_q;
    int layer;               // 0, 1, 2 (strides of the Q ensemble)
    const GemmRange *range;  // a k-range with a per-environment bias (t = 0: the action columns, lay_cvec), or null
    // what follows it
    int kind;                // MR_*
    float *out;              // MR_LN_*: packed operand buffer of ldo columns
    int ldo, width;
    const float *actions;    // MR_LN_SIMNORM into X: also write the action columns of step act_t (null: no)
    int act_t, act_nsub, act_noff;
    TwoHotParams th;
    PiHeadParams pi;
};

inline bool mid_ok(const tdmpc2_plan *h, size_t rows_p) {
    const Layered &L = h->lay;
    if (!h->split || !L.mid || L.ksplit == 0 || !L.mws[0] || !L.HA2 || L.row_env) return false;
    const long cus = h->num_cus > 0 ? h->num_cus : 256;
    const long maxct = (std::max(h->cfg.mlp_dim, h->cfg.latent_dim) + 31) / 32;
    return 2 * (long)(rows_p / GM_TM) * ((maxct + 7) / 8) <= cus;  // a launch's two widest problems fill at most one round of the chip
}

// one launch of g_gemm_m + one of m_rows for n (1 or 2) layers over the same `rows` sample rows
int mid_stage(tdmpc2_plan *h, hipStream_t st, size_t rows, size_t rows_p, int rpe, const MidOp *ops, int n, bool rows_one_by_one = false) {
    Layered &L = h->lay;
    const long cus = h->num_cus > 0 ? h->num_cus : 256;
    GemmMParams G{};
    MRowParams R{};
    G.nprob = R.nprob = n;
    long tiles = 0;
    for (int i = 0; i < n; ++i) tiles += (long)(rows_p / GM_TM) * ((ops[i].ly->CT + 7) / 8);
    const int pmax = L.knob[LK_MID_PARTS_MAX];
    const int pall = (int)std::max<long>(1, std::min<long>(pmax, cus / std::max<long>(tiles, 1)));
    unsigned gblk = 0, rblk = 0;
    int nrow = 0;
    for (int i = 0; i < n; ++i) {
        const MidOp &o = ops[i];
        const HostLayer &ly = *o.ly;
        const bool sel = o.sel != nullptr;
        GemmMProb &g = G.pr[i];
        g.A = reinterpret_cast<const _Float16 *>(o.A); g.KBa = o.lda / 16;
        g.a_kb0 = o.range ? o.range->col_off / 16 : 0;
        g.nk = (o.range ? o.range->kblocks : ly.KB) / 2;  // k32-slabs
        g.kb0 = o.range ? o.range->kb0 : 0; g.kbs = ly.KB;
        g.wp = ly.wps; g.w_sel_stride = sel && o.is_q ? q_wstride(h, o.layer) : 0;
        g.CT = ly.CT; g.ncolblk = (ly.CT + 7) / 8; g.nrowblk = (int)(rows_p / GM_TM);
        g.parts = std::max(1, std::min(pall, g.nk / 2));
        g.sel = o.sel; g.sel_stride = 2; g.rows_per_env = rpe;
        g.ldw = g.ncolblk * 256; g.part_stride = (long)rows_p * g.ldw;
        if ((size_t)g.parts * g.part_stride > L.mws_cap) return fail(TDMPC2_ERR_STATE, "few-row path: partial-sum workspace too small");
        g.ws = L.mws[i];
        g.nblk = 8 * ((g.ncolblk * g.parts + 7) / 8) * g.nrowblk;
        g.rows = (int)rows;
        gblk += (unsigned)g.nblk;
        // the row-side description of the layer (m_rows, or the GEMM's own epilogue when the tile is whole: parts == 1)
        MRowProb r{};
        r.kind = o.kind; r.rows = (int)rows; r.rows_per_env = rpe;
        if ((int)rows < MR_WIDE_MIN) r.nwg = (o.kind == MR_LN_MISH || o.kind == MR_LN_SIMNORM) ? (int)rows : (int)((rows + 3) / 4);  // m_rows<256>
        else r.nwg = (int)((rows + MR_R - 1) / MR_R);                                                                            // m_rows<512>
        r.ws = g.ws; r.part_stride = g.part_stride; r.ldw = g.ldw; r.parts = g.parts;
        r.oscale = ly.oscale; r.osc_sel_stride = sel ? (long)(3 * sizeof(LayerScal) / sizeof(float)) : 0;
        if (o.slot >= 0 && h->cfg.multitask) {
            r.bias = L.bias_tab + (size_t)o.slot * L.Mp; r.bias_env_stride = (long)h->nnets * L.Mp; r.bias_sel_stride = sel ? L.Mp : 0;
        } else {
            r.bias = ly.bias; r.bias_env_stride = 0; r.bias_sel_stride = sel && o.is_q ? q_bstride(h, o.layer) : 0;
        }
        if (o.range && o.range->bias_env) { r.bias = o.range->bias_env; r.bias_env_stride = o.range->bias_env_stride; r.bias_sel_stride = 0; }
        r.sel = o.sel; r.sel_stride = 2; r.row_env = nullptr;
        const bool ln = o.kind == MR_LN_MISH || o.kind == MR_LN_SIMNORM;
        if (ln) {
            r.width = o.width; r.g = ly.g; r.b = ly.b; r.gb_sel_stride = sel && o.is_q ? q_gstride(h, o.layer) : 0;
            r.ascale = ly.ascale; r.asc_sel_stride = sel ? (long)(3 * sizeof(LayerScal) / sizeof(float)) : 0;
            r.out = reinterpret_cast<char *>(o.out); r.KBo = o.ldo / 16;
            r.actions = o.actions; r.act_t = o.act_t; r.act_A = h->cfg.action_dim; r.act_L = h->cfg.latent_dim; r.act_ldx = L.Kin;
            r.act_N = h->cfg.num_samples; r.act_H = h->cfg.horizon; r.act_nsub = o.act_nsub; r.act_noff = o.act_noff;
        }
        r.term = L.TERM;
        r.th = o.th; r.pi = o.pi;
        // whole-K tiles of a NormedLinear: LayerNorm + activation + split in the GEMM's own epilogue (no partial sums, no m_rows) -- the
        // 317M model's hidden layers (2 x 128 tiles = the chip), the 48M model from four plans on.  Needs the exchange buffers and the
        // handle's consent (L.fuse_ln: off after a reported wait, apply_modes).  Rows that also carry the next step's actions stay with m_rows.
        float *stats = i == 0 ? L.stats : L.stats2;
        const size_t stats_need = rows_p * ((ly.CT + 3) / 4) * 2;
        if (ln && g.parts == 1 && L.fuse_ln && L.knob[LK_MID_FUSE_LN] && !o.actions && stats && L.arrive && stats_need <= L.stats_cap &&
            (L.arrive_pending ? 0 : L.arrive_off) + (size_t)g.nrowblk <= L.arrive_cap && rpe % GM_TM == 0) {
            if (L.arrive_pending) {
                L.arrive_pending = false;
                int rc0 = lay_arrive_reset(h, st);
                if (rc0) return rc0;
            }
            g.epi = o.kind == MR_LN_MISH ? 1 : 2;
            g.oscale = r.oscale; g.osc_sel_stride = r.osc_sel_stride;
            g.bias = r.bias; g.bias_env_stride = r.bias_env_stride; g.bias_sel_stride = r.bias_sel_stride;
            g.ln_g = r.g; g.ln_b = r.b; g.gb_sel_stride = r.gb_sel_stride; g.ascale = r.ascale; g.asc_sel_stride = r.asc_sel_stride;
            g.width = o.width; g.stats = stats; g.arrive = L.arrive + L.arrive_off; g.err = h->cl_err_dev; g.fault = h->cl_fault;
            g.out = o.out; g.KBo = o.ldo / 16;
            L.arrive_off += (size_t)g.nrowblk;
        } else {
            R.pr[nrow++] = r;
            rblk += (unsigned)r.nwg;
        }
    }
    R.nprob = nrow;
    // two problems of one shape: one per half of the XCDs (g_gemm_m); each problem's share of the grid is 4 x ceil(jobs / 4) x row blocks
    if (n == 2 && L.knob[LK_MID_SPLIT_XCD] && G.pr[0].ncolblk * G.pr[0].parts == G.pr[1].ncolblk * G.pr[1].parts && G.pr[0].nrowblk == G.pr[1].nrowblk) {
        const int jobs = G.pr[0].ncolblk * G.pr[0].parts;
        G.split_xcd = 1;
        gblk = 8u * (unsigned)((jobs + 3) / 4) * (unsigned)G.pr[0].nrowblk;
        G.pr[0].nblk = G.pr[1].nblk = (int)gblk / 2;
    }
    hipLaunchKernelGGL(g_gemm_m, dim3(gblk), dim3(512), 0, st, G);
    LAUNCH_CHECK();
    const bool small = (int)rows < MR_WIDE_MIN;
    if (nrow == 0) return 0;  // both layers' epilogues ran inside the GEMM
    if (rows_one_by_one && nrow == 2) {  // the second problem's rows read what the first one's wrote (the two Q heads' two-hots, qtmp):
        R.serial = 1;                    // one grid over problem 0's rows, every wavefront does its row of both problems in order
        rblk = (unsigned)R.pr[0].nwg;
    }
    if (small) hipLaunchKernelGGL(m_rows<256>, dim3(rblk), dim3(256), 0, st, R);
    else hipLaunchKernelGGL(m_rows<512>, dim3(rblk), dim3(512), 0, st, R);
    LAUNCH_CHECK();
    return 0;
}

inline MidOp mid_hidden(const Layered &L, const float *A, int lda, const HostNet &net, int layer, int slot, const int *sel, bool is_q,
                        float *out, int mlp_dim, const GemmRange *range = nullptr) {
    MidOp o{};
    o.A = A; o.lda = lda; o.ly = &net.l[layer]; o.slot = slot; o.sel = sel; o.is_q = is_q; o.layer = layer; o.range = range;
    o.kind = MR_LN_MISH; o.out = out; o.ldo = L.Mp; o.width = mlp_dim;
    return o;
}
inline MidOp mid_head(const Layered &L, const float *A, const HostNet &net, const int *sel, bool is_q, int kind) {
    MidOp o{};
    o.A = A; o.lda = L.Mp; o.ly = &net.l[2]; o.slot = -1; o.sel = sel; o.is_q = is_q; o.layer = 2; o.kind = kind;
    return o;
}

int lay_estimate_value_m(tdmpc2_plan *h, hipStream_t st, int E, const float *z0, const float *act_mask, const float *disc_pow,
                         const float *actions, const float *pi_eps, long pi_eps_estride, const int *qidx, unsigned long long seed,
                         unsigned call, int iter, float *value, float *trace, int n_off, int n_sub) {
    const tdmpc2_plan_cfg &c = h->cfg;
    const Layered &L = h->lay;
    const int NF = c.num_samples, H = c.horizon, A = c.action_dim;
    const int N = n_sub > 0 ? n_sub : NF;
    const bool ranged = N != NF;
    const size_t rows = (size_t)E * N, rows_p = round_up(rows, GBM);
    int rc;
    h->lay.arrive_pending = L.fuse_ln;  // (whole-K tiles run the NormedLinear epilogue inside g_gemm_m: mid_stage zeroes the stage's arrival counters
                                   // in front of the first launch that needs them -- a single 48M plan has none and pays no memset)
    const bool pifold = L.pifold && E == 1 && !ranged && c.num_pi_trajs > 0;
    if (L.cvec_ready) {
        if (c.episodic) HIP_TRY(hipMemsetAsync(L.TERM, 0, rows * sizeof(float), st));
        if (pifold) {  // the policy head of step 0 reads z_0 from the z columns of its rows (nothing else of this stage reads them: lay_cvec)
            hipLaunchKernelGGL(l_init_x_s, init_x_grid((size_t)L.Ppad, L.Kin), dim3(256), 0, st, L.X, L.Kin, c.latent_dim, L.Ppad, z0, (float *)nullptr,
                               (float *)nullptr, L.Ppad);
            LAUNCH_CHECK();
        }
    } else {
        hipLaunchKernelGGL(l_init_x_s, init_x_grid(rows, L.Kin), dim3(256), 0, st, L.X, L.Kin, c.latent_dim, N, z0, L.G, L.TERM, (int)rows);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(l_set_action_s, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, st, L.X, L.Kin, c.latent_dim, A, NF, H, 0, (int)rows, actions, N, n_off);
    LAUNCH_CHECK();
    for (int t = 0; t < H; ++t) {
        if (pifold) {  // a_t = pi(z_t) for the P policy-prior rows (tdmpc2.py:156-158): rows [0, P) of X, into their action columns and `actions`
            const int P = c.num_pi_trajs;
            const size_t prow = (size_t)P, prow_p = round_up(prow, GBM);
            MidOp o0 = mid_hidden(L, L.X, L.Kin, h->pi, 0, BE_PI, nullptr, false, L.HA, c.mlp_dim);
            if ((rc = mid_stage(h, st, prow, prow_p, L.Ppad, &o0, 1))) return rc;
            MidOp o1 = mid_hidden(L, L.HA, L.Mp, h->pi, 1, -1, nullptr, false, L.HB, c.mlp_dim);
            if ((rc = mid_stage(h, st, prow, prow_p, L.Ppad, &o1, 1))) return rc;
            MidOp o2 = mid_head(L, L.HB, h->pi, nullptr, false, MR_PI);
            PiHeadParams &p = o2.pi;
            p.rows = P; p.rows_per_env = L.Ppad; p.nvalid = P; p.A = A; p.L = c.latent_dim; p.ldx = L.Kin;
            p.lsmin = c.log_std_min; p.lsdif = c.log_std_dif; p.mask = act_mask;
            p.eps = L.pifold_eps ? L.pifold_eps + (size_t)t * P * A : nullptr; p.eps_estride = (long)H * P * A;  // tape layout [E, H, P, A]
            p.seed = seed; p.call = call; p.site = SITE_PITRAJ; p.iter = t; p.X = L.X; p.actions = h->actions; p.t = t; p.H = H;
            p.N = c.num_samples; p.trace = nullptr; p.row_env = nullptr; p.n_off = 0;
            if ((rc = mid_stage(h, st, prow, prow_p, L.Ppad, &o2, 1))) return rc;
        }
        GemmRange r_rew{c.latent_dim / 16, (L.Kin - c.latent_dim) / 16, L.cvec, (long)L.Mp, c.latent_dim};
        GemmRange r_dyn = r_rew;
        r_dyn.bias_env = L.cvec + (size_t)L.cvec_rows * L.Mp;
        const bool shortk = t == 0 && L.cvec_ready;
        {   // [dynamics.l0 | reward.l0]: X -> HA, HA2
            MidOp ops[2] = {mid_hidden(L, L.X, L.Kin, h->dyn, 0, BE_DYN, nullptr, false, L.HA, c.mlp_dim, shortk ? &r_dyn : nullptr),
                            mid_hidden(L, L.X, L.Kin, h->rew, 0, BE_REW, nullptr, false, L.HA2, c.mlp_dim, shortk ? &r_rew : nullptr)};
            if ((rc = mid_stage(h, st, rows, rows_p, N, ops, 2))) return rc;
        }
        {   // [dynamics.l1 | reward.l1]
            MidOp ops[2] = {mid_hidden(L, L.HA, L.Mp, h->dyn, 1, -1, nullptr, false, L.HB, c.mlp_dim),
                            mid_hidden(L, L.HA2, L.Mp, h->rew, 1, -1, nullptr, false, L.HB2, c.mlp_dim)};
            if ((rc = mid_stage(h, st, rows, rows_p, N, ops, 2))) return rc;
        }
        {   // [dynamics.l2 -> SimNorm -> z_{t+1} into X (+ a_{t+1}) | reward head -> two_hot_inv -> G += disc (1 - term) r]
            MidOp ops[2] = {mid_head(L, L.HB, h->dyn, nullptr, false, MR_LN_SIMNORM), mid_head(L, L.HB2, h->rew, nullptr, false, MR_TWOHOT)};
            ops[0].out = L.X; ops[0].ldo = L.Kin; ops[0].width = c.latent_dim;
            if (t + 1 < H) { ops[0].actions = actions; ops[0].act_t = t + 1; ops[0].act_nsub = N; ops[0].act_noff = n_off; }
            ops[1].th = lay_twohot_params(h, rows, N, 0, t, disc_pow, value, trace, 0, 0, nullptr);
            if ((rc = mid_stage(h, st, rows, rows_p, N, ops, 2))) return rc;
        }
        if (c.episodic) {  // termination head on the new latent (tdmpc2.py:133-134)
            MidOp o0 = mid_hidden(L, L.X, L.Kin, h->term, 0, -1, nullptr, false, L.HA, c.mlp_dim);
            if ((rc = mid_stage(h, st, rows, rows_p, N, &o0, 1))) return rc;
            MidOp o1 = mid_hidden(L, L.HA, L.Mp, h->term, 1, -1, nullptr, false, L.HB, c.mlp_dim);
            if ((rc = mid_stage(h, st, rows, rows_p, N, &o1, 1))) return rc;
            MidOp o2 = mid_head(L, L.HB, h->term, nullptr, false, MR_TERM);
            if ((rc = mid_stage(h, st, rows, rows_p, N, &o2, 1))) return rc;
        }
    }
    // a_H = pi(z_H)
    {
        MidOp o0 = mid_hidden(L, L.X, L.Kin, h->pi, 0, BE_PI, nullptr, false, L.HA, c.mlp_dim);
        if ((rc = mid_stage(h, st, rows, rows_p, N, &o0, 1))) return rc;
        MidOp o1 = mid_hidden(L, L.HA, L.Mp, h->pi, 1, -1, nullptr, false, L.HB, c.mlp_dim);
        if ((rc = mid_stage(h, st, rows, rows_p, N, &o1, 1))) return rc;
        MidOp o2 = mid_head(L, L.HB, h->pi, nullptr, false, MR_PI);
        PiHeadParams &p = o2.pi;
        p.rows = (int)rows; p.rows_per_env = N; p.nvalid = N; p.A = A; p.L = c.latent_dim; p.ldx = L.Kin;
        p.lsmin = c.log_std_min; p.lsdif = c.log_std_dif; p.mask = act_mask; p.eps = pi_eps; p.eps_estride = pi_eps_estride;
        p.seed = seed; p.call = call; p.site = SITE_PI; p.iter = iter; p.X = L.X; p.actions = nullptr; p.t = 0; p.H = H; p.N = NF;
        p.trace = trace; p.row_env = nullptr; p.n_off = n_off;
        if ((rc = mid_stage(h, st, rows, rows_p, N, &o2, 1))) return rc;
    }
    // value = G + disc^H (1 - term) avg of the two selected Q heads
    {
        MidOp ops[2] = {mid_hidden(L, L.X, L.Kin, h->q[0], 0, BE_Q0, qidx, true, L.HA, c.mlp_dim),
                        mid_hidden(L, L.X, L.Kin, h->q[0], 0, BE_Q0, qidx + 1, true, L.HA2, c.mlp_dim)};
        if ((rc = mid_stage(h, st, rows, rows_p, N, ops, 2))) return rc;
    }
    {
        MidOp ops[2] = {mid_hidden(L, L.HA, L.Mp, h->q[0], 1, -1, qidx, true, L.HB, c.mlp_dim),
                        mid_hidden(L, L.HA2, L.Mp, h->q[0], 1, -1, qidx + 1, true, L.HB2, c.mlp_dim)};
        if ((rc = mid_stage(h, st, rows, rows_p, N, ops, 2))) return rc;
    }
    {   // the heads' GEMMs as one launch; their two-hots one behind the other (the second reads the first's value)
        MidOp ops[2] = {mid_head(L, L.HB, h->q[0], qidx, true, MR_TWOHOT), mid_head(L, L.HB2, h->q[0], qidx + 1, true, MR_TWOHOT)};
        ops[0].th = lay_twohot_params(h, rows, N, 1, 0, disc_pow, value, trace, ranged ? NF : 0, n_off, nullptr);
        ops[1].th = lay_twohot_params(h, rows, N, 2, 0, disc_pow, value, trace, ranged ? NF : 0, n_off, nullptr);
        if ((rc = mid_stage(h, st, rows, rows_p, N, ops, 2, /*rows_one_by_one=*/true))) return rc;
    }
    return 0;
}

// TDMPC2._estimate_value (tdmpc2/tdmpc2.py:122-136) for E plans with the step actions in `actions` [E,H,N,A].
int lay_estimate_value(tdmpc2_plan *h, hipStream_t st, int E, const float *z0, const float *act_mask, const float *disc_pow,
                       const float *actions, const float *pi_eps, long pi_eps_estride, const int *qidx /* dense [E,2] */,
                       unsigned long long seed, unsigned call, int iter, float *value, float *trace, int n_off, int n_sub) {
    // rows n_off .. n_off + n_sub of every plan (default: all num_samples rows): the sample-row range of one rank when a
    // plan is sharded over GPUs (tdmpc2_plan_shard_values); n_sub is a multiple of the GEMM row tile
    const tdmpc2_plan_cfg &c = h->cfg;
    const Layered &L = h->lay;
    const int NF = c.num_samples, H = c.horizon, A = c.action_dim;
    const int N = n_sub > 0 ? n_sub : NF;
    const bool ranged = N != NF;
    const size_t rows = (size_t)E * N, rows_p = round_up(rows, GBM);
    int rc;
    if (mid_ok(h, rows_p))  // few rows (single plans): K-part tiles + row kernels on one stream (layered_mid.cuh)
        return lay_estimate_value_m(h, st, E, z0, act_mask, disc_pow, actions, pi_eps, pi_eps_estride, qidx, seed, call, iter, value, trace, n_off, n_sub);
    if (L.pifold)  // lay_run decided with the same predicate that this call computes the policy-prior rows' actions: never reached
        return fail(TDMPC2_ERR_STATE, "policy-prior rows were left to a stage that does not compute them");
    if ((rc = lay_arrive_reset(h, st))) return rc;
    // X <- [z0 | .] for every sample row, G <- 0, TERM <- 0.  With the shared z0 products (lay_cvec) nothing reads the z columns
    // before the first dynamics step has written z_1 there (the t = 0 GEMMs contract the action columns only), G is
    // overwritten at t = 0 and the action / padding columns are set by every step: only TERM is left to clear.
    if (h->split && L.cvec_ready) {
        if (c.episodic) HIP_TRY(hipMemsetAsync(L.TERM, 0, rows * sizeof(float), st));
    } else {
        if (h->split) hipLaunchKernelGGL(l_init_x_s, init_x_grid(rows, L.Kin), dim3(256), 0, st, L.X, L.Kin, c.latent_dim, N, z0, L.G, L.TERM, (int)rows);
        else hipLaunchKernelGGL(l_init_x, dim3((unsigned)rows), dim3(256), 0, st, L.X, L.Kin, c.latent_dim, N, z0, L.G, L.TERM);
        LAUNCH_CHECK();
    }
    // Two chains at a time (h->lay.side: a second stream + a second buffer set): the reward chain of step t runs beside the
    // dynamics chain, the second Q head beside the first.  Hazards: reward.l0 is the side chain's only reader of X -- the
    // dynamics' last layer (the writer of z_{t+1}) waits for it; the termination update waits for the reward's two-hot (which
    // reads TERM); everything is joined before the value is formed, i.e. inside the stage.
    const bool two = L.side != nullptr;  // (TDMPC2_ONE_STREAM at create: no side stream on the handle)
    hipStream_t sd = two ? L.side : st;
    const LayBufs b2 = lay_bufs(h, two ? 1 : 0);
    for (int t = 0; t < H; ++t) {
        const int total = (int)rows * A;
        if (h->split)
            hipLaunchKernelGGL(l_set_action_s, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, st, L.X, L.Kin, c.latent_dim, A, NF, H, t,
                               (int)rows, actions, N, n_off);
        else
            hipLaunchKernelGGL(l_set_action, dim3((total + 255) / 256), dim3(256), 0, st, L.X, L.Kin, c.latent_dim, A, NF, H, t,
                               (int)rows, actions, N, n_off);
        LAUNCH_CHECK();
        if (two) {  // fork: the side stream sees [z_t | a_t]
            HIP_TRY(hipEventRecord(L.ev_fork, st));
            HIP_TRY(hipStreamWaitEvent(sd, L.ev_fork, 0));
        }
        // t = 0: the z columns' products come from lay_cvec; contract the action columns only
        GemmRange r_rew{c.latent_dim / 16, (L.Kin - c.latent_dim) / 16, L.cvec, (long)L.Mp, c.latent_dim};
        GemmRange r_dyn = r_rew;
        r_dyn.bias_env = L.cvec + (size_t)L.cvec_rows * L.Mp;
        const bool shortk = t == 0 && L.cvec_ready;
        // reward(z, a_t) -> two_hot_inv -> G += disc * (1 - term) * r
        if ((rc = lay_hidden(h, sd, h->rew, BE_REW, rows, rows_p, N, nullptr, false, &b2, two ? L.ev_xread : nullptr,
                             shortk ? &r_rew : nullptr))) return rc;
        if ((rc = lay_head_twohot(h, sd, b2.HB, rows, rows_p, N, h->rew.l[2], 0, 0, nullptr, b2.LG, 0, t, disc_pow, value, trace))) return rc;
        if (two) HIP_TRY(hipEventRecord(L.ev_side, sd));
        // z = next(z, a_t)
        if ((rc = lay_dynamics(h, st, rows, rows_p, N, two ? L.ev_xread : nullptr, shortk ? &r_dyn : nullptr))) return rc;
        if (c.episodic) {  // termination head on the new latent (tdmpc2.py:133-134)
            if ((rc = lay_hidden(h, st, h->term, -1, rows, rows_p, N, nullptr, false))) return rc;
            if ((rc = lay_gemm(h, st, L.HB, L.Mp, rows_p, N, h->term.l[2], 0, 0, -1, nullptr, L.LG, L.ldl))) return rc;
            if (two) HIP_TRY(hipStreamWaitEvent(st, L.ev_side, 0));  // the reward's two-hot has read TERM
            hipLaunchKernelGGL(l_term, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, L.LG, L.ldl, (int)rows, L.TERM);
            LAUNCH_CHECK();
        }
    }
    // a_H = pi(z_H); value = G + disc^H (1 - term) avg of the two selected Q heads
    if ((rc = lay_policy(h, st, rows, rows_p, N, N, act_mask, pi_eps, pi_eps_estride, seed, call, SITE_PI, iter, nullptr, 0,
                         trace, n_off))) return rc;
    if (two) {
        HIP_TRY(hipEventRecord(L.ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(sd, L.ev_fork, 0));
    }
    // head 0 on the main stream, head 1 beside it
    if ((rc = lay_hidden(h, st, h->q[0], BE_Q0, rows, rows_p, N, qidx, true))) return rc;
    if ((rc = lay_head_twohot(h, st, L.HB, rows, rows_p, N, h->q[0].l[2], q_wstride(h, 2), q_bstride(h, 2), qidx, L.LG, 1, 0, disc_pow, value,
                              trace, ranged ? NF : 0, n_off))) return rc;
    if ((rc = lay_hidden(h, sd, h->q[0], BE_Q0, rows, rows_p, N, qidx + 1, true, &b2))) return rc;
    if ((rc = lay_gemm(h, sd, b2.HB, L.Mp, rows_p, N, h->q[0].l[2], q_wstride(h, 2), q_bstride(h, 2), -1, qidx + 1, b2.LG,
                       L.ldl))) return rc;
    if (two) {  // join: G (all reward two-hots) and the second head's logits are complete
        HIP_TRY(hipEventRecord(L.ev_side, sd));
        HIP_TRY(hipStreamWaitEvent(st, L.ev_side, 0));
    }
    return lay_twohot(h, st, rows, N, 2, 0, disc_pow, value, trace, ranged ? NF : 0, n_off, b2.LG);
}

// The P policy-prior trajectories on the few-row path: H x pi (3 layers) and (H - 1) x dynamics, one problem per launch.
int lay_pitraj_m(tdmpc2_plan *h, hipStream_t st, int E, const float *z0, const float *act_mask, const float *tape_eps,
                 unsigned long long seed, unsigned call) {
    const tdmpc2_plan_cfg &c = h->cfg;
    const Layered &L = h->lay;
    const int P = c.num_pi_trajs, H = c.horizon, A = c.action_dim, rpe = L.Ppad;
    const size_t rows = (size_t)E * rpe, rows_p = round_up(rows, GBM);
    int rc;
    hipLaunchKernelGGL(l_init_x_s, init_x_grid(rows, L.Kin), dim3(256), 0, st, L.X, L.Kin, c.latent_dim, rpe, z0, (float *)nullptr,
                       (float *)nullptr, (int)rows);
    LAUNCH_CHECK();
    for (int t = 0; t < H; ++t) {
        MidOp o0 = mid_hidden(L, L.X, L.Kin, h->pi, 0, BE_PI, nullptr, false, L.HA, c.mlp_dim);
        if ((rc = mid_stage(h, st, rows, rows_p, rpe, &o0, 1))) return rc;
        MidOp o1 = mid_hidden(L, L.HA, L.Mp, h->pi, 1, -1, nullptr, false, L.HB, c.mlp_dim);
        if ((rc = mid_stage(h, st, rows, rows_p, rpe, &o1, 1))) return rc;
        MidOp o2 = mid_head(L, L.HB, h->pi, nullptr, false, MR_PI);
        PiHeadParams &p = o2.pi;
        p.rows = (int)rows; p.rows_per_env = rpe; p.nvalid = P; p.A = A; p.L = c.latent_dim; p.ldx = L.Kin;
        p.lsmin = c.log_std_min; p.lsdif = c.log_std_dif; p.mask = act_mask;
        p.eps = tape_eps ? tape_eps + (size_t)t * P * A : nullptr; p.eps_estride = (long)H * P * A;  // tape layout [E, H, P, A]
        p.seed = seed; p.call = call; p.site = SITE_PITRAJ; p.iter = t; p.X = L.X; p.actions = h->actions; p.t = t; p.H = H;
        p.N = c.num_samples; p.trace = nullptr; p.row_env = nullptr; p.n_off = 0;
        if ((rc = mid_stage(h, st, rows, rows_p, rpe, &o2, 1))) return rc;
        if (t == H - 1) break;
        MidOp d0 = mid_hidden(L, L.X, L.Kin, h->dyn, 0, BE_DYN, nullptr, false, L.HA, c.mlp_dim);
        if ((rc = mid_stage(h, st, rows, rows_p, rpe, &d0, 1))) return rc;
        MidOp d1 = mid_hidden(L, L.HA, L.Mp, h->dyn, 1, -1, nullptr, false, L.HB, c.mlp_dim);
        if ((rc = mid_stage(h, st, rows, rows_p, rpe, &d1, 1))) return rc;
        MidOp d2 = mid_head(L, L.HB, h->dyn, nullptr, false, MR_LN_SIMNORM);
        d2.out = L.X; d2.ldo = L.Kin; d2.width = c.latent_dim;
        if ((rc = mid_stage(h, st, rows, rows_p, rpe, &d2, 1))) return rc;
    }
    return 0;
}

// The P policy-prior trajectories (tdmpc2/tdmpc2.py:154-160): rows n < P of actions[E, H, N, A].
int lay_pitraj(tdmpc2_plan *h, hipStream_t st, int E, const float *z0, const float *act_mask, const float *tape_eps,
               unsigned long long seed, unsigned call) {
    const tdmpc2_plan_cfg &c = h->cfg;
    const Layered &L = h->lay;
    const int P = c.num_pi_trajs, H = c.horizon, A = c.action_dim, rpe = L.Ppad;
    const size_t rows = (size_t)E * rpe, rows_p = round_up(rows, GBM);
    int rc;
    if (mid_ok(h, rows_p)) return lay_pitraj_m(h, st, E, z0, act_mask, tape_eps, seed, call);
    if ((rc = lay_arrive_reset(h, st))) return rc;
    if (h->split)
        hipLaunchKernelGGL(l_init_x_s, init_x_grid(rows, L.Kin), dim3(256), 0, st, L.X, L.Kin, c.latent_dim, rpe, z0, (float *)nullptr,
                           (float *)nullptr, (int)rows);
    else
        hipLaunchKernelGGL(l_init_x, dim3((unsigned)rows), dim3(256), 0, st, L.X, L.Kin, c.latent_dim, rpe, z0, (float *)nullptr,
                           (float *)nullptr);
    LAUNCH_CHECK();
    for (int t = 0; t < H; ++t) {
        // tape layout [E, H, P, A]: env stride H*P*A, this step's slice at t*P*A
        const float *eps = tape_eps ? tape_eps + (size_t)t * P * A : nullptr;
        if ((rc = lay_policy(h, st, rows, rows_p, rpe, P, act_mask, eps, (long)H * P * A, seed, call, SITE_PITRAJ, t,
                             h->actions, t))) return rc;
        if (t == H - 1) break;
        if ((rc = lay_dynamics(h, st, rows, rows_p, rpe))) return rc;
    }
    return 0;
}

// One CEM iteration's sampled actions (tdmpc2.py:176-181: rows n >= P of h->actions, every step) and its two Q heads per plan.
int lay_sample_iteration(tdmpc2_plan *h, hipStream_t st, int E, int iter, const float *act_mask, const tdmpc2_noise *tape,
                         uint64_t seed, unsigned call, int *qbuf) {
    const tdmpc2_plan_cfg &c = h->cfg;
    const int H = c.horizon, N = c.num_samples, A = c.action_dim, P = c.num_pi_trajs, I = c.iterations;
    SampleParams sp{};
    sp.E = E; sp.H = H; sp.N = N; sp.A = A; sp.P = P; sp.iter = iter; sp.mean = h->mean; sp.std = h->std; sp.mask = act_mask;
    sp.eps = tape ? tape->sample_eps + (size_t)iter * H * (N - P) * A : nullptr;
    sp.eps_estride = (long)I * H * (N - P) * A;
    sp.seed = seed; sp.call = call; sp.actions = h->actions;
    // in-kernel Philox and at least two heads: the iteration's Q heads are drawn by l_sample's first workgroup (one launch less per iteration)
    const bool fold_q = !tape && c.num_q >= 2;
    sp.qidx = fold_q ? qbuf : nullptr; sp.nq = c.num_q;
    {   // ... and the arrival counters of the stage that follows are zeroed by the same launch (stream order: the previous stage is done)
        Layered &L = h->lay;
        if (L.arrive) {
            if (L.arrive_off > L.arrive_high) L.arrive_high = L.arrive_off;
            sp.arrive = L.arrive; sp.arrive_n = (int)L.arrive_high;
            L.arrive_off = 0;
            L.arrive_clean = true;
        }
    }
    hipLaunchKernelGGL(l_sample, dim3(1024), dim3(256), 0, st, sp);
    LAUNCH_CHECK();
    if (fold_q) return 0;
    return lay_set_qidx(h, st, E, tape ? tape->qidx + (size_t)iter * 2 : nullptr, (long)I * 2, c.num_q, iter, seed, call, qbuf);
}

int lay_set_qidx(tdmpc2_plan *, hipStream_t st, int E, const int *qidx, long stride, int nq, int iter, uint64_t seed, unsigned call,
                 int *dst) {
    if (qidx) hipLaunchKernelGGL(l_copy_qidx, dim3((E + 255) / 256), dim3(256), 0, st, E, qidx, stride, dst);
    else hipLaunchKernelGGL(l_qidx, dim3((E + 255) / 256), dim3(256), 0, st, E, nq, iter, (unsigned long long)seed, call, dst);
    LAUNCH_CHECK();
    return 0;
}

// Everything of TDMPC2._plan after encode() (tdmpc2/tdmpc2.py:154-206) on the layered path.
int lay_run(tdmpc2_plan *h, hipStream_t st, int E, const float *z0, const float *task_emb, const float *act_mask,
            const float *disc_pow, float *prev_mean, const uint8_t *t0, int eval_mode, const tdmpc2_noise *tape,
            uint64_t seed, float *action, const tdmpc2_debug *dbg) {
    const tdmpc2_plan_cfg &c = h->cfg;
    const int H = c.horizon, N = c.num_samples, A = c.action_dim, K = c.num_elites, P = c.num_pi_trajs, I = c.iterations;
    const unsigned call = h->call++;
    int rc;
    if ((rc = lay_setup(h, st, E, task_emb, prev_mean, t0, true))) return rc;
    if ((rc = lay_cvec(h, st, E, z0))) return rc;
    // one plan on the few-row path: the policy-prior trajectories ride along in iteration 0's stage (lay_estimate_value_m) -- their own
    // pass would repeat the dynamics of rows the stage rolls out anyway (12 of its 27 layer launches on the 317M model)
    const bool pifold = P > 0 && E == 1 && h->lay.knob[LK_MID_PIFOLD] && mid_ok(h, round_up((size_t)E * N, GBM));
    if (P > 0 && !pifold && (rc = lay_pitraj(h, st, E, z0, act_mask, tape ? tape->pi_traj_eps : nullptr, seed, call))) return rc;
    int refit_stage = 0;
    const size_t refit_lds = refit_lds_bytes(N, K, H, A, &refit_stage);
    for (int it = 0; it < I; ++it) {
        if ((rc = lay_sample_iteration(h, st, E, it, act_mask, tape, seed, call, h->lay.qidx))) return rc;
        if (h->profiling && h->ev_used + 2 <= (int)h->ev.size()) HIP_TRY(hipEventRecord(h->ev[h->ev_used], st));
        h->lay.pifold = pifold && it == 0;
        h->lay.pifold_eps = tape ? tape->pi_traj_eps : nullptr;
        rc = lay_estimate_value(h, st, E, z0, act_mask, disc_pow, h->actions, tape ? tape->pi_eps + (size_t)it * N * A : nullptr,
                                (long)I * N * A, h->lay.qidx, seed, call, it, h->value, nullptr);
        h->lay.pifold = false;
        if (rc) return rc;
        if (h->profiling && h->ev_used + 2 <= (int)h->ev.size()) {
            HIP_TRY(hipEventRecord(h->ev[h->ev_used + 1], st));
            h->ev_used += 2;
        }
        if (dbg && dbg->actions)
            HIP_TRY(hipMemcpy2DAsync(dbg->actions + (size_t)it * H * N * A, (size_t)I * H * N * A * 4, h->actions,
                                     (size_t)H * N * A * 4, (size_t)H * N * A * 4, E, hipMemcpyDeviceToDevice, st));
        RefitParams fp{};
        fp.Nvalid = c.num_valid_samples; fp.E = E; fp.N = N; fp.H = H; fp.A = A; fp.K = K; fp.iter = it; fp.last = (it == I - 1); fp.eval_mode = eval_mode; fp.stage = refit_stage;
        fp.temperature = c.temperature; fp.min_std = c.min_std; fp.max_std = c.max_std;
        fp.value = h->value; fp.actions = h->actions; fp.act_mask = act_mask; fp.mean = h->mean; fp.std = h->std;
        fp.gumbel_exp = tape ? tape->gumbel_exp : nullptr; fp.final_eps = tape ? tape->final_eps : nullptr;
        fp.seed = seed; fp.call = call; fp.prev_mean = prev_mean; fp.action = action;
        fp.err = h->lay.fuse_ln ? h->cl_err_dev : nullptr;  // a fused-epilogue wait that gave up: NaN action, prev_mean kept
        if (dbg) {
            if (dbg->value) { fp.dbg_value = dbg->value + (size_t)it * N; fp.dbg_value_es = (long)I * N; }
            if (dbg->elite_idx) { fp.dbg_idx = dbg->elite_idx + (size_t)it * K; fp.dbg_idx_es = (long)I * K; }
            if (dbg->score) { fp.dbg_score = dbg->score + (size_t)it * K; fp.dbg_score_es = (long)I * K; }
            if (dbg->mean) { fp.dbg_mean = dbg->mean + (size_t)it * H * A; fp.dbg_mean_es = (long)I * H * A; }
            if (dbg->std) { fp.dbg_std = dbg->std + (size_t)it * H * A; fp.dbg_std_es = (long)I * H * A; }
        }
        if ((rc = launch_refit(fp, E, N, refit_lds, st))) return rc;
    }
    return TDMPC2_OK;
}

// pi + two Q heads on `rows` latent rows (TDMPC2._td_target, tdmpc2/tdmpc2.py:239-254, and the forward half of update_pi,
// tdmpc2.py:208-225) on the layered family: the same GEMM / row kernels as a planning step, the row -> task map (multitask)
// selecting the first-layer bias, action mask and discount of each row.
int lay_value(tdmpc2_plan *h, hipStream_t st, int rows, const float *z, bool target, bool reduce_min, const float *pi_eps,
              const int *qidx_dev /* [2] */, unsigned long long seed, unsigned call, const float *reward, const float *terminated,
              float discount, const int *row_task /* padded [rows_p] or null */, float *action, float *out) {
    const tdmpc2_plan_cfg &c = h->cfg;
    Layered &L = h->lay;
    const size_t rows_p = round_up((size_t)rows, GBM);
    const int rpe = (int)rows_p;  // one "plan" spanning the call: sel index 0, noise / action index = row
    {
        int rc0 = lay_arrive_reset(h, st);
        if (rc0) return rc0;
    }
    if (h->split) hipLaunchKernelGGL(l_init_rows_s, dim3((unsigned)(rows_p / 32)), dim3(256), 0, st, L.X, L.Kin, c.latent_dim, z, rows);
    else hipLaunchKernelGGL(l_init_rows, dim3((unsigned)rows_p), dim3(256), 0, st, L.X, L.Kin, c.latent_dim, z, rows);
    LAUNCH_CHECK();
    struct Restore {  // the helpers read these from the handle
        Layered &L; const HostNet *q; const float *bt; const int *re;
        ~Restore() { L.qarr = q; L.bias_tab = bt; L.row_env = re; }
    } restore{L, L.qarr, L.bias_tab, L.row_env};
    L.qarr = target ? h->tq : h->q;
    if (c.multitask) { L.bias_tab = h->beff_tab; L.row_env = row_task; }
    int rc;
    // a = pi(z): into the action columns of X, and into `action` [rows, A] when asked for
    if ((rc = lay_policy(h, st, (size_t)rows, rows_p, rpe, rows, c.multitask ? h->mask_tab : nullptr, pi_eps, 0, seed, call, SITE_PI, 0,
                         action, 0))) return rc;
    for (int j = 0; j < 2; ++j) {
        if ((rc = lay_hidden(h, st, L.qarr[0], BE_Q0, (size_t)rows, rows_p, rpe, qidx_dev + j, true))) return rc;
        if ((rc = lay_gemm(h, st, L.HB, L.Mp, rows_p, rpe, L.qarr[0].l[2], q_wstride(h, 2), q_bstride(h, 2), -1, qidx_dev + j, L.LG,
                           L.ldl))) return rc;
        ValueHeadParams p{};
        p.lg = L.LG; p.ld = L.ldl; p.rows = rows; p.num_bins = c.num_bins; p.mode = j; p.reduce_min = reduce_min ? 1 : 0;
        p.bins = h->bins; p.qtmp = L.QT; p.out = out; p.reward = reward; p.terminated = terminated; p.discount = discount;
        p.disc_tab = (c.multitask && reward) ? h->disc_tab : nullptr; p.row_env = row_task;
        p.err = (h->split && L.fuse_ln) ? h->cl_err_dev : nullptr; p.action = action; p.A = c.action_dim;
        hipLaunchKernelGGL(l_value_head, dim3((unsigned)((rows + RW_THREADS / 64 - 1) / (RW_THREADS / 64))), dim3(RW_THREADS), 0, st, p);
        LAUNCH_CHECK();
    }
    return 0;
}
