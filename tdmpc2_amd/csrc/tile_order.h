// Tile order of the fused-epilogue GEMM launches (g_gemm_s<.., EPI != 0>): which output tile a workgroup computes, and --
// because block b runs on XCD b % 8 and every XCD dispatches its share of the grid in order -- which XCD's L2 sees which
// weights and activation rows, and where the column blocks of a row block (which wait for each other) sit in
// the dispatch order.  Plain C++ (no HIP types): the kernel and the host include it, and tests/test_tile_order.py compiles
// it with g++ and checks the invariants the waits rely on.
#pragma once

#ifndef __HIPCC__
#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
#endif

struct GemmSOrder {
    int xcd_rows;   // 1: XCD-local row blocks; g = 2 / 4: a row block's column blocks on g neighbouring XCDs (ncolblk / g each)
    int ncol_grid;  // row-major: blocks per row of the grid (>= ncolblk, a multiple of 8 when padded); 0 with xcd_rows
    int nblk;       // workgroups to launch
};

// The rule (measured: profiles/README.md r3v, r3u, r3s).  `force_xcd_rows` / `col_pad`: -1 = automatic, 0 / 1 = never / always.
__host__ __device__ inline GemmSOrder gemm_s_order(int nrowblk, int ncolblk, int force_xcd_rows = -1, int col_pad = 1) {
    GemmSOrder o;
    // Row-major puts block rb * ncolblk + cb on XCD (rb * ncolblk + cb) % 8.  With ncolblk a multiple of 8 (317M model: 16) an
    // XCD only ever sees column blocks x and x + 8 -- 1/8 of the weights, every A row.  Otherwise (48M: 7; SimNorm layers: 3, 6)
    // it sees every column block and nearly every row block; XCD-local row blocks then read each A row through ONE L2.
    const bool xr = force_xcd_rows >= 0 ? force_xcd_rows != 0 : (ncolblk % 8 != 0 && nrowblk >= 64);
    // force_xcd_rows = 2 / 4: XCD rectangles -- XCD x computes column part x % g of the row blocks of class x / g.  Per XCD and
    // launch that is A/(8/g) + W/g instead of A/8 + W (g = 1) or A + W/8 (row-major, 8 | ncolblk): least for g ~ sqrt(8 W / A)
    const int g = (force_xcd_rows == 2 || force_xcd_rows == 4) && ncolblk % force_xcd_rows == 0 ? force_xcd_rows : 1;
    o.xcd_rows = xr ? g : 0;
    o.ncol_grid = 0;
    if (xr) {
        // g = 1: XCD x: row blocks x, x + 8, ...; a row block past the end leaves
        o.nblk = 8 * ((nrowblk + 8 / g - 1) / (8 / g)) * (ncolblk / g);
    } else {
        // few row blocks, more than 8 column blocks, not a multiple of 8 (single plans of the 48M model: 14): pad the row of
        // blocks to a multiple of 8 -- the XCD becomes a function of the column block alone
        o.ncol_grid = (col_pad != 0 && ncolblk % 8 != 0 && ncolblk > 8) ? (ncolblk + 7) / 8 * 8 : ncolblk;
        o.nblk = nrowblk * o.ncol_grid;
    }
    return o;
}

// Block b -> (row block, column block); false: the block has no tile (padding) and leaves at once.
__host__ __device__ inline bool gemm_s_tile(int b, int nrowblk, int ncolblk, int xcd_rows, int ncol_grid, int &rb, int &cb) {
    if (xcd_rows) {  // XCD x's t-th workgroup: column block t % ncolblk of row block (t / ncolblk) * 8 + x  (g = 1)
        const int x = b & 7, t = b >> 3, g = xcd_rows;
        const int cpb = ncolblk / g, nrc = 8 / g;  // column blocks per part, row classes
        const int rbl = t / cpb;
        cb = (x % g) * cpb + (t - rbl * cpb);
        rb = rbl * nrc + x / g;
        return rb < nrowblk;  // (the whole row block: every one of its workgroups takes the same exit)
    }
    const int ncg = ncol_grid ? ncol_grid : ncolblk;  // column blocks of a row block on consecutive block ids
    rb = b / ncg;
    cb = b - rb * ncg;
    return cb < ncolblk;
}

// ---------------------------------------------------------------- g_gemm_w with a K-split tail (round 5)
// 256 x 256 tiles on one workgroup per CU quantise badly: the 48M model's hidden layers are 60 x 7 = 420 tiles on 256 CUs,
// 1.64 rounds paid as 2 (as 4 for the two chains of a stage).  The cure is a finer grain for the LAST, partly filled round only:
// per XCD the first `full` tiles of its list (whole rounds of its CUs) are computed as before, every remaining tile is split
// along K into `parts` workgroups whose partial accumulators meet in an L2 / Infinity-Cache workspace; the part that arrives
// last adds them IN PART ORDER and runs the tile's NormedLinear epilogue (layered_wide.cuh).  Lists: XCD x owns the row blocks
// x, x + 8, ... below 8 q (q = nrowblk / 8) -- the XCD-local order of gemm_s_tile: the peers of a row block consecutive on one
// XCD --; the tiles of the nrowblk % 8 remaining row blocks are dealt out in contiguous runs, so that the lists differ by at
// most one tile (the plain XCD-local order gives 4 XCDs a whole row block more: 56 against 49 tiles for the 48M model).
// Order of the tail inside an XCD: groups of GW_SE tiles, PART-MAJOR inside a group (t0.p0 t1.p0 t2.p0 t3.p0 t0.p1 ...).  An
// XCD does not hand its share of a grid to "whichever CU is free": consecutive workgroups go round-robin to its 4 shader engines
// (8 CUs each), and a workgroup only ever runs on its engine.  With the parts of a tile on consecutive slots (tile-major), the
// parts p of ALL tiles land on engine p; the last arrivers -- which wait for their row block's peers -- pile up on one engine, fill
// its 8 CUs, and the parts still queued for that engine never start (317M model, 16 peers per row block: 8 or 15 of 16 arrivals,
// then the bounded wait gave up; fault codes in profiles/README.md r5c).  Part-major in groups of 4 keeps every part of a tile on
// ONE engine and spreads the tiles -- and so the waiters -- over all four, as the whole tiles are.
#ifndef GW_SE
#define GW_SE 4
#endif
// ---------------------------------------------------------------------------------------------------------------------------
// g_gemm_w's DMA ring (layered_wide.cuh): the schedule arithmetic -- which slabs a phase requests, reads, and how many DMA
// requests it may leave in flight -- in one place, shared by the kernel and by tests/test_ring_schedule.py, which replays the
// ring for every contraction length and checks that no slab is read before it has landed or overwritten before it was read.
// A slab = one k16-step of the tile = 4 DMA requests per wave; `ns` = ring slots (GW_NSLOT).
struct GwTailStep {
    bool issue;  // slab ss + ns exists: request it into the slot slab ss has just left
    bool next;   // slab ss + 1 exists: read it from LDS during this phase's MFMAs
    int vmc;     // requests that may stay in flight at the top of the phase (slab ss + 1 must have landed)
};
__host__ __device__ inline int gw_prologue_slabs(int nk, int ns) { return nk < ns ? nk : ns; }
// after the prologue slab 0 must have landed: every later slab's requests may stay in flight
__host__ __device__ inline int gw_prologue_vmcnt(int npro) { return npro > 1 ? 4 * (npro - 1) : 0; }
// a whole trip of `u` steady phases from slab s on: each of them has a slab s' + ns <= nk - 1 to request
__host__ __device__ inline bool gw_steady_trip(int s, int nk, int ns, int u) { return s + u - 1 + ns < nk; }
__host__ __device__ inline int gw_steady_vmcnt(int ns) { return 4 * (ns - 2); }
__host__ __device__ inline GwTailStep gw_tail_step(int ss, int nk, int ns) {
    const int last_req = ss + ns - 1 < nk - 1 ? ss + ns - 1 : nk - 1;  // the newest slab requested so far
    const int inflight = last_req - (ss + 1);                         // slabs behind ss + 1 that may still be on their way
    return GwTailStep{ss + ns < nk, ss + 1 < nk, inflight > 0 ? 4 * inflight : 0};
}
// the immediate gw_phase waits with (s_waitcnt takes constants: a ladder over the values that occur)
__host__ __device__ inline int gw_phase_vmcnt(bool steady, int vmc, int ns) {
    if (steady || vmc >= gw_steady_vmcnt(ns)) return gw_steady_vmcnt(ns);
    return vmc == 8 ? 8 : vmc == 4 ? 4 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// g_gemm_m's DMA ring (layered_mid.cuh): the same schedule with GM_REQ = 6 requests per wave and k32-slab and a ring of 3.
constexpr int GM_REQ = 6;
struct GmTailStep {
    bool issue;  // slab ss + ns exists: request it into the slot slab ss has just left
    bool next;   // slab ss + 1 exists: read it from LDS during this phase's MFMAs
    int vmc;     // requests that may stay in flight at the top of the phase (slab ss + 1 must have landed)
};
__host__ __device__ inline int gm_prologue_slabs(int nk, int ns) { return nk < ns ? nk : ns; }
__host__ __device__ inline int gm_prologue_vmcnt(int npro) { return npro > 1 ? GM_REQ * (npro - 1) : 0; }
__host__ __device__ inline bool gm_steady_trip(int s, int nk, int ns, int u) { return s + u - 1 + ns < nk; }
__host__ __device__ inline int gm_steady_vmcnt(int ns) { return GM_REQ * (ns - 2); }
__host__ __device__ inline GmTailStep gm_tail_step(int ss, int nk, int ns) {
    const int last_req = ss + ns - 1 < nk - 1 ? ss + ns - 1 : nk - 1;
    const int inflight = last_req - (ss + 1);
    return GmTailStep{ss + ns < nk, ss + 1 < nk, inflight > 0 ? GM_REQ * inflight : 0};
}

struct GemmWOrder {
    int parts;     // K-parts of a tail tile (1: nothing is split -- the caller then keeps the order of gemm_s_order)
    int full;      // per XCD: tiles [0, full) of its list are whole
    int max_tail;  // most tail tiles on any XCD (slots of the partial-sum workspace per XCD)
    int per_xcd;   // workgroup slots per XCD = full + max_tail * parts
    int nblk;      // 8 * per_xcd
    int rounds1k;  // the rule's estimate of the launch's duration in 1/1000 tile times (tests / logs)
};

__host__ __device__ inline int gemm_w_list_len(int nrowblk, int ncolblk, int x) {
    const int q = nrowblk / 8, S = (nrowblk - 8 * q) * ncolblk;
    return q * ncolblk + (S * (x + 1) / 8 - S * x / 8);
}

// cus_per_xcd: resident workgroups per XCD (one per CU: 32); nk: k16-slabs of the contraction; max_parts <= 4;
// ovh1k: what a split tile costs beyond its share of the slabs -- partial store, the last arriver's reads -- in 1/1000 of a
// whole tile's time (the caller derives it from nk)
__host__ __device__ inline GemmWOrder gemm_w_order(int nrowblk, int ncolblk, int cus_per_xcd, int nk, int max_parts, int ovh1k) {
    GemmWOrder o;
    const int q = nrowblk / 8, S = (nrowblk - 8 * q) * ncolblk;
    const int n_min = q * ncolblk + S / 8, n_max = q * ncolblk + (S + 7) / 8;
    o.full = n_min / cus_per_xcd * cus_per_xcd;
    o.max_tail = n_max - o.full;
    o.parts = 1;
    int best = o.max_tail > 0 ? 1000 * ((o.max_tail + cus_per_xcd - 1) / cus_per_xcd) : 0;
    for (int P = 2; P <= max_parts && P <= 4; ++P) {
        if (nk / P < 8) break;  // too few slabs per part to amortise the ring's fill
        const int c = 1000 * ((o.max_tail * P + cus_per_xcd - 1) / cus_per_xcd) / P + ovh1k;
        if (c < best - 50) {  // at least a twentieth of a tile time better
            best = c;
            o.parts = P;
        }
    }
    o.rounds1k = 1000 * (o.full / cus_per_xcd) + best;
    o.per_xcd = o.full + (o.max_tail + GW_SE - 1) / GW_SE * GW_SE * o.parts;  // whole groups of GW_SE tiles
    o.nblk = 8 * o.per_xcd;
    return o;
}

// Block b -> (row block, column block, K-part, workspace slot of a split tile or -1); false: the block has no tile.
__host__ __device__ inline bool gemm_w_tile(int b, int nrowblk, int ncolblk, int full, int parts, int max_tail, int &rb, int &cb, int &part,
                                            int &slot) {
    const int x = b & 7, t = b >> 3;
    const int q = nrowblk / 8, S = (nrowblk - 8 * q) * ncolblk, s0 = S * x / 8;
    const int n = q * ncolblk + (S * (x + 1) / 8 - s0);
    int i;
    if (t < full) {
        i = t; part = 0; slot = -1;
    } else {  // groups of GW_SE tiles, part-major inside a group (see above)
        const int u = t - full, gsz = GW_SE * parts, g = u / gsz, r = u - g * gsz;
        part = r / GW_SE;
        i = full + g * GW_SE + (r - part * GW_SE);
        slot = x * max_tail + (i - full);
    }
    if (i >= n) return false;
    if (i < q * ncolblk) {
        const int rbl = i / ncolblk;
        rb = rbl * 8 + x; cb = i - rbl * ncolblk;
    } else {
        const int s = s0 + i - q * ncolblk;
        rb = 8 * q + s / ncolblk; cb = s - (s / ncolblk) * ncolblk;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Issue placement inside a phase of the two LDS-DMA ring GEMMs (24 MFMAs per wave and phase in both): which of the wave's requests goes
// out behind MFMA k.  tests/test_ring_schedule.py checks that every request of a phase goes out exactly once and none behind the last MFMA
// (the counted waits of the schedules above assume GW_REQ / GM_REQ requests per phase, whatever their position in it).
constexpr int RING_MFMAS_PER_PHASE = 24;
constexpr int GW_REQ = 4;
// Where a phase's four DMA requests are issued: behind MFMAs 12, 15, 18, 21 (the phase's second half), not all four in front of the
// first MFMA -- 8 waves x 4 requests right behind the barrier held the MFMAs behind them back.  Same bits; 48M model at 30 plans + 1.7 %,
// 317M at 8 plans + 2.2 % (in front of the first four MFMAs, beside the LDS reads: + 1.2 / + 1.4 %; every sixth MFMA of the whole phase:
// 0; 13, 16, 19, 22: the same as this; the last four: + 1.8 / + 1.8 %: profiles/r6zu_/r6zv_issue_placement_ab.txt).
// -DGW_SPREAD_ISSUE=0: the old placement (A/B).
#ifndef GW_SPREAD_ISSUE
#define GW_SPREAD_ISSUE 1
#endif
__host__ __device__ constexpr int gw_req_at(int k) {  // the request that goes out behind MFMA k of a phase (-1: none)
    return GW_SPREAD_ISSUE == 1 ? ((k >= 12 && k < RING_MFMAS_PER_PHASE - 1 && k % 3 == 0) ? (k - 12) / 3 : -1) : -1;
}
// Where a phase's six DMA requests are issued: behind MFMAs 12, 14, .. 22 -- one request per two MFMAs of the phase's second half -- and
// not all six in front of the first MFMA (the 8 waves' 48 requests right behind the barrier held the MFMAs behind them back: the 317M
// plan + 2.3 %, four 48M plans + 1.8 %, same bits; in front of the first six MFMAs, beside the LDS reads: - 2.5 %; every fourth MFMA of the
// whole phase + 1.3 %; the last six + 1.4 %: profiles/r6zu_/r6zv_issue_placement_ab.txt).  -DGM_SPREAD_ISSUE=0: the old placement (A/B).
#ifndef GM_SPREAD_ISSUE
#define GM_SPREAD_ISSUE 1
#endif
__host__ __device__ constexpr int gm_req_at(int k) {  // the request that goes out behind MFMA k of a phase (-1: none)
    return GM_SPREAD_ISSUE == 1 ? ((k >= 12 && k < RING_MFMAS_PER_PHASE - 1 && k % 2 == 0) ? (k - 12) / 2 : -1) : -1;
}
