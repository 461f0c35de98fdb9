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
