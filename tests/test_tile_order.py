"""CPU test of the tile order of the fused-epilogue GEMM launches (tdmpc2_amd/csrc/tile_order.h -- the header the kernel and
the host include, compiled here with g++ behind a two-function C shim).

The column blocks of a row block wait for each other inside the launch.  What keeps that wait short -- and
what the measured locality gains rest on -- are properties of the block -> tile map under the hardware's placement rule
(block b runs on XCD b % 8, every XCD dispatches its share in order):

* every tile is computed exactly once, padding blocks have no tile;
* XCD-local order: all column blocks of a row block run on ONE XCD and are consecutive in that XCD's dispatch order (at most
  one partly dispatched row block per XCD and launch);
* row-major order: the column blocks of a row block are consecutive block ids; when the row of blocks is padded to a multiple
  of 8 the XCD is a function of the column block alone (an XCD streams its own column blocks' weights only);
* the rule picks the order the measurements chose (profiles/README.md r3v, r3u, r3s).
"""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = r"""
#include "tile_order.h"
extern "C" void order(int nrowblk, int ncolblk, int force_xcd_rows, int col_pad, int *out) {
    const GemmSOrder o = gemm_s_order(nrowblk, ncolblk, force_xcd_rows, col_pad);
    out[0] = o.xcd_rows; out[1] = o.ncol_grid; out[2] = o.nblk;
}
extern "C" int tile(int b, int nrowblk, int ncolblk, int xcd_rows, int ncol_grid, int *rb, int *cb) {
    return gemm_s_tile(b, nrowblk, ncolblk, xcd_rows, ncol_grid, *rb, *cb) ? 1 : 0;
}
"""


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("tile_order")
    src = d / "shim.cpp"
    src.write_text(SHIM)
    so = d / "libtile_order.so"
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "tdmpc2_amd", "csrc"), str(src), "-o", str(so)],
                   check=True)
    return ctypes.CDLL(str(so))


def order(lib, nrowblk, ncolblk, force=-1, col_pad=1):
    out = (ctypes.c_int * 3)()
    lib.order(nrowblk, ncolblk, force, col_pad, out)
    return {"xcd_rows": out[0], "ncol_grid": out[1], "nblk": out[2]}


def tiles(lib, nrowblk, ncolblk, o):
    """[(block, rb, cb)] of the blocks that have a tile, in block order."""
    rb, cb = ctypes.c_int(), ctypes.c_int()
    res = []
    for b in range(o["nblk"]):
        if lib.tile(b, nrowblk, ncolblk, o["xcd_rows"], o["ncol_grid"], ctypes.byref(rb), ctypes.byref(cb)):
            res.append((b, rb.value, cb.value))
    return res


SHAPES = [(nr, nc) for nr in (1, 5, 8, 23, 24, 64, 120, 130, 512) for nc in (1, 3, 6, 7, 8, 14, 16, 32)]


@pytest.mark.parametrize("force", [-1, 0, 1, 2, 4])
@pytest.mark.parametrize("nrowblk,ncolblk", SHAPES)
def test_every_tile_exactly_once_and_peers_together(lib, nrowblk, ncolblk, force):
    o = order(lib, nrowblk, ncolblk, force)
    t = tiles(lib, nrowblk, ncolblk, o)
    assert sorted((rb, cb) for _, rb, cb in t) == [(r, c) for r in range(nrowblk) for c in range(ncolblk)]
    by_rb = {}
    for b, rb, cb in t:
        by_rb.setdefault(rb, []).append((b, cb))
    if o["xcd_rows"]:
        g = o["xcd_rows"]  # XCDs a row block is spread over (1: XCD-local row blocks; 2 / 4: XCD rectangles)
        assert g in (1, 2, 4) and ncolblk % g == 0 and (g == 1 or force == g)
        assert o["nblk"] == 8 * ((nrowblk + 8 // g - 1) // (8 // g)) * (ncolblk // g)
        for rb, peers in by_rb.items():
            xs = sorted({b % 8 for b, _ in peers})
            assert xs == [g * (rb % (8 // g)) + i for i in range(g)]        # g neighbouring XCDs (g = 1: one)
            for x in xs:
                local = sorted(b // 8 for b, _ in peers if b % 8 == x)      # positions in that XCD's dispatch order
                assert local == list(range(local[0], local[0] + ncolblk // g))  # consecutive there ...
                assert local[0] == (rb // (8 // g)) * (ncolblk // g)        # ... and at the same position on each of the g XCDs
        # in an XCD's order row blocks follow each other whole: at most one of them is partly dispatched at any time
        for x in range(8):
            seq = [rb for b, rb, _ in t if b % 8 == x]
            assert seq == sorted(seq)
    else:
        assert o["ncol_grid"] >= ncolblk and o["nblk"] == nrowblk * o["ncol_grid"]
        for rb, peers in by_rb.items():
            ids = sorted(b for b, _ in peers)
            assert ids == list(range(ids[0], ids[0] + ncolblk))             # consecutive block ids
            assert ids[0] == rb * o["ncol_grid"]
        if o["ncol_grid"] % 8 == 0:
            assert all(b % 8 == cb % 8 for b, _, cb in t)                   # the XCD is a function of the column block


def test_the_rule_is_the_measured_one(lib):
    # 48M model, 30 plans: 120 row blocks x 7 column blocks (128 x 256 tiles) -> XCD-local row blocks
    assert order(lib, 120, 7)["xcd_rows"] == 1
    # its SimNorm output layer (3 column blocks), and the 317M model's (6) at 8 plans
    assert order(lib, 120, 3)["xcd_rows"] == 1 and order(lib, 64, 6)["xcd_rows"] == 1
    # 317M hidden layers: 16 column blocks -- row-major already gives every XCD two column blocks; no padding needed
    o = order(lib, 64, 16)
    assert o == {"xcd_rows": 0, "ncol_grid": 16, "nblk": 1024}
    # single plan of the 48M model: 16 row blocks x 14 column blocks -> row-major, padded to 16
    assert order(lib, 16, 14) == {"xcd_rows": 0, "ncol_grid": 16, "nblk": 256}
    # ... never padded when that would leave an XCD without work (7 -> 8), nor with the switch off
    assert order(lib, 24, 7)["ncol_grid"] == 7 and order(lib, 16, 14, col_pad=0)["ncol_grid"] == 14
    # XCD rectangles are asked for by the caller (TDMPC2_GEMM_W_XCD_ROWS=2; measured, not the default: profiles/README.md r4s, r4t, r4za)
    assert order(lib, 32, 16, force=2) == {"xcd_rows": 2, "ncol_grid": 0, "nblk": 512}
    assert order(lib, 60, 7, force=2)["xcd_rows"] == 1  # 7 column blocks do not split over 2 XCDs
    # the switches
    assert order(lib, 120, 7, force=0)["xcd_rows"] == 0 and order(lib, 8, 16, force=1)["xcd_rows"] == 1
