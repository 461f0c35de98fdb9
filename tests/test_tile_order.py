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
extern "C" void w_order(int nrowblk, int ncolblk, int cus_per_xcd, int nk, int max_parts, int ovh1k, int *out) {
    const GemmWOrder o = gemm_w_order(nrowblk, ncolblk, cus_per_xcd, nk, max_parts, ovh1k);
    out[0] = o.parts; out[1] = o.full; out[2] = o.max_tail; out[3] = o.per_xcd; out[4] = o.nblk; out[5] = o.rounds1k;
}
extern "C" int w_tile(int b, int nrowblk, int ncolblk, int full, int parts, int max_tail, int *rb, int *cb, int *part, int *slot) {
    return gemm_w_tile(b, nrowblk, ncolblk, full, parts, max_tail, *rb, *cb, *part, *slot) ? 1 : 0;
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
    # XCD rectangles are asked for by the caller (TDMPC2_X_GEMM_W_XCD_ROWS=2; measured, not the default: profiles/README.md r4s, r4t, r4za)
    assert order(lib, 32, 16, force=2) == {"xcd_rows": 2, "ncol_grid": 0, "nblk": 512}
    assert order(lib, 60, 7, force=2)["xcd_rows"] == 1  # 7 column blocks do not split over 2 XCDs
    # the switches
    assert order(lib, 120, 7, force=0)["xcd_rows"] == 0 and order(lib, 8, 16, force=1)["xcd_rows"] == 1


# ---------------------------------------------------------------- g_gemm_w's K-split tail (round 5)
def w_order(lib, nrowblk, ncolblk, nk=112, cus=32, max_parts=4, ovh=88):
    out = (ctypes.c_int * 6)()
    lib.w_order(nrowblk, ncolblk, cus, nk, max_parts, ovh, out)
    return dict(zip(("parts", "full", "max_tail", "per_xcd", "nblk", "rounds1k"), out))


def w_tiles(lib, nrowblk, ncolblk, o):
    rb, cb, part, slot = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    res = []
    for b in range(o["nblk"]):
        if lib.w_tile(b, nrowblk, ncolblk, o["full"], o["parts"], o["max_tail"], ctypes.byref(rb), ctypes.byref(cb), ctypes.byref(part),
                      ctypes.byref(slot)):
            res.append((b, rb.value, cb.value, part.value, slot.value))
    return res


@pytest.mark.parametrize("nrowblk,ncolblk", [(nr, nc) for nr in (1, 2, 4, 7, 8, 12, 23, 46, 60, 64, 130) for nc in (1, 3, 7, 16)])
def test_k_split_order_covers_every_tile_with_all_its_parts(lib, nrowblk, ncolblk):
    """gemm_w_order / gemm_w_tile: every tile is either whole (one workgroup, no workspace slot) or split into exactly `parts`
    workgroups with part indices 0 .. parts - 1 that share ONE workspace slot, sit on ONE XCD and are consecutive in its
    dispatch order's same shader engine (every 4th slot); slots are unique per split tile and below 8 x max_tail; whole tiles come first in every XCD's order; the
    XCDs' lists differ by at most one tile; the peers of a row block below 8 (nrowblk / 8) stay on one XCD, consecutively."""
    for nk in (16, 50, 112, 256):
        o = w_order(lib, nrowblk, ncolblk, nk=nk, ovh=12000 // (nk + 25))
        t = w_tiles(lib, nrowblk, ncolblk, o)
        by_tile = {}
        for b, rb, cb, part, slot in t:
            by_tile.setdefault((rb, cb), []).append((b, part, slot))
        assert sorted(by_tile) == [(r, c) for r in range(nrowblk) for c in range(ncolblk)]
        if o["parts"] == 1:  # nothing split: the launch keeps gemm_s_order's order (the kernel reads this one with parts > 1 only)
            assert nk // 2 < 8 or o["max_tail"] == 0 or o["rounds1k"] % 1000 == 0
            continue
        slots = set()
        per_xcd = [0] * 8
        for (rb, cb), ws in by_tile.items():
            xs = {b % 8 for b, _, _ in ws}
            assert len(xs) == 1
            per_xcd[xs.pop()] += 1
            if len(ws) == 1 and ws[0][2] < 0:
                assert ws[0][1] == 0 and ws[0][0] // 8 < o["full"]
                continue
            assert o["parts"] > 1 and sorted(p for _, p, _ in ws) == list(range(o["parts"]))
            assert len({s for _, _, s in ws}) == 1 and 0 <= ws[0][2] < 8 * o["max_tail"] and ws[0][2] not in slots
            slots.add(ws[0][2])
            # part-major in groups of 4 tiles: the parts of a tile sit 4 slots apart -- on ONE shader engine (slot % 4), in part order
            ts = [b // 8 for b, _, _ in sorted(ws, key=lambda w: w[1])]
            assert ts == list(range(ts[0], ts[0] + 4 * o["parts"], 4)) and ts[0] >= o["full"] and len({t % 4 for t in ts}) == 1
        assert max(per_xcd) - min(per_xcd) <= 1
        q = nrowblk // 8
        for rb in range(8 * q):
            peers = sorted(b for (r, _), ws in by_tile.items() if r == rb for b, _, _ in ws)
            assert {b % 8 for b in peers} == {rb % 8}


def test_k_split_rule_picks_what_the_round_arithmetic_says(lib):
    """The shapes the rule was made for: the 48M model's hidden layers at the benched E = 30 (60 x 7 tiles: per XCD 32 whole +
    20 / 21 tail tiles -> 3 parts: 63 workgroups = two sub-rounds of a third), its SimNorm layer (60 x 3 = 180 tiles, none whole
    -> 4 parts), one 317M plan (4 x 16 = 64 tiles -> 4 parts fill the 256 CUs), and launches that already fill their rounds
    (317M, 8 plans: 512 tiles -> nothing split)."""
    o = w_order(lib, 60, 7, nk=112, ovh=88)
    assert (o["parts"], o["full"], o["max_tail"]) == (3, 32, 21) and o["nblk"] == 8 * (32 + 24 * 3)  # (21 tail tiles: 6 groups of 4)
    assert 1600 < o["rounds1k"] < 1800  # 1 + 2/3 + overhead, against 2 rounds unsplit
    o = w_order(lib, 60, 3, nk=112, ovh=88)
    assert (o["parts"], o["full"]) == (4, 0)
    o = w_order(lib, 4, 16, nk=256, ovh=43)
    assert (o["parts"], o["full"], o["max_tail"], o["nblk"]) == (4, 0, 8, 256)
    o = w_order(lib, 32, 16, nk=256, ovh=43)
    assert o["parts"] == 1
    o = w_order(lib, 60, 7, nk=2, ovh=400)  # the t = 0 first layers contract the action columns only: K = 32
    assert o["parts"] == 1


# ---------------------------------------------------------------- a CPU model of the dispatcher (round 6)
# DESIGN 8 argues deadlock-freedom of the launches whose workgroups wait for each other (the NormedLinear epilogue inside the
# GEMM: the column blocks of a row block; with a K-split tail the last arriver of every tile) from how the hardware was SEEN to
# place workgroups: 8 XCDs, block b on XCD b % 8; an XCD hands the workgroups of its share of a grid, in order, round-robin to
# its 4 shader engines of 8 CUs; a workgroup only ever runs on its engine; a waiting workgroup keeps its CU.  This is that
# model, and the invariant every order the library can pick has to satisfy in it -- so that a new order is judged here, on the
# CPU, and not by thousands of stress stages on a GPU.
#
# Per XCD and launch: the ordered list of its workgroups (slot t -> engine t % 4; a padding block takes a slot and leaves at
# once).  After a prefix of p slots has been dispatched, a workgroup can be HELD -- waiting, possibly for ever -- only if its row
# block is not completely inside the prefix; a tile in K-parts holds at most ONE workgroup (its last arriver; the other parts
# leave), and only once all its parts are inside the prefix -- on whichever of its parts' engines the adversary likes.
# H[p][e] = the most workgroups the launch can hold on engine e after prefix p.  The launch's next workgroup needs engine p % 4.
# One launch alone deadlocks if H[p][p % 4] >= capacity for some p < n; two launches A, B in flight (the two chains of a stage,
# dispatched in ANY interleaving) deadlock if there are prefixes pa, pb with BOTH next workgroups' engines full of held
# workgroups: H_A[pa][ea] + H_B[pb][ea] >= cap and H_A[pa][eb] + H_B[pb][eb] >= cap.  (Sufficient for safety: everything that is
# not held finishes by itself and frees its CU.)  Row blocks that span XCDs (the row-major orders of few-row launches) are
# outside this per-XCD argument: for those the test asks for plain co-residency of both launches.
import numpy as np

SE, CUS_PER_SE = 4, 8


def _xcd_lists(entries, nblk):
    """entries: {block id: (rb, tile key, n_parts)}; -> per XCD the slot list [(rb, tile, n_parts) or None (padding)]"""
    out = [[] for _ in range(8)]
    for b in range(nblk):
        out[b % 8].append(entries.get(b))
    return out


def _held_bound(slots, rb_sizes):
    """H[p][e] for p = 0 .. n (see above).  rb_sizes: workgroups of each row block in the WHOLE launch (all XCDs)."""
    n = len(slots)
    H = np.zeros((n + 1, SE), dtype=np.int32)
    seen_rb, seen_tile, tile_engines = {}, {}, {}
    for p in range(1, n + 1):
        s = slots[p - 1]
        if s is not None:
            rb, tile, parts = s
            seen_rb[rb] = seen_rb.get(rb, 0) + 1
            seen_tile[tile] = seen_tile.get(tile, 0) + 1
            tile_engines.setdefault(tile, set()).add((p - 1) % SE)
        h = np.zeros(SE, dtype=np.int32)
        for tile, cnt in seen_tile.items():
            rb, parts = tile[0], tile[2]
            if cnt == parts and seen_rb[rb] < rb_sizes[rb]:  # a complete tile of an incomplete row block: one held workgroup
                for e in tile_engines[tile]:
                    h[e] += 1
        H[p] = h
    return H


def _deadlock_alone(H, cap):
    n = H.shape[0] - 1
    return any(H[p][p % SE] >= cap for p in range(n))


def _deadlock_pair(HA, HB, cap):
    nA, nB = HA.shape[0] - 1, HB.shape[0] - 1
    for ea in range(SE):
        pa = np.arange(ea, nA, SE)
        if pa.size == 0:
            continue
        for eb in range(SE):
            pb = np.arange(eb, nB, SE)
            if pb.size == 0:
                continue
            blocked_a = (HA[pa][:, ea][:, None] + HB[pb][:, ea][None, :]) >= cap
            blocked_b = (HA[pa][:, eb][:, None] + HB[pb][:, eb][None, :]) >= cap
            if (blocked_a & blocked_b).any():
                return True
    return False


def _launch_model(lib, nrowblk, ncolblk, nk, ksplit, tile_fn=None):
    """Per-XCD held bounds of one fused-epilogue launch of the wide tile (one workgroup per CU) as the host would order it:
    K-split tail when `ksplit` and the rule says so, else XCD-local row blocks (>= 16 row blocks) or row-major."""
    entries, nblk, spans = {}, 0, False
    o = w_order(lib, nrowblk, ncolblk, nk=nk, ovh=12000 // (nk + 25)) if ksplit else {"parts": 1}
    if o["parts"] > 1:
        nblk = o["nblk"]
        rb, cb, part, slot = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        for b in range(nblk):
            got = (tile_fn or _w_tile)(lib, b, nrowblk, ncolblk, o)
            if got:
                r, c, _, sl = got
                entries[b] = (r, (r, c, o["parts"] if sl >= 0 else 1), o["parts"] if sl >= 0 else 1)
    else:
        so = order(lib, nrowblk, ncolblk, 1 if nrowblk >= 16 else 0, 1)
        nblk, spans = so["nblk"], so["xcd_rows"] == 0
        for b, r, c in tiles(lib, nrowblk, ncolblk, so):
            entries[b] = (r, (r, c, 1), 1)
    rb_sizes = {}
    for r, tile, parts in entries.values():
        rb_sizes[r] = rb_sizes.get(r, 0) + 1
    lists = _xcd_lists(entries, nblk)
    return [(_held_bound(l, rb_sizes), sum(s is not None for s in l)) for l in lists], spans


def _w_tile(lib, b, nrowblk, ncolblk, o):
    rb, cb, part, slot = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    if lib.w_tile(b, nrowblk, ncolblk, o["full"], o["parts"], o["max_tail"], ctypes.byref(rb), ctypes.byref(cb), ctypes.byref(part), ctypes.byref(slot)):
        return rb.value, cb.value, part.value, slot.value
    return None


def _w_tile_tile_major(lib, b, nrowblk, ncolblk, o):
    """Round 5's FIRST tail order (r5c): the K-parts of a tile on consecutive slots -- part p of every tile on engine p."""
    x, t = b & 7, b >> 3
    if t < o["full"]:
        return _w_tile(lib, b, nrowblk, ncolblk, o)
    u = t - o["full"]
    i, part = o["full"] + u // o["parts"], u % o["parts"]
    # the shipped map, asked for tile i's FIRST part, tells which tile that is
    g, r = (i - o["full"]) // 4, (i - o["full"]) % 4
    got = _w_tile(lib, 8 * (o["full"] + g * 4 * o["parts"] + r) + x, nrowblk, ncolblk, o)
    return None if got is None else (got[0], got[1], part, got[3])


# the GEMM shapes of a stage: (column blocks of 256, k16-slabs) of the hidden layers / the SimNorm output layer
C3 = [(7, 112), (7, 50), (3, 112)]
C4 = [(16, 256), (16, 88), (6, 256)]


@pytest.mark.parametrize("model,shapes,rows_per_plan", [("c3", C3, 512), ("c4", C4, 1024)])
def test_no_order_the_library_picks_can_deadlock_in_the_dispatcher_model(lib, model, shapes, rows_per_plan):
    cap = CUS_PER_SE  # the wide tile: one workgroup per CU
    proven, coresident = 0, 0
    for E in range(1, 65):
        if (E * rows_per_plan) % 256:
            continue  # the wide tile takes whole 256-row blocks
        nrowblk = E * rows_per_plan // 256
        for ksplit in (False, True):
            launches = [_launch_model(lib, nrowblk, nc, nk, ksplit) for nc, nk in shapes]
            for i, (la, spans_a) in enumerate(launches):
                for x in range(8):
                    assert not _deadlock_alone(la[x][0], cap), (model, E, ksplit, shapes[i], x)
                for j, (lb, spans_b) in enumerate(launches):
                    if spans_a or spans_b:
                        # row blocks on several XCDs (row-major orders of launches with < 16 row blocks): the per-XCD argument does
                        # not apply; such a pair is only claimed safe when both launches fit the chip together
                        if all(la[x][1] + lb[x][1] <= SE * cap for x in range(8)):
                            coresident += 1
                        continue
                    for x in range(8):
                        assert not _deadlock_pair(la[x][0], lb[x][0], cap), (model, E, ksplit, shapes[i], shapes[j], x)
                    proven += 1
    assert proven > 100
    print(f"[{model}] dispatcher model: {proven} launch pairs proven per XCD, {coresident} row-major pairs co-resident")


def test_the_dispatcher_model_reports_the_orders_that_deadlocked_on_the_gpu(lib):
    """Negative controls.  (i) r5c: the K-parts of a tile on consecutive slots -- the 317M model (16 column blocks in 4 parts = 64
    workgroups per row block and XCD; two plans: one row block per XCD) stopped at 8 or 15 of 16 arrivals with ONE stream, on every
    launch with more than 32 workgroups per XCD.  (ii) r5g: all column blocks of a row block
    on one engine (four row blocks of an XCD slot-interleaved) -- the 48M model's two chains deadlocked on every stage."""
    cap = CUS_PER_SE
    # (i)
    good, _ = _launch_model(lib, 12, 16, 256, True)  # three plans: 24 tiles x 4 parts on every XCD
    bad, _ = _launch_model(lib, 12, 16, 256, True, tile_fn=_w_tile_tile_major)
    assert good[0][1] == 96
    assert not any(_deadlock_alone(good[x][0], cap) for x in range(8))
    assert any(_deadlock_alone(bad[x][0], cap) for x in range(8))
    # (ii) 60 row blocks x 7 column blocks, XCD-local; engine-local variant: slot t of an XCD -> row block 4 (t // 28) + t % 4, column t // 4 % 7
    nrowblk, ncolblk = 60, 7
    entries = {}
    for x in range(8):
        rbs = list(range(x, nrowblk, 8))
        for t in range(-(-len(rbs) // 4) * 4 * ncolblk):
            k = 4 * (t // (4 * ncolblk)) + t % 4
            if k < len(rbs):
                entries[8 * t + x] = (rbs[k], (rbs[k], (t // 4) % ncolblk, 1), 1)
    rb_sizes = {r: ncolblk for r in range(nrowblk)}
    nblk = 8 * (max(entries) // 8 + 1)
    se_local = [_held_bound(l, rb_sizes) for l in _xcd_lists(entries, nblk)]
    shipped, _ = _launch_model(lib, nrowblk, ncolblk, 112, False)
    assert not any(_deadlock_pair(shipped[x][0], shipped[x][0], cap) for x in range(8))
    assert any(_deadlock_pair(se_local[x], se_local[x], cap) for x in range(8))
