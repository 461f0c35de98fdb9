"""The two hand-written rings, replayed on the CPU: g_gemm_w's DMA ring (tdmpc2_amd/csrc/layered_wide.cuh; first part) and the
register ring of the fused family's contraction loop (fused_kernels.cuh: kloop_asm, the loop of the benched ks_rollout; second part).

The kernel's main loop overlaps three things per k16-slab: the DMA request of slab s + NS into the ring slot slab s has just
left (`global_load_lds`, counted by vmcnt), the LDS reads of slab s + 1 into the other register set, and the MFMAs of slab s.
What it may leave in flight at each barrier is arithmetic on (s, nk): tile_order.h's gw_* helpers, which the kernel calls.
This test compiles those helpers with g++ and replays the ring of ONE wave for every contraction length (a K-split part can be
any length from 1 up), with the memory system's one guarantee -- requests complete in issue order, `s_waitcnt vmcnt(n)` returns
when at most n are outstanding -- and an adversary that lands nothing before it must:

* a slab is read from LDS only after all of its requests have landed (and after the barrier that follows the wait);
* a ring slot is overwritten only after the slab in it has been read, and never while its own requests are in flight;
* every slab is multiplied exactly once, in order; nothing is in flight at the end;
* the waits are as weak as they can be: one request more in flight at any wait would break the first property
  (the counted waits are not conservative vmcnt(0)s in disguise).
"""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tdmpc2_amd", "csrc")
SHIM = r"""
#include "tile_order.h"
extern "C" int prologue_slabs(int nk, int ns) { return gw_prologue_slabs(nk, ns); }
extern "C" int prologue_vmcnt(int npro) { return gw_prologue_vmcnt(npro); }
extern "C" int steady_trip(int s, int nk, int ns, int u) { return gw_steady_trip(s, nk, ns, u) ? 1 : 0; }
extern "C" int steady_vmcnt(int ns) { return gw_steady_vmcnt(ns); }
extern "C" void tail_step(int ss, int nk, int ns, int *out) {
    const GwTailStep t = gw_tail_step(ss, nk, ns);
    out[0] = t.issue; out[1] = t.next; out[2] = t.vmc;
}
extern "C" int phase_vmcnt(int steady, int vmc, int ns) { return gw_phase_vmcnt(steady != 0, vmc, ns); }
"""
REQ = 4  # DMA requests per slab and wave: A and W, two 1 KiB planes each


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("ring")
    (d / "shim.cpp").write_text(SHIM)
    so = d / "libring.so"
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I", CSRC, str(d / "shim.cpp"), "-o", str(so)], check=True)
    return ctypes.CDLL(str(so))


def _constants():
    src = open(os.path.join(CSRC, "layered_wide.cuh")).read()
    ns = int(re.search(r"constexpr int GW_NSLOT = (\d+);", src).group(1))
    u = 2 * ns if ns % 2 else ns
    # the kernel really calls the helpers this test replays
    for name in ("gw_prologue_slabs(nk, NS)", "gw_steady_trip(s, nk, NS, GW_U)", "gw_tail_step(s + PH, nk, NS)", "gw_phase_vmcnt(STEADY, vmc, NS)",
                 "gw_steady_vmcnt(NS)"):
        assert name in src, name
    return ns, u


class Ring:
    """one wave's view: `fly` = issued requests not yet known to have landed (oldest first), `slot[i]` = slab in ring slot i"""

    def __init__(self, ns, slack=0):
        self.ns, self.slack = ns, slack
        self.fly, self.slot, self.landed, self.read = [], [None] * ns, set(), set()

    def request(self, slab):
        i = slab % self.ns
        old = self.slot[i]
        assert old is None or old in self.read, f"slab {slab} overwrites slab {old}, which was never read"
        assert all(f != old for f in self.fly), f"slab {slab} requested into a slot whose slab {old} is still landing"
        self.slot[i] = slab
        self.fly += [slab] * REQ

    def wait(self, n):
        n += self.slack  # the adversary: only what the wait forces has landed
        while len(self.fly) > n:
            s = self.fly.pop(0)
            if s not in self.fly:
                self.landed.add(s)

    def lds_read(self, slab):
        assert self.slot[slab % self.ns] == slab, f"slab {slab} is not in its slot"
        assert slab in self.landed, f"slab {slab} read before its requests have landed"
        self.read.add(slab)


def replay(lib, nk, ns, u, slack=0):
    """-> slabs in the order their MFMAs ran"""
    r = Ring(ns, slack)
    npro = lib.prologue_slabs(nk, ns)
    for d in range(npro):
        r.request(d)
    ladder = {5: 16, 4: 12, 3: 8, 2: 4}  # the kernel's prologue ladder over npro ...
    pv = ladder.get(min(npro, 5), 0)
    assert pv == min(lib.prologue_vmcnt(npro), 16)  # ... is gw_prologue_vmcnt
    r.wait(pv)
    r.lds_read(0)  # after the barrier
    done, s, out = [], 0, (ctypes.c_int * 3)()

    def phase(ss, steady, issue, nxt, vmc):
        # top of the phase: LDS reads of slab ss (issued in the phase before) complete (lgkmcnt(0)), then the counted wait, barrier
        r.wait(lib.phase_vmcnt(1 if steady else 0, vmc, ns))
        if issue:
            r.request(ss + ns)  # into the slot slab ss has just left: every wave has its copy in registers (barrier)
        if nxt:
            r.lds_read(ss + 1)
        assert ss in r.read
        done.append(ss)

    while lib.steady_trip(s, nk, ns, u):
        for ph in range(u):
            phase(s + ph, True, True, True, lib.steady_vmcnt(ns))
        s += u
    while s < nk:
        for ph in range(u):
            if s + ph < nk:
                lib.tail_step(s + ph, nk, ns, out)
                phase(s + ph, False, bool(out[0]), bool(out[1]), out[2])
        s += u
    assert not r.fly or all(f in r.landed for f in r.fly) or slack, "requests in flight at the end of the loop"
    r.wait(0)
    assert not r.fly
    return done


@pytest.mark.parametrize("nk", list(range(1, 41)) + [63, 64, 65, 111, 112, 113, 256])
def test_every_slab_lands_before_it_is_read_and_is_multiplied_once(lib, nk):
    ns, u = _constants()
    assert replay(lib, nk, ns, u) == list(range(nk))


def test_nothing_is_left_in_flight_when_the_loop_ends(lib):
    # the K-split tail reuses the ring's LDS for its ticket right behind the loop: the last phase must have waited for everything
    ns, u = _constants()
    for nk in range(1, 40):
        r_out = (ctypes.c_int * 3)()
        lib.tail_step(nk - 1, nk, ns, r_out)
        assert (r_out[0], r_out[1], r_out[2]) == (0, 0, 0), nk
        assert lib.phase_vmcnt(0, 0, ns) == 0


@pytest.mark.parametrize("nk", [2, 3, 4, 5, 6, 9, 12, 37, 64])
def test_the_counted_waits_are_tight(lib, nk):
    """one more request in flight at every wait and a slab is read before it has landed"""
    ns, u = _constants()
    with pytest.raises(AssertionError, match="read before its requests have landed"):
        replay(lib, nk, ns, u, slack=1)


def test_ring_of_five_slots_would_also_be_scheduled_correctly(lib):
    # GW_NSLOT = 5 was measured (profiles/README.md r4c): the arithmetic is written for any depth the wait ladders cover
    for nk in range(1, 30):
        ns, u = 5, 10
        done = []
        ring = Ring(ns)
        npro = lib.prologue_slabs(nk, ns)
        for d in range(npro):
            ring.request(d)
        ring.wait(min(lib.prologue_vmcnt(npro), 16))
        ring.lds_read(0)
        out = (ctypes.c_int * 3)()
        for ss in range(nk):  # all phases through the tail arithmetic
            lib.tail_step(ss, nk, ns, out)
            ring.wait(out[2])
            if out[0]:
                ring.request(ss + ns)
            if out[1]:
                ring.lds_read(ss + 1)
            done.append(ss)
        ring.wait(0)
        assert done == list(range(nk))


# ------------------------------------------------------------------------------------------------------------------------
# g_gemm_m's ring (layered_mid.cuh, round 6: the few-row path): the same schedule with 6 requests per wave and k32-slab and THREE slots.
MSHIM = r"""
#include "tile_order.h"
extern "C" int req() { return GM_REQ; }
extern "C" int prologue_slabs(int nk, int ns) { return gm_prologue_slabs(nk, ns); }
extern "C" int prologue_vmcnt(int npro) { return gm_prologue_vmcnt(npro); }
extern "C" int steady_trip(int s, int nk, int ns, int u) { return gm_steady_trip(s, nk, ns, u) ? 1 : 0; }
extern "C" int steady_vmcnt(int ns) { return gm_steady_vmcnt(ns); }
extern "C" void tail_step(int ss, int nk, int ns, int *out) {
    const GmTailStep t = gm_tail_step(ss, nk, ns);
    out[0] = t.issue; out[1] = t.next; out[2] = t.vmc;
}
extern "C" int mfmas_per_phase() { return RING_MFMAS_PER_PHASE; }
extern "C" int gw_requests() { return GW_REQ; }
extern "C" int gm_request_behind(int k) { return gm_req_at(k); }
extern "C" int gw_request_behind(int k) { return gw_req_at(k); }
"""


@pytest.fixture(scope="module")
def mlib(tmp_path_factory):
    d = tmp_path_factory.mktemp("mring")
    (d / "shim.cpp").write_text(MSHIM)
    so = d / "libmring.so"
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I", CSRC, str(d / "shim.cpp"), "-o", str(so)], check=True)
    return ctypes.CDLL(str(so))


def _m_constants(mlib):
    src = open(os.path.join(CSRC, "layered_mid.cuh")).read()
    ns = int(re.search(r"constexpr int GM_NS = (\d+);", src).group(1))
    u = int(re.search(r"constexpr int GM_U = (\d+);", src).group(1))
    assert u % ns == 0 and u % 2 == 0  # whole turns of the ring and of the two register sets per unrolled trip
    req = mlib.req()
    assert "gm_wait_vm<(GM_NS - 2) * GM_REQ>()" in src and mlib.steady_vmcnt(ns) == (ns - 2) * req  # the steady wait of gm_phase
    for name in ("gm_prologue_slabs(nk, GM_NS)", "gm_steady_trip(s, nk, GM_NS, GM_U)", "gm_tail_step(s + PH, nk, GM_NS)"):
        assert name in src, name
    # the DMA requests one gm_issue makes are GM_REQ
    body = src[src.index("__device__ __forceinline__ void gm_issue("):]
    body = body[:body.index("\n}\n")]
    assert body.count("gw_glds(") == req
    return ns, u, req


def m_replay(mlib, nk, ns, u, req, slack=0):
    global REQ
    keep, REQ = REQ, req  # (Ring.request counts REQ requests per slab)
    try:
        r = Ring(ns, slack)
        npro = mlib.prologue_slabs(nk, ns)
        for d in range(npro):
            r.request(d)
        assert mlib.prologue_vmcnt(npro) == req * (npro - 1)  # the kernel's ladder over npro = 3, 2, 1: 2, 1, 0 x GM_REQ
        r.wait(mlib.prologue_vmcnt(npro))
        r.lds_read(0)
        done, s, out = [], 0, (ctypes.c_int * 3)()

        def phase(ss, steady, issue, nxt, vmc):
            # gm_phase's ladder (a ring of three: at most one newer slab in flight): steady or vmc >= REQ -> (NS - 2) REQ; else 0
            assert ns == 3
            w = (ns - 2) * req if (steady or vmc >= req) else 0
            r.wait(w)
            if issue:
                r.request(ss + ns)
            if nxt:
                r.lds_read(ss + 1)
            assert ss in r.read
            done.append(ss)

        while mlib.steady_trip(s, nk, ns, u):
            for ph in range(u):
                phase(s + ph, True, True, True, mlib.steady_vmcnt(ns))
            s += u
        while s < nk:
            for ph in range(u):
                if s + ph < nk:
                    mlib.tail_step(s + ph, nk, ns, out)
                    phase(s + ph, False, bool(out[0]), bool(out[1]), out[2])
            s += u
        assert not r.fly or slack, "requests in flight at the end of the loop (the epilogue reuses the ring's LDS)"
        r.wait(0)
        return done
    finally:
        REQ = keep


@pytest.mark.parametrize("nk", list(range(1, 45)) + [50, 88, 111, 112, 113, 256])
def test_g_gemm_m_every_slab_lands_before_it_is_read_and_is_multiplied_once(mlib, nk):
    """any K-part length from one k32-slab up (the t = 0 first layers contract 1 slab; the 317M model's hidden layers 128)"""
    ns, u, req = _m_constants(mlib)
    assert m_replay(mlib, nk, ns, u, req) == list(range(nk))


@pytest.mark.parametrize("nk", [2, 3, 6, 7, 8, 13, 28, 64])
def test_g_gemm_m_counted_waits_are_tight(mlib, nk):
    ns, u, req = _m_constants(mlib)
    with pytest.raises(AssertionError, match="read before its requests have landed"):
        m_replay(mlib, nk, ns, u, req, slack=1)


# ------------------------------------------------------------------------------------------------------------------------
# The fused family's hand-ordered contraction loop (fused_kernels.cuh: kloop_asm -- the loop of the benched ks_rollout): a
# REGISTER ring of KL_RD weight blocks (4 global_load_dwordx4 each) and two activation fragment sets read from LDS a step ahead.
KSHIM = r"""
#include "kloop_schedule.h"
extern "C" int rd() { return KL_RD; }
extern "C" int w_loads() { return KL_W_LOADS; }
extern "C" int steady_trip(int k, int nk, int r) { return kl_steady_trip(k, nk, r) ? 1 : 0; }
extern "C" int steady_vmcnt(int r) { return kl_steady_vmcnt(r); }
extern "C" int behind(int kk, int nk, int r) { return kl_behind(kk, nk, r); }
extern "C" int next_act(int kk, int nk) { return kl_next_act(kk, nk) ? 1 : 0; }
extern "C" int issue(int kk, int nk, int r) { return kl_issue(kk, nk, r) ? 1 : 0; }
"""


@pytest.fixture(scope="module")
def klib(tmp_path_factory):
    d = tmp_path_factory.mktemp("kloop")
    (d / "shim.cpp").write_text(KSHIM)
    so = d / "libkloop.so"
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I", CSRC, str(d / "shim.cpp"), "-o", str(so)], check=True)
    return ctypes.CDLL(str(so))


def test_the_kernel_calls_the_schedule_functions_this_file_replays(klib):
    src = open(os.path.join(CSRC, "fused_kernels.cuh")).read()
    body = src[src.index("void kloop_asm("):src.index("void kloop_s(")]
    for name in ("constexpr int RD = KL_RD;", "kl_steady_trip(k, nk, RD)", "kl_next_act(kk, nk)", "kl_behind(kk, nk, RD)", "kl_issue(kk, nk, RD)"):
        assert name in body, name
    # the steady loop's literal wait is the helper's value; the ladder of the last steps is vmcnt(4 * behind)
    steady = body[body.index("kl_steady_trip"):body.index("// the last RD")]
    assert f's_waitcnt vmcnt({klib.steady_vmcnt(klib.rd())})' in steady
    for b, n in ((5, 20), (4, 16), (3, 12), (2, 8), (1, 4)):
        assert re.search(rf"behind (>=|==) {b}\) asm volatile\(\"s_waitcnt vmcnt\({n}\)\"", body), (b, n)
    assert klib.w_loads() == 4 and body.count("global_load_dwordx4") == 0  # (a_load_w, one asm statement of 4 loads, sits above)
    assert src.count('"global_load_dwordx4 %') == 4


def kloop_replay(klib, nk, slack=0):
    """one wave: -> blocks in the order their 12 MFMAs were issued"""
    rd, wl = klib.rd(), klib.w_loads()
    fly = []            # weight loads issued, oldest first: block ids
    w_landed = set()
    ring = [None] * rd  # block whose fragments (will) sit in ring slot d
    act_req, act_ok = set(), set()  # activation fragments requested from LDS / known to have landed (lgkmcnt(0))
    used = []

    def load_w(slot, blk):
        old = ring[slot]
        assert old is None or old in used, f"ring slot {slot}: block {blk} loaded over block {old} before its MFMAs were issued"
        assert old not in fly, f"ring slot {slot} reloaded while block {old} is still landing"
        ring[slot] = blk
        fly.extend([blk] * wl)

    def vmcnt(n):
        n += slack
        while len(fly) > n:
            b = fly.pop(0)
            if b not in fly:
                w_landed.add(b)

    def mfma(slot, kk):
        assert ring[slot] == kk, f"step {kk}: ring slot {slot} holds block {ring[slot]}"
        assert kk in w_landed, f"step {kk}: weight fragments used before their loads have landed"
        assert kk in act_ok, f"step {kk}: activation fragments used before lgkmcnt(0)"
        used.append(kk)

    for d in range(rd):
        if d < nk:
            load_w(d, d)
    act_req.add(0)
    k = 0
    while klib.steady_trip(k, nk, rd):
        for d in range(rd):
            kk = k + d
            act_ok |= act_req                      # s_waitcnt lgkmcnt(0)
            act_req.add(kk + 1)                    # the other fragment set: block kk - 1's MFMAs were issued a step ago
            assert kk + 1 < nk
            vmcnt(klib.steady_vmcnt(rd))
            mfma(d, kk)
            assert kk + rd < nk
            load_w(d, kk + rd)
        k += rd
    while k < nk:
        for d in range(rd):
            kk = k + d
            if kk < nk:
                act_ok |= act_req
                if klib.next_act(kk, nk):
                    act_req.add(kk + 1)
                vmcnt(wl * klib.behind(kk, nk, rd))
                mfma(d, kk)
                if klib.issue(kk, nk, rd):
                    load_w(d, kk + rd)
        k += rd
    assert not fly, "weight loads in flight when the loop ends (the compiler reuses their destination registers)"
    assert act_req <= act_ok, "LDS reads in flight when the loop ends"
    assert max(act_req) == nk - 1
    return used


@pytest.mark.parametrize("nk", list(range(1, 36)) + [48, 63, 64, 65])
def test_kloop_every_fragment_lands_before_its_mfmas(klib, nk):
    # (the benched layers: K = 512 + action padding -> 32 .. 36 blocks; heads 32; the t = 0 short contraction 1 .. 4)
    assert kloop_replay(klib, nk) == list(range(nk))


@pytest.mark.parametrize("nk", [2, 3, 4, 5, 8, 33, 36])
def test_kloop_counted_waits_are_tight(klib, nk):
    with pytest.raises(AssertionError, match="used before their loads have landed"):
        kloop_replay(klib, nk, slack=1)


@pytest.mark.parametrize("kernel", ["g_gemm_m", "g_gemm_w"])
def test_every_dma_request_of_a_phase_goes_out_exactly_once_between_its_mfmas(mlib, kernel):
    """Round 6: the ring GEMMs issue a phase's LDS-DMA requests BETWEEN its MFMAs (tile_order.h: gm_req_at / gw_req_at -- request r behind
    MFMA k) instead of all of them behind the barrier.  The counted waits replayed above assume GM_REQ / GW_REQ requests per phase and
    wave: every request index must come up exactly once, in order, behind an MFMA that has a successor in the phase -- and the kernel
    source must carry a hook behind each of those MFMAs (a table entry without a hook would silently drop a request: the next phase's
    vmcnt would then wait for a slab that was never asked for... or not wait for one that was)."""
    n = mlib.mfmas_per_phase()
    req, at, fname, hook, mf = ((mlib.req(), mlib.gm_request_behind, "layered_mid.cuh", "GM_AT", "gm_phase") if kernel == "g_gemm_m"
                                else (mlib.gw_requests(), mlib.gw_request_behind, "layered_wide.cuh", "GW_AT", "gw_phase"))
    placed = [(k, at(k)) for k in range(n) if at(k) >= 0]
    assert [r for _, r in placed] == list(range(req)), placed   # each request once, in request order (the last one advances the pointers)
    assert all(k < n - 1 for k, _ in placed)                      # never behind the phase's last MFMA (no hook there)
    assert at(n - 1) < 0 and at(n) < 0
    src = open(os.path.join(CSRC, fname)).read()
    body = src[src.index(f"__device__ __forceinline__ void {mf}("):]
    body = body[:body.index("\n}\n")]
    assert body.count("GW_MFMA(") == n, body.count("GW_MFMA(")   # the phase's MFMAs, written out
    hooks = [int(m) for m in re.findall(hook + r"\((\d+)\)", body)]
    assert hooks == list(range(n - 1)), hooks                    # a hook behind every MFMA but the last, in order
    # and the hook is followed by the next MFMA, i.e. sits between two MFMAs
    for k in range(n - 1):
        i = body.index(f"{hook}({k})")
        assert "GW_MFMA(" in body[i:], k
