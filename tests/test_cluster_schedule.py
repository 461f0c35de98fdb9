"""CPU: the exchange-tile schedule of the cluster path (tdmpc2_amd/csrc/cluster_kernels.cuh) is race-free by construction.

Members of a cluster exchange every layer's raw sums through a small set of tiles S0..S5 in L2, with one hand-over
(arrival words + wait) per exchange.  A member may run ahead of another by at most the distance between two hand-overs,
so the rule that makes tile reuse safe is:

    a tile exchanged at hand-over j and read by every member before that member ARRIVES at a hand-over r (r > j)
    may be rewritten only by writes that belong to hand-over w >= r + 1,

because a member that writes for hand-over w has passed hand-over w - 1 >= r, i.e. every member has arrived at r and is done
reading.  (Writes "belong to" the next hand-over after them in program order.)

This test restates the kernel's program order for all four shapes of a launch -- plain, with the in-launch policy prior
(cluster 0 of a plan, first launch), episodic, and both -- and checks the rule for every write, plus the same rule for the
head tile S4 (written by the head's producers, read by its consumers before their next hand-over).  The restatement follows the
kernel body line by line; if the kernel's slot numbers change, this file has to change with it (the GPU parity tests would
catch a wrong schedule only as a rare flake, which is why it is pinned here as well).
"""
import itertools

import pytest


def launch_program(H, pifold, ep):
    """The kernel body of ks_rollout_cl<APAD, EP> as a list of events:
    ("write", slot) raw sums stored for the NEXT hand-over; ("handover",); ("read", slot) exchange tile -> registers;
    ("head",) a narrow head: producers write S4 and signal on the head words, consumers read S4 (no cluster hand-over)."""
    ev = []

    def gemm(*slots):
        for s in slots:
            ev.append(("write", s))
        ev.append(("handover",))

    def layer(slot):  # cl_layer: contraction, hand-over, epilogue from the same tile
        gemm(slot)
        ev.append(("read", slot))

    for t in range(H):
        term_step = ep and t > 0
        if pifold:
            layer(0)  # pi.l0 -> S0
            layer(3)  # pi.l1 -> S3
            ev.append(("head",))
        gemm(0, 1)  # dyn.l0 -> S0 (parked), rew.l0 -> S1
        if term_step:
            gemm(5)  # term.l0 -> S5 (parked)
        ev.append(("read", 1))
        layer(2)  # rew.l1 -> S2
        ev.append(("head",))
        if term_step:
            ev.append(("read", 5))
            layer(3)  # term.l1 -> S3
            ev.append(("head",))
        ev.append(("read", 0))
        layer(1)  # dyn.l1 -> S1
        layer(2)  # dyn.l2 -> S2
    if ep:
        gemm(5)  # term.l0(z_H) -> S5
    layer(0)  # pi.l0 -> S0
    layer(1)  # pi.l1 -> S1
    ev.append(("head",))
    if ep:
        ev.append(("read", 5))
        layer(0)  # term.l1(z_H) -> S0
        ev.append(("head",))
    gemm(2, 3)  # q1.l0 -> S2 (parked), q0.l0 -> S3
    ev.append(("read", 3))
    layer(0)  # q0.l1 -> S0
    ev.append(("head",))
    ev.append(("read", 2))
    layer(1)  # q1.l1 -> S1
    ev.append(("head",))
    return ev


def check(ev):
    """Returns the number of hand-overs; raises AssertionError on a reuse that the rule does not cover."""
    # annotate every event with the index of the last hand-over before it (0 = none yet)
    k = 0
    exchanged_at = {}  # slot -> hand-over that published its current content
    last_read_before = {}  # slot -> hand-over the readers arrive at next, after their read of the current content
    pending = {}  # slot -> True while written but not yet handed over
    head_reads_before = None
    for i, e in enumerate(ev):
        if e[0] == "write":
            s = e[1]
            w = k + 1  # this write belongs to the next hand-over
            if s in exchanged_at:
                assert s in last_read_before, f"event {i}: S{s} rewritten before anybody read it"
                assert w >= last_read_before[s] + 1, (
                    f"event {i}: S{s} (read before arrival at hand-over {last_read_before[s]}) rewritten for hand-over {w}")
            pending[s] = True
        elif e[0] == "handover":
            k += 1
            for s in list(pending):
                exchanged_at[s] = k
                last_read_before.pop(s, None)
                del pending[s]
        elif e[0] == "read":
            s = e[1]
            assert s in exchanged_at and s not in pending, f"event {i}: S{s} read before it was handed over"
            last_read_before[s] = k + 1  # the reader arrives at hand-over k + 1 only after this read
        elif e[0] == "head":
            # producers write S4 after passing hand-over k; the previous head's consumers finished before arriving at a
            # hand-over <= k: two heads must be separated by at least one cluster hand-over
            assert head_reads_before is None or k >= head_reads_before, f"event {i}: two heads without a hand-over between them"
            head_reads_before = k + 1
    return k


@pytest.mark.parametrize("H,pifold,ep", list(itertools.product([1, 2, 3, 5], [False, True], [False, True])))
def test_exchange_tiles_are_never_rewritten_under_a_reader(H, pifold, ep):
    n = check(launch_program(H, pifold, ep))
    # the phase numbers of a launch start at iter * cl_phases(H): the bound must cover every shape
    assert n <= 8 * H + 7
    heads = sum(1 for e in launch_program(H, pifold, ep) if e[0] == "head")
    assert heads <= 3 * H + 4


def test_the_checker_catches_a_bad_schedule():
    """dyn.l1 into the tile that still holds the parked dyn.l0 sums of a slower member would be a race."""
    ev = [("write", 0), ("write", 1), ("handover",), ("read", 1), ("write", 0), ("handover",), ("read", 0)]
    with pytest.raises(AssertionError):
        check(ev)
    # and the minimal distance the rule allows passes
    check([("write", 0), ("handover",), ("read", 0), ("write", 1), ("handover",), ("read", 1), ("write", 0), ("handover",)])


def test_restatement_matches_the_kernel_source():
    """Cheap guard against drift: the slot literals of the kernel body appear in the order restated above (plain shape)."""
    import os
    import re

    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tdmpc2_amd", "csrc", "cluster_kernels.cuh")).read()
    body = src[src.index("void ks_rollout_cl("):]
    calls = re.findall(r"cl_(gemm|layer<\d>|epi<\d>)\(c, x, (.*?)\);", body, flags=re.S)
    slots = []
    for kind, args in calls:
        a = [x.strip() for x in re.sub(r"\s+", " ", args).split(",")]
        if kind == "gemm":
            # cl_gemm(c, x, wref(la), slot_a, lb, slot_b, kb0, kb1)
            slots.append(("gemm", a[1], a[3]))
        elif kind.startswith("epi"):
            slots.append(("epi", a[0]))
        else:
            # cl_layer<ACT>(c, x, CL_L(L) [= 1 macro argument], bias, kb0, kb1, slot, ...)
            slots.append(("layer", a[4]))
    want = [("layer", "0"), ("layer", "3"),  # policy prior
            ("gemm", "0", "1"), ("gemm", "5", "0"), ("epi", "1"), ("layer", "2"), ("epi", "5"), ("layer", "3"), ("epi", "0"),
            ("layer", "1"), ("layer", "2"),  # step
            ("gemm", "5", "0"), ("layer", "0"), ("layer", "1"), ("epi", "5"), ("layer", "0"),  # value: termination, policy
            ("gemm", "2", "3"), ("epi", "3"), ("layer", "0"), ("epi", "2"), ("layer", "1")]
    assert slots == want, slots


# ---------------------------------------------------------------------------------------------------- two clusters per tile
MAXH = 8


def launch_program_cl2(H, pifold, role):
    """ks_rollout_cl2 (tdmpc2_amd/csrc/cluster2_kernels.cuh): the program order of one cluster -- role 0 = D (dynamics, policy,
    first Q head), role 1 = R (rewards, second Q head).  Own exchange tiles: 0 / 1 layers, 4 head, 5 policy head, 6 + t: Z[t],
    6 + MAXH + t: the prior policy head of step t.  ("peer_read", slot): a tile of the OTHER cluster."""
    ev = []

    def layer(slot):
        ev.append(("write", slot))
        ev.append(("handover",))
        ev.append(("read", slot))

    if role == 0:
        for t in range(H):
            if pifold:
                layer(0)  # pi.l0
                layer(1)  # pi.l1
                ev.append(("head", 6 + MAXH + t))
            layer(0)  # dyn.l0
            layer(1)  # dyn.l1
            layer(6 + t)  # dyn.l2 -> Z[t]
        layer(0)  # pi.l0
        layer(1)  # pi.l1
        ev.append(("head", 5))
        layer(0)  # q0.l0
        layer(1)  # q0.l1
        ev.append(("head", 4))
    else:
        for t in range(H):
            if t > 0:
                ev.append(("peer_read", 6 + t - 1))
            if pifold:
                ev.append(("peer_read", 6 + MAXH + t))
            layer(0)  # rew.l0
            layer(1)  # rew.l1
            ev.append(("head", 4))
        ev.append(("peer_read", 6 + H - 1))
        ev.append(("peer_read", 5))
        layer(0)  # q1.l0
        layer(1)  # q1.l1
        ev.append(("head", 4))
    return ev


@pytest.mark.parametrize("H,pifold", list(itertools.product([1, 2, 3, 5, 8], [False, True])))
def test_two_cluster_schedule(H, pifold):
    """Each cluster's own tiles follow the reuse rule; every tile the OTHER cluster reads is written exactly once per launch (so a
    reader that lags by any number of hand-overs still finds it), and the reader only names tiles its peer really writes."""
    progs = [launch_program_cl2(H, pifold, r) for r in (0, 1)]
    for prog in progs:
        own = [(e[0], e[1]) if e[0] in ("write", "read") else (("head",) if e[0] == "head" else e) for e in prog if e[0] != "peer_read"]
        n = check(own)
        assert n <= 8 * H + 7
        assert sum(1 for e in prog if e[0] == "head") <= 3 * H + 4
    d_writes = [e[1] for e in progs[0] if e[0] == "write"] + [e[1] for e in progs[0] if e[0] == "head"]
    for slot in {e[1] for e in progs[1] if e[0] == "peer_read"}:
        assert d_writes.count(slot) == 1, (slot, d_writes)
    assert max(d_writes) < 6 + 2 * MAXH
    # R's own head tile: two heads are separated by a hand-over (the checker's rule), and D's head tiles 4 / 5 / prior are distinct
    d_heads = [e[1] for e in progs[0] if e[0] == "head"]
    assert len(set(d_heads)) == len(d_heads)


def test_two_cluster_restatement_matches_the_kernel_source():
    """The slot expressions of cluster2_kernels.cuh in program order (D then R), against the restatement above."""
    import os
    import re

    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tdmpc2_amd", "csrc", "cluster2_kernels.cuh")).read()
    body = src[src.index("void ks_rollout_cl2("):]
    d_body, r_body = body[:body.index("R: rewards, second Q head")], body[body.index("R: rewards, second Q head"):]
    d_slots = re.findall(r"cl_layer<\d>\(c, x, CL_L\(p\.(\w+(?:\[q\d\])?)\.l\[(\d)\]\), [^;]*?, (?:KBA|ZKB16), ([^,;]+),", d_body)
    assert [(n, l, s.strip()) for n, l, s in d_slots] == [
        ("pi", "0", "0"), ("pi", "1", "1"), ("dyn", "0", "0"), ("dyn", "1", "1"), ("dyn", "2", "6 + t"),
        ("pi", "0", "0"), ("pi", "1", "1"), ("q[q0]", "0", "0"), ("q[q0]", "1", "1")]
    r_slots = re.findall(r"cl_layer<\d>\(c, x, CL_L\(p\.(\w+(?:\[q\d\])?)\.l\[(\d)\]\), [^;]*?, (?:KBA|ZKB16), ([^,;]+),", r_body)
    assert [(n, l, s.strip()) for n, l, s in r_slots] == [("rew", "0", "0"), ("rew", "1", "1"), ("q[q1]", "0", "0"), ("q[q1]", "1", "1")]
    assert re.findall(r"cl_epi<1>\(c, xp, ([^,]+),", r_body) == ["6 + t - 1", "6 + p.H - 1"]
    assert "cl_head_logits(c, x, p.pi.l[2], true, 6 + MAXH + t)" in d_body and "cl_head_logits(c, x, p.pi.l[2], true, 5)" in d_body
    assert "peer_head_to_staging(6 + MAXH + t" in r_body and "peer_head_to_staging(5," in r_body
