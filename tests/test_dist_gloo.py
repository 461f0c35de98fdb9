"""CPU, world_size 2, gloo: the N>1 host path — env sharding, weight broadcast, action gather.
The compute leg is the oracle here (test stand-in only); on GPUs it is the HIP planner."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tdmpc2_amd.dist import shard_range


def test_shard_range_partitions_exactly():
    for n in (1, 7, 64, 512, 513):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_envs, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import cases
        from oracle import planner_oracle as po
        from tdmpc2_amd import synth
        from tdmpc2_amd.dist import broadcast_state_dict, gather_actions

        c = cases.build_case("tiny")
        cfg = c["cfg"]
        # rank 0 holds the real weights, the others start from zeros: broadcast must fix that
        sd = {k: torch.as_tensor(v).clone() for k, v in c["sd"].items()}
        if rank != 0:
            for v in sd.values():
                v.zero_()
        broadcast_state_dict(sd, src=0)
        assert all(torch.equal(sd[k], torch.as_tensor(v)) for k, v in c["sd"].items())
        z0 = synth.make_latents(cfg, n_envs, seed=1)
        tape = synth.make_noise_tape(cfg, n_envs, c["iterations"], seed=2)
        a0, a1 = shard_range(n_envs, world, rank)
        model = po.OracleModel(cfg, sd)
        loc_tape = {k: v[a0:a1] for k, v in tape.items()}
        act, _, _ = po.plan_batch(model, z0[a0:a1], loc_tape, np.zeros((a1 - a0, cfg.horizon, cfg.action_dim), np.float32),
                                  [True] * (a1 - a0), False, None, [0.99] * (a1 - a0), c["iterations"])
        full = gather_actions(act, n_envs)
        assert full.shape == (n_envs, cfg.action_dim)
        torch.save(full, os.path.join(out_dir, f"rank{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_planning_equals_single_process(tmp_path):
    n_envs, world = 5, 2  # uneven split: 3 + 2
    mp.spawn(_worker, args=(world, _free_port(), n_envs, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0, r1)
    from oracle import cases
    from oracle import planner_oracle as po
    from tdmpc2_amd import synth

    c = cases.build_case("tiny")
    cfg = c["cfg"]
    model = po.OracleModel(cfg, {k: torch.as_tensor(v) for k, v in c["sd"].items()})
    z0 = synth.make_latents(cfg, n_envs, seed=1)
    tape = synth.make_noise_tape(cfg, n_envs, c["iterations"], seed=2)
    want, _, _ = po.plan_batch(model, z0, tape, np.zeros((n_envs, cfg.horizon, cfg.action_dim), np.float32),
                               [True] * n_envs, False, None, [0.99] * n_envs, c["iterations"])
    assert torch.allclose(r0, want, atol=1e-6)


# ---------------------------------------------------------------- one plan sharded over ranks (tdmpc2_amd/dist.py: sharded_plan)
class OracleShardBackend:
    """Stand-in for NativePlanner's shard_* entry points built from the oracle's pieces (CPU tests only): the same
    replicated prologue / sampling / refit and row-range evaluation, so that `dist.sharded_plan` -- the host logic under
    test -- can run over gloo.  Rows are evaluated in fixed chunks of `chunk` rows whatever the range, so results do not
    depend on how the rows are split over ranks (identical matmul shapes -> bit-identical values)."""

    def __init__(self, case, chunk=32):
        from oracle import planner_oracle as po

        self.po = po
        self.cfg = case["cfg"]
        self.iterations = case["iterations"]
        self.shard_granularity = chunk
        self.chunk = chunk
        self.model = po.OracleModel(self.cfg, {k: torch.as_tensor(v) for k, v in case["sd"].items()})
        self.discount = case["discounts"][0]

    def shard_begin(self, z0, prev_mean, t0, task_emb=None, act_mask=None, tape=None, seed=0):
        cfg, po = self.cfg, self.po
        H, N, P, A = cfg.horizon, cfg.num_samples, cfg.num_pi_trajs, cfg.action_dim
        assert z0.shape[0] == 1
        self.tape = {k: torch.as_tensor(v[0]) for k, v in tape.items()}
        z = z0[:1]
        self.actions = torch.empty(H, N, A)
        _z = z.repeat(P, 1)
        for t in range(H - 1):
            self.actions[t, :P] = self.model.pi(_z, None, self.tape["pi_traj_eps"][t])
            _z = self.model.next(_z, self.actions[t, :P], None)
        self.actions[-1, :P] = self.model.pi(_z, None, self.tape["pi_traj_eps"][H - 1])
        self.mean = torch.zeros(H, A)
        self.std = torch.full((H, A), cfg.max_std)
        if not bool(t0[0]):
            self.mean[:-1] = prev_mean[0, 1:]

    def shard_values(self, it, r0, r1, z0, disc_pow, value, act_mask=None, seed=0):
        cfg, po = self.cfg, self.po
        P = cfg.num_pi_trajs
        self.actions[:, P:] = (self.mean.unsqueeze(1) + self.std.unsqueeze(1) * self.tape["sample_eps"][it]).clamp(-1, 1)
        for c0 in range(r0, r1, self.chunk):
            sl = slice(c0, c0 + self.chunk)
            sub = type("C", (), {})()
            sub.__dict__.update(vars(cfg))
            sub.num_samples = self.chunk
            m = po.OracleModel(sub, self.model.sd)
            v = po.estimate_value(m, z0[:1].repeat(self.chunk, 1), self.actions[:, sl], None, self.discount,
                                  self.tape["pi_eps"][it][sl], self.tape["qidx"][it])
            value[0, sl] = v.squeeze(1)

    def shard_refit(self, it, value, prev_mean, action, act_mask=None, eval_mode=False, seed=0, stages=None):
        cfg, po = self.cfg, self.po
        v, idx, score, elite_actions, self.mean, self.std = po.refit(cfg, value[0].unsqueeze(1).clone(), self.actions, None)
        if stages is not None:
            stages.setdefault("elite_idx", []).append(idx.clone())
            stages.setdefault("value", []).append(value[0].clone())
        if it == self.iterations - 1:
            pick = po.gumbel_select(score.squeeze(1), self.tape["gumbel_exp"])
            a = elite_actions[0, pick]
            if not eval_mode:
                a = a + self.std[0] * self.tape["final_eps"]
            action[0] = a.clamp(-1, 1)
            prev_mean[0] = self.mean


def _shard_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import cases
        from tdmpc2_amd.dist import sharded_plan

        c = cases.build_case("tiny")
        be = OracleShardBackend(c)
        z0 = torch.as_tensor(c["z0"][:1])
        prev = torch.as_tensor(c["prev_mean"][:1]).clone()
        t0 = torch.tensor([0], dtype=torch.uint8)  # warm start
        tape = {k: v[:1] for k, v in c["tape"].items()}
        stages = {}
        a = sharded_plan(be, z0, None, prev, t0, tape=tape, stages=stages)
        torch.save({"action": a, "prev_mean": prev, "idx": torch.stack(stages["elite_idx"]), "value": torch.stack(stages["value"])},
                   os.path.join(out_dir, f"shard{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_plan_sharded_over_two_ranks_is_bit_identical_to_one_rank(tmp_path):
    """N split over 2 ranks, value slices all-gathered per iteration, replicated refit: both ranks end with the
    same action, bit-identical to the unsharded run of the same backend, and equal to the oracle's plan()."""
    from oracle import cases
    from oracle import planner_oracle as po
    from tdmpc2_amd.dist import sharded_plan

    mp.spawn(_shard_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "shard0.pt")
    r1 = torch.load(tmp_path / "shard1.pt")
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k
    c = cases.build_case("tiny")
    be = OracleShardBackend(c)
    prev = torch.as_tensor(c["prev_mean"][:1]).clone()
    tape = {k: v[:1] for k, v in c["tape"].items()}
    stages = {}
    a = sharded_plan(be, torch.as_tensor(c["z0"][:1]), None, prev, torch.tensor([0], dtype=torch.uint8), tape=tape, stages=stages)
    assert torch.equal(a, r0["action"]) and torch.equal(prev, r0["prev_mean"])
    assert torch.equal(torch.stack(stages["elite_idx"]), r0["idx"])
    wa, wpm, st = po.plan(be.model, z0=torch.as_tensor(c["z0"][:1]), tape=po.env_tape(c["tape"], 0),
                          prev_mean=torch.as_tensor(c["prev_mean"][0]), t0=False, eval_mode=False, task=None,
                          discount=c["discounts"][0], iterations=c["iterations"])
    assert torch.allclose(a[0], wa, atol=1e-5) and torch.allclose(prev[0], wpm, atol=1e-5)


class FaultyShardBackend(OracleShardBackend):
    """The stand-in with NativePlanner's fault interface: on `faulty_rank` the first attempt's value slice is garbage (what a
    bounded inter-workgroup wait that gave up leaves behind) and take_fault() reports it once; plan_safely_once() -- which
    sharded_plan calls on EVERY rank before it re-plans (TDMPC2_TUNE_SAFE_ONCE) -- puts exactly the next plan on the kernels that
    cannot fault and leaves the settings asked for (`fused`) alone."""

    def __init__(self, case, faulty):
        super().__init__(case)
        self.faulty, self.fused, self.pending, self.counter, self.log = faulty, True, 0, 7, []
        self.safe_once, self.setter_calls = False, 0

    def shard_begin(self, *a, **kw):
        self.log.append(("begin", self.counter, self.fused and not self.safe_once))
        self.counter += 1
        return super().shard_begin(*a, **kw)

    def shard_refit(self, it, *a, **kw):
        out = super().shard_refit(it, *a, **kw)
        if it == self.iterations - 1:
            self.safe_once = False  # the library drops the flag when the plan it covered has been enqueued
        return out

    def plan_safely_once(self, on=True):
        self.safe_once = bool(on)

    def shard_values(self, it, r0, r1, z0, disc_pow, value, **kw):
        super().shard_values(it, r0, r1, z0, disc_pow, value, **kw)
        if self.faulty and self.fused and not self.safe_once and it == 1:
            value[0, r0:r1] = 123.0
            self.pending += 1

    def take_fault(self):
        n, self.pending = self.pending, 0
        return n

    def call_counter(self):
        return self.counter

    def set_call_counter(self, v):
        self.counter = int(v)

    def set_fuse_ln(self, on):  # the caller's settings: sharded_plan must not touch them
        self.fused = bool(on)
        self.setter_calls += 1

    def set_cluster(self, mode):
        self.setter_calls += 1


def _fault_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import cases
        from tdmpc2_amd.dist import sharded_plan

        c = cases.build_case("tiny")
        be = FaultyShardBackend(c, faulty=(rank == 1))
        prev = torch.as_tensor(c["prev_mean"][:1]).clone()
        tape = {k: v[:1] for k, v in c["tape"].items()}
        a = sharded_plan(be, torch.as_tensor(c["z0"][:1]), None, prev, torch.tensor([0], dtype=torch.uint8), tape=tape, seed=3)
        torch.save({"action": a, "prev_mean": prev, "retries": torch.tensor(be.last_shard_retries),
                    "log": be.log, "fused": be.fused, "setter_calls": be.setter_calls, "safe_once": be.safe_once},
                   os.path.join(out_dir, f"fault{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_a_fault_on_one_rank_makes_every_rank_replan_once(tmp_path):
    """dist.sharded_plan: rank 1's slice of iteration 1 is garbage and only rank 1 knows.  The verdict is all-reduced, BOTH
    ranks switch kernels for the retry, restore prev_mean and the Philox call counter and plan the step again; the result is the
    plan of a run that never faulted."""
    from oracle import cases
    from tdmpc2_amd.dist import sharded_plan

    mp.spawn(_fault_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "fault0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "fault1.pt", weights_only=False)
    assert int(r0["retries"]) == 1 and int(r1["retries"]) == 1
    # the retry's safe kernels are a property of the retry (TDMPC2_TUNE_SAFE_ONCE): the caller-facing tuning keys were never
    # touched, the flag is gone afterwards; a rank that really faulted stays downgraded inside the library until it re-arms
    assert r0["fused"] and r1["fused"] and r0["setter_calls"] == 0 and r1["setter_calls"] == 0
    assert not r0["safe_once"] and not r1["safe_once"]
    # two attempts on each rank, the second with the first one's call counter and the safe kernels
    for r in (r0, r1):
        assert [e[0] for e in r["log"]] == ["begin", "begin"]
        assert r["log"][0][1] == r["log"][1][1] and r["log"][0][2] and not r["log"][1][2]
    assert r0["log"][0][1] == r1["log"][0][1]  # (rank 0's counter, adopted by rank 1)
    assert torch.equal(r0["action"], r1["action"]) and torch.equal(r0["prev_mean"], r1["prev_mean"])
    c = cases.build_case("tiny")
    be = FaultyShardBackend(c, faulty=False)
    prev = torch.as_tensor(c["prev_mean"][:1]).clone()
    tape = {k: v[:1] for k, v in c["tape"].items()}
    a = sharded_plan(be, torch.as_tensor(c["z0"][:1]), None, prev, torch.tensor([0], dtype=torch.uint8), tape=tape, seed=3)
    assert be.last_shard_retries == 0
    assert torch.equal(a, r0["action"]) and torch.equal(prev, r0["prev_mean"])


def test_sharded_plan_rejects_ragged_splits():
    from oracle import cases
    from tdmpc2_amd.dist import sharded_plan

    c = cases.build_case("tiny")
    be = OracleShardBackend(c, chunk=48)  # 64 rows do not split into multiples of 48
    with pytest.raises(ValueError, match="does not split"):
        sharded_plan(be, torch.as_tensor(c["z0"][:1]), None, torch.zeros(1, 2, 4), torch.tensor([1], dtype=torch.uint8),
                     tape={k: v[:1] for k, v in c["tape"].items()})


def _bcast_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tdmpc2_amd.dist import broadcast_state_dict

        g = torch.Generator().manual_seed(0)
        ref = {"a": torch.randn(1000, generator=g), "b": torch.randn(37, 5, generator=g), "big": torch.randn(5000, generator=g),
               "i": torch.arange(11), "c": torch.randn(3, generator=g)}
        sd = {k: (v.clone() if rank == 0 else torch.zeros_like(v)) for k, v in ref.items()}
        # a 4 KB bucket: "a" fills one, "big" (20 KB) exceeds it and is broadcast in place, the rest share buckets
        broadcast_state_dict(sd, src=0, bucket_bytes=4096)
        assert all(torch.equal(sd[k], ref[k]) for k in ref)
    finally:
        dist.destroy_process_group()


def test_broadcast_streams_in_buckets():
    mp.spawn(_bcast_worker, args=(2, _free_port()), nprocs=2, join=True)


class PhiloxLikeBackend(FaultyShardBackend):
    """The stand-in WITHOUT a tape: like the library's in-kernel Philox, the noise is a function of (seed, call counter at
    shard_begin) -- ranks whose counters differ sample different actions."""

    def shard_begin(self, z0, prev_mean, t0, task_emb=None, act_mask=None, tape=None, seed=0):
        from tdmpc2_amd import synth

        assert tape is None
        drawn = synth.make_noise_tape(self.cfg, 1, self.iterations, seed=(int(seed) * 1000003 + self.counter) % (2**31))
        return super().shard_begin(z0, prev_mean, t0, task_emb=task_emb, act_mask=act_mask, tape=drawn, seed=seed)


def _stream_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import cases
        from tdmpc2_amd.dist import sharded_plan

        c = cases.build_case("tiny")
        be = PhiloxLikeBackend(c, faulty=False)
        # rank 1's handle has planned on its own in the meantime: its counter ran ahead
        be.counter = 5 if rank == 0 else 9
        prev = torch.as_tensor(c["prev_mean"][:1]).clone()
        z0, t0 = torch.as_tensor(c["z0"][:1]), torch.tensor([0], dtype=torch.uint8)
        a = sharded_plan(be, z0, None, prev, t0, tape=None, seed=(1 << 40) + 7)
        # the counters are compared with the FIRST value all-gather (no round trip of their own): the ranks found out after the
        # plan, adopted rank 0's counter and planned again -- once; from then on they stay in step
        assert be.last_shard_realigns == 1 and be.last_shard_retries == 0 and be.counter == 6
        a2 = sharded_plan(be, z0, None, prev.clone(), t0, tape=None, seed=(1 << 40) + 7)
        assert be.last_shard_realigns == 0 and be.counter == 7
        torch.save({"action": a, "prev_mean": prev, "action2": a2, "log": be.log}, os.path.join(out_dir, f"stream{rank}.pt"))
        # a seed that carries the rank (what TDMPC2._seed does) must be refused on EVERY rank, not diverge silently
        raised = False
        try:
            sharded_plan(be, z0, None, prev.clone(), t0, tape=None, seed=(rank << 32) ^ 3)
        except ValueError as ex:
            raised = "seed differs between ranks" in str(ex)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write(str(int(raised)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_plan_ranks_agree_on_the_philox_stream(tmp_path):
    """Without a tape the ranks of a sharded plan must draw identical noise.  The seed and the call counter ride with the first
    value all-gather (no host round trip of their own): ranks whose counters differ re-plan ONCE with rank
    0's counter -- the result is the plan a single process makes with that counter -- and stay in step afterwards; a
    rank-dependent seed raises everywhere."""
    from oracle import cases
    from tdmpc2_amd.dist import sharded_plan

    mp.spawn(_stream_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").read_text() == "1" and (tmp_path / "ok1").read_text() == "1"
    r0 = torch.load(tmp_path / "stream0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "stream1.pt", weights_only=False)
    assert torch.equal(r0["action"], r1["action"]) and torch.equal(r0["prev_mean"], r1["prev_mean"]) and torch.equal(r0["action2"], r1["action2"])
    # rank 0 planned with 5 twice (attempt, re-plan), rank 1 with 9 then 5; the second plan used 6 on both
    assert [e[1] for e in r0["log"]][:3] == [5, 5, 6] and [e[1] for e in r1["log"]][:3] == [9, 5, 6]
    c = cases.build_case("tiny")
    be = PhiloxLikeBackend(c, faulty=False)
    be.counter = 5
    prev = torch.as_tensor(c["prev_mean"][:1]).clone()
    a = sharded_plan(be, torch.as_tensor(c["z0"][:1]), None, prev, torch.tensor([0], dtype=torch.uint8), tape=None, seed=(1 << 40) + 7)
    assert torch.equal(a, r0["action"]) and torch.equal(prev, r0["prev_mean"])
