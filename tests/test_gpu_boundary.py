"""-m gpu: the drop-in TDMPC2 class (act/plan/load/save) against the oracle, including encode()."""
import numpy as np
import pytest
import torch

from tests.helpers import ACT_ATOL, boundary_gap, record_parity

pytestmark = pytest.mark.gpu


def _agent(name, max_envs=1):
    from oracle import cases
    from tdmpc2_amd.tdmpc2 import TDMPC2

    c = cases.build_case(name)
    agent = TDMPC2(c["cfg"].replace(), device=torch.device("cuda", 0), max_envs=max_envs)
    agent.load({"model": {k: torch.as_tensor(v) for k, v in c["sd"].items()}})
    return c, agent


@pytest.mark.parametrize("name", ["c1", "mt5"])
def test_act_matches_oracle_with_encode(name):
    """agent.act(obs, t0, eval_mode, task) — the evaluate.py:80 call — vs the oracle fed the same tape."""
    from oracle import planner_oracle as po
    from tdmpc2_amd import synth

    c, agent = _agent(name)
    cfg = c["cfg"]
    assert agent.cfg.iterations == c["iterations"]
    model = po.OracleModel(cfg, {k: torch.as_tensor(v) for k, v in c["sd"].items()})
    obs = synth.make_obs(cfg, c["n_envs"], seed=3)
    e = 0
    task = None if c["tasks"] is None else c["tasks"][e]
    tape_e = po.env_tape(c["tape"], e)
    agent.noise_tape = {k: v.unsqueeze(0).to(agent.device).contiguous() for k, v in tape_e.items()}
    # two consecutive steps: t0 then warm start from the planner's own _prev_mean
    prev = torch.zeros(cfg.horizon, cfg.action_dim)
    for step, t0 in enumerate([True, False]):
        a = agent.act(torch.as_tensor(obs[e]), t0=t0, eval_mode=False, task=task)
        assert a.device.type == "cpu" and a.shape == (cfg.action_dim,)
        wa, wpm, st = po.plan(model, obs=torch.as_tensor(obs[e:e + 1]), tape=tape_e, prev_mean=prev, t0=t0,
                              eval_mode=False, task=task, discount=c["discounts"][e], iterations=c["iterations"])
        da = (a - wa).abs().max().item()
        dm = (agent._prev_mean.cpu() - wpm).abs().max().item()
        print(f"[{name}] step {step}: action diff {da:.2e}, prev_mean diff {dm:.2e}")
        record_parity(f"{name}/act()/step{step}", action_abs=da, prev_mean_abs=dm)
        if da >= ACT_ATOL:
            # legitimate only through an elite-boundary swap: the oracle's own k-th / (k+1)-th values must be closer than
            # 1e-4 at some iteration (top-k is discontinuous there); anything else is a wiring regression and fails
            gaps = [boundary_gap(st["value"][it].numpy(), cfg.num_elites) for it in range(c["iterations"])]
            assert min(gaps) < 1e-4, f"act() differs from the oracle by {da:.2e} with no elite boundary within 1e-4 (gaps {gaps})"
            return
        assert dm < ACT_ATOL
        prev = wpm


def test_save_load_round_trip(tmp_path):
    from tdmpc2_amd import checkpoint
    from tdmpc2_amd.tdmpc2 import TDMPC2

    c, agent = _agent("c1")
    fp = tmp_path / "agent.pt"
    agent.save(fp)
    blob = torch.load(fp, weights_only=False)
    assert set(blob) == {"model"}
    assert "_Qs.params.__batch_size" in blob["model"] and "_target_Qs_params.2.bias" in blob["model"]
    other = TDMPC2(c["cfg"].replace(), device=torch.device("cuda", 0))
    other.load(str(fp))
    for (k1, v1), (k2, v2) in zip(agent.model.state_dict().items(), other.model.state_dict().items()):
        if torch.is_tensor(v1):
            assert k1 == k2 and torch.equal(v1, v2)
    # old-format (released-checkpoint style) keys load too
    old = checkpoint.to_old_format(blob["model"])
    third = TDMPC2(c["cfg"].replace(), device=torch.device("cuda", 0))
    third.load({"model": old})
    assert torch.equal(third.model._Qs.params.layer(1).weight, agent.model._Qs.params.layer(1).weight)


def test_plan_batch_vectorised_envs():
    from tdmpc2_amd import synth

    c, agent = _agent("c1", max_envs=8)
    obs = torch.as_tensor(synth.make_obs(c["cfg"], 8, seed=9))
    a = agent.act_batch(obs, t0=True)
    assert a.shape == (8, c["cfg"].action_dim) and torch.isfinite(a).all() and a.abs().max() <= 1
    b = agent.act_batch(obs, t0=False)
    assert not torch.equal(a, b)


@pytest.mark.parametrize("name", ["c1", "c1_wide"])
def test_plan_is_hip_graph_capturable(name):
    """`run` allocates nothing and launches only on the caller's stream: a whole plan (setup, policy prior, I x (rollout, refit))
    captures into a hipGraph (what the reference obtains with torch.compile's reduce-overhead mode, tdmpc2.py:52) and the
    replay reproduces the eager result bit for bit on the same noise tape.  c1: two plans (one cluster per tile); c1_wide: ONE
    plan -- the two-clusters-per-tile kernel (ks_rollout_cl2), whose arrival words ks_setup zeroes inside the graph."""
    from tests.gpu_common import case_on_gpu, plan_inputs

    c, model, planner = case_on_gpu(name)
    inp = plan_inputs(c, model)
    kw = dict(task_emb=inp["task_emb"], act_mask=inp["act_mask"], tape=inp["tape"])
    pm_eager = inp["prev_mean"].clone()
    a_eager = planner.plan(inp["z0"], inp["disc_pow"], pm_eager, inp["t0"], **kw).clone()
    pm_static, out = inp["prev_mean"].clone(), torch.empty_like(a_eager)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):  # warm-up on the side stream, then capture
        planner.plan(inp["z0"], inp["disc_pow"], pm_static.clone(), inp["t0"], out=out, **kw)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        planner.plan(inp["z0"], inp["disc_pow"], pm_static, inp["t0"], out=out, **kw)
    from tests.helpers import load_golden

    want = torch.as_tensor(load_golden(name)["action"]).to(a_eager.device)
    assert (a_eager - want).abs().max() < ACT_ATOL, (a_eager.tolist(), planner.take_fault(), planner.fault_info())
    for i in range(2):
        pm_static.copy_(inp["prev_mean"])
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, a_eager) and torch.equal(pm_static, pm_eager), \
            (i, float((out - want).abs().max()), float((a_eager - want).abs().max()), planner.take_fault())


def test_a_handle_moved_to_another_stream_orders_its_own_calls():
    """One handle = one workspace: a plan on the NULL stream followed AT ONCE by a plan of the same handle on a non-blocking
    stream (no wait_stream in between -- what test_plan_is_hip_graph_capturable did, and what made it fail whenever a second
    hardware queue already existed: profiles/README.md 'r03k, root cause') must not overlap.  The handle leaves an event
    behind every call and makes the first call on a different stream wait for it (StreamTurn, tdmpc2_plan.hip)."""
    from tests.gpu_common import case_on_gpu, plan_inputs

    c, model, planner = case_on_gpu("c1")
    inp = plan_inputs(c, model)
    kw = dict(task_emb=inp["task_emb"], act_mask=inp["act_mask"], tape=inp["tape"])
    z_b = inp["z0"].roll(1, 0).contiguous()  # the second plan: other latents -> another answer
    pm = inp["prev_mean"]
    want_a = planner.plan(inp["z0"], inp["disc_pow"], pm.clone(), inp["t0"], **kw).clone()
    want_b = planner.plan(z_b, inp["disc_pow"], pm.clone(), inp["t0"], **kw).clone()
    torch.cuda.synchronize()
    assert not torch.equal(want_a, want_b)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):  # the stream's hardware queue exists before the race starts
        torch.zeros(8, device=want_a.device).add_(1)
    torch.cuda.synchronize()
    for _ in range(20):
        pm_a, pm_b = pm.clone(), pm.clone()
        torch.cuda.synchronize()
        a = planner.plan(inp["z0"], inp["disc_pow"], pm_a, inp["t0"], **kw).clone()  # NULL stream, still running when ...
        with torch.cuda.stream(s):                                                   # ... the same handle plans on `s`
            b = planner.plan(z_b, inp["disc_pow"], pm_b, inp["t0"], **kw).clone()
        a2 = planner.plan(inp["z0"], inp["disc_pow"], pm.clone(), inp["t0"], **kw).clone()  # and back
        torch.cuda.synchronize()
        assert torch.equal(a, want_a) and torch.equal(b, want_b) and torch.equal(a2, want_a)
    assert planner.take_fault() == 0


def test_rebinding_weights_and_concurrent_handles():
    """`load()` / a training step re-binds weights into a live handle (tdmpc2.py:81-95 semantics); two handles on two
    streams share no mutable state: both reproduce their own single-handle results when run concurrently."""
    from oracle import cases
    from tests.gpu_common import dev, plan_inputs
    from oracle import planner_oracle as po
    from tdmpc2_amd.native import NativePlanner

    c1 = cases.build_case("c1")
    sd_a = {k: torch.as_tensor(v) for k, v in c1["sd"].items()}
    from tdmpc2_amd import synth
    sd_b = {k: torch.as_tensor(v) for k, v in synth.make_state_dict(c1["cfg"], seed=123).items()}
    model = po.OracleModel(c1["cfg"], sd_a)
    inp = plan_inputs(c1, model)
    kw = dict(task_emb=None, act_mask=None, tape=inp["tape"])

    def run(pl, stream=None):
        if stream is None:
            return pl.plan(inp["z0"], inp["disc_pow"], inp["prev_mean"].clone(), inp["t0"], **kw).clone()
        with torch.cuda.stream(stream):  # every op of this plan, the prev_mean copy included, on that stream
            return pl.plan(inp["z0"], inp["disc_pow"], inp["prev_mean"].clone(), inp["t0"], **kw)

    pa = NativePlanner(c1["cfg"], c1["iterations"], dev(), max_envs=c1["n_envs"])
    pb = NativePlanner(c1["cfg"], c1["iterations"], dev(), max_envs=c1["n_envs"])
    pa.bind_state_dict(sd_a)
    pb.bind_state_dict(sd_b)
    ra, rb = run(pa), run(pb)
    assert (ra - rb).abs().max() > 1e-3  # different weights, different plans
    # re-bind: handle b takes a's weights and must now reproduce a's plan exactly
    pb.bind_state_dict(sd_a)
    assert torch.equal(run(pb), ra)
    pb.bind_state_dict(sd_b)
    # concurrently on two streams
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(3):
        outs.append((run(pa, s1), run(pb, s2)))
    torch.cuda.synchronize()
    for oa, ob in outs:
        assert torch.equal(oa, ra) and torch.equal(ob, rb)


@pytest.mark.parametrize("name", ["c1", "mt5"])
def test_agent_td_target_matches_reference_golden(name):
    """TDMPC2._td_target with the reference's signature (next_z [H, B, L], reward / terminated [H, B, 1], task [B]) against the
    reference-minted fixture."""
    from oracle import cases
    from tests.helpers import load_golden

    c, agent = _agent(name)
    tb = cases.td_batch(c["cfg"])
    task = None if tb["tasks"] is None else torch.as_tensor(tb["tasks"])
    td = agent._td_target(torch.as_tensor(tb["next_z"]), torch.as_tensor(tb["reward"]), torch.as_tensor(tb["terminated"]), task,
                          pi_eps=torch.as_tensor(tb["pi_eps"]), qidx=torch.as_tensor(tb["qidx"]).to(agent.device))
    want = torch.as_tensor(load_golden(name)["td_target"])
    assert td.shape == want.shape
    err = ((td.cpu() - want).abs() / want.abs().clamp(min=1)).max().item()
    record_parity(f"{name}/agent._td_target", value_rel=err)
    assert err < 1e-4
    a, q = agent.policy_value(torch.as_tensor(tb["next_z"]), task)
    assert a.shape == (*tb["next_z"].shape[:2], c["cfg"].action_dim) and q.shape == (*tb["next_z"].shape[:2], 1)
    assert torch.isfinite(a).all() and torch.isfinite(q).all()


def test_pixel_observation_agent_plans():
    """obs = 'rgb': the conv encoder (host-side module with the reference's keys) feeds the planner its latent; act() works
    end to end and equals planning from the same latent through the library directly (same seed, fresh handle)."""
    from tdmpc2_amd.config import named_config
    from tdmpc2_amd.native import NativePlanner
    from tdmpc2_amd.tdmpc2 import TDMPC2

    cfg = named_config("c1")
    cfg.obs = "rgb"
    cfg.obs_shape = {"rgb": (9, 64, 64)}
    torch.manual_seed(0)
    agent = TDMPC2(cfg, device=torch.device("cuda", 0))
    for p in agent.model._reward[2].parameters():  # fresh models zero these heads (all values equal): make them informative
        torch.nn.init.normal_(p, std=0.05)
    agent.sync_planner_weights()
    obs = torch.randint(0, 256, (9, 64, 64)).float()
    torch.manual_seed(3)  # ShiftAug's shift
    a = agent.act(obs, t0=True)
    assert a.shape == (cfg.action_dim,) and torch.isfinite(a).all() and a.abs().max() <= 1
    torch.manual_seed(3)
    z = agent.model.encode(obs.to(agent.device).unsqueeze(0), None).float().contiguous()
    ref = NativePlanner(cfg, agent.cfg.iterations, agent.device, max_envs=1)
    ref.bind_state_dict(agent.model.planner_state_dict())
    disc = agent._disc_pow(None).unsqueeze(0).contiguous()
    b = ref.plan(z, disc, torch.zeros(1, cfg.horizon, cfg.action_dim, device=agent.device),
                 torch.ones(1, dtype=torch.uint8, device=agent.device), seed=agent._seed)
    assert torch.equal(a, b[0].cpu())


def test_act_replans_a_step_whose_cluster_plan_gave_up(monkeypatch):
    """a cluster hand-over that times out must not leak a garbage action into env.step.  The library returns NaN
    and keeps `_prev_mean`; `act()` sees the fault after its `.cpu()` sync and plans the step again on the path without
    hand-overs.  With a noise tape the re-planned action is exactly what a cluster-less agent returns."""
    from oracle import planner_oracle as po
    from tdmpc2_amd import synth

    monkeypatch.setenv("TDMPC2_CLUSTER_FAULT", "1")
    c, faulty = _agent("c1")
    faulty.planner()  # the handle reads the test hook at create
    monkeypatch.delenv("TDMPC2_CLUSTER_FAULT")
    monkeypatch.setenv("TDMPC2_CLUSTER", "0")
    _, plain = _agent("c1")
    plain.planner()
    monkeypatch.delenv("TDMPC2_CLUSTER")
    obs = torch.as_tensor(synth.make_obs(c["cfg"], 1, seed=3)[0])
    tape = {k: v.unsqueeze(0).to(faulty.device).contiguous() for k, v in po.env_tape(c["tape"], 0).items()}
    faulty.noise_tape = plain.noise_tape = tape
    for t0 in (True, False):
        x = faulty.act(obs, t0=t0, eval_mode=False)
        y = plain.act(obs, t0=t0, eval_mode=False)
        assert torch.isfinite(x).all() and torch.equal(x, y)
        assert torch.equal(faulty._prev_mean, plain._prev_mean)
    assert faulty.planner().take_fault() == 0  # act() consumed the report
