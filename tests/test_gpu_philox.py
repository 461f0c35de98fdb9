"""-m gpu: the FAST mode (tape = NULL, in-kernel Philox4x32-10) pinned against the oracle.

The headline bench runs without a noise tape.  `tdmpc2_plan_export_noise` writes the draws such a plan makes -- the six
sites of the reference (tdmpc2/tdmpc2.py:176,204; tdmpc2/common/world_model.py:156,212; tdmpc2/common/math.py:90) -- as a
tape.  Three things are checked with it:
  1. a tape = NULL plan and the plan replayed from its exported tape are BIT-identical in every stage (so the export is what
     the kernels drew: sampling, policy noise, Q-head choice, Gumbel pick, final noise), on every kernel family;
  2. at the BENCHED launch geometry (c2, I = 6, E = 256, 64-row workgroups, separate k_refit, tape = NULL) the plans of
     sampled environments equal the oracle's plan() on the exported draws: action and _prev_mean within 1e-4;
  3. the draws are what the reference draws in distribution: N(0, 1) for the three normal sites (moments, tails, KS),
     Exp(1) for the Gumbel pick, uniform ordered pairs of distinct heads for qidx, independent streams per plan.
"""
import numpy as np
import pytest
import torch

from tests.helpers import ACT_ATOL, record_parity, value_err

pytestmark = pytest.mark.gpu

STAGES = ("value", "elite_idx", "score", "mean", "std", "actions")


def _many_env_inputs(c, model, E, seed=21):
    """E environments of a golden case's MODEL: fresh latents, random warm-start means, alternating t0."""
    from tdmpc2_amd import synth
    from tests.gpu_common import dev, disc_pow

    cfg = c["cfg"]
    d = dev()
    z0 = synth.make_latents(cfg, E, seed=seed)
    prev = np.random.default_rng(seed + 1).uniform(-0.5, 0.5, (E, cfg.horizon, cfg.action_dim)).astype(np.float32)
    t0 = np.array([(e % 3 == 0) for e in range(E)])
    emb = mask = tasks = None
    if cfg.multitask:
        tasks = [(5 * e + 1) % len(cfg.tasks) for e in range(E)]
        embs = []
        for t in tasks:
            v = model.sd["_task_emb.weight"][t]
            n = v.norm(2)
            embs.append(v * (1.0 / (n + 1e-7)) if n > 1.0 else v)
        emb = torch.stack(embs).to(d).contiguous()
        mask = model.sd["_action_masks"][torch.tensor(tasks)].to(d).contiguous()
        from tdmpc2_amd.config import get_discount

        disc_t = torch.tensor([get_discount(cfg, L) for L in cfg.episode_lengths])
        discounts = [disc_t[t] for t in tasks]
    else:
        discounts = [c["discounts"][0]] * E
    return dict(z0_np=z0, prev_np=prev, t0_np=t0, tasks=tasks, discounts=discounts,
                z0=torch.as_tensor(z0).to(d), prev_mean=torch.as_tensor(prev).to(d), t0=torch.as_tensor(t0.astype(np.uint8)).to(d),
                task_emb=emb, act_mask=mask, disc_pow=disc_pow(cfg, discounts).to(d))


def _plan(planner, inp, tape, seed, eval_mode=False):
    prev = inp["prev_mean"].clone()
    a, st = planner.plan(inp["z0"], inp["disc_pow"], prev, inp["t0"], eval_mode=eval_mode, task_emb=inp["task_emb"],
                         act_mask=inp["act_mask"], tape=tape, seed=seed, debug=True)
    torch.cuda.synchronize()
    return a, prev, st


# (case, path, E, tuning) -- every kernel family that draws: fused 64-row and 32-row workgroups, the cluster path (E = 1),
# in-launch and separate refit, the layered family (multitask and episodic), multitask fused
GEOMETRIES = [
    ("c2_i6", 1, 256, dict(rows=64, fold=0)),   # THE benched launch geometry
    ("c1", 1, 3, dict(rows=32, fold=1, cluster=0)),
    ("c1", 1, 1, dict(cluster=1)),
    ("c1", 1, 1, dict(cluster=2)),              # two clusters per tile (ks_rollout_cl2): the E = 1 latency path
    ("c2_i6", 1, 1, dict(cluster=2)),
    ("c2_ep", 1, 1, dict(cluster=1)),
    ("mt5", 1, 5, dict()),
    ("small_mt", 2, 3, dict()),
    ("small_ep_fire", 2, 2, dict()),
    ("c1", 2, 2, dict()),
]


@pytest.mark.parametrize("name,path,E,tune", GEOMETRIES, ids=[f"{g[0]}-p{g[1]}-E{g[2]}" + "".join(f"-{k}{v}" for k, v in g[3].items()) for g in GEOMETRIES])
def test_exported_tape_reproduces_the_philox_plan_bit_for_bit(name, path, E, tune):
    from oracle import cases
    from oracle import planner_oracle as po
    from tdmpc2_amd.native import NativePlanner
    from tests.gpu_common import dev

    c = cases.build_case(name)
    model = po.OracleModel(c["cfg"], {k: torch.as_tensor(v) for k, v in c["sd"].items()})
    planner = NativePlanner(c["cfg"], c["iterations"], dev(), max_envs=E, path=path)
    planner.bind_state_dict(model.sd)
    if "rows" in tune:
        planner.set_rows_per_workgroup(tune["rows"])
    if "fold" in tune:
        planner.set_fold_refit(tune["fold"])
    if "cluster" in tune:
        planner.set_cluster(tune["cluster"])
    inp = _many_env_inputs(c, model, E)
    seed = 0x1234_5678_9ABC_0000 + E
    planner.plan(inp["z0"], inp["disc_pow"], inp["prev_mean"].clone(), inp["t0"], task_emb=inp["task_emb"],
                 act_mask=inp["act_mask"], seed=1)  # an earlier plan: the counter is not 0 any more
    call = planner.call_counter()
    a0, p0, st0 = _plan(planner, inp, None, seed)
    assert planner.call_counter() == call + 1
    tape = planner.export_noise(seed, call, E)
    a1, p1, st1 = _plan(planner, inp, tape, seed=999)  # (the seed is ignored with a tape)
    assert torch.isfinite(a0).all()
    assert torch.equal(a0, a1) and torch.equal(p0, p1), (name, (a0 - a1).abs().max().item())
    for k in STAGES:
        assert torch.equal(st0[k], st1[k]), (name, k)
    # ... and a partial export (a range of environments) is the same slice
    if E >= 3:
        part = planner.export_noise(seed, call, 2, env_first=E - 2)
        for k, v in part.items():
            assert torch.equal(v, tape[k][E - 2:]), k
    # another call counter / another seed is another stream
    other = planner.export_noise(seed, call + 1, 1, fields=("final_eps", "qidx"))
    assert not torch.equal(other["final_eps"], tape["final_eps"][:1])
    planner.close()


def test_benched_geometry_matches_the_oracle_on_its_own_draws():
    """c2 at I = 6, E = 256 plans in one call, tape = NULL, 64-row workgroups on 2 048 workgroups (8 rounds), k_refit as a
    launch of its own -- what `bench.py` times.  The draws of 8 sampled environments replay through the oracle's plan():
    per-iteration values, elite sets, mean / std, the action and the new _prev_mean."""
    from oracle import cases
    from oracle import planner_oracle as po
    from tdmpc2_amd.native import NativePlanner
    from tests.gpu_common import dev

    E = 256
    c = cases.build_case("c2_i6")
    cfg, I = c["cfg"], c["iterations"]
    assert I == 6
    model = po.OracleModel(cfg, {k: torch.as_tensor(v) for k, v in c["sd"].items()})
    planner = NativePlanner(cfg, I, dev(), max_envs=E, path=1)
    planner.bind_state_dict(model.sd)
    planner.set_rows_per_workgroup(64)
    planner.set_fold_refit(0)
    inp = _many_env_inputs(c, model, E, seed=33)
    seed = 20260924
    call = planner.call_counter()
    a, prev, st = _plan(planner, inp, None, seed)
    worst = dict(value=0.0, mean=0.0, action=0.0, prev_mean=0.0)
    swaps = 0
    for e in (0, 1, 37, 100, 129, 200, 254, 255):
        tp = planner.export_noise(seed, call, 1, env_first=e)
        tp = {k: v[0].cpu() for k, v in tp.items()}
        wa, wpm, wst = po.plan(model, z0=torch.as_tensor(inp["z0_np"][e:e + 1]), tape=tp, prev_mean=torch.as_tensor(inp["prev_np"][e]),
                               t0=bool(inp["t0_np"][e]), eval_mode=False, task=None, discount=inp["discounts"][e], iterations=I)
        # iteration 0 is sampled from the same (mean, std) and the same draws: identical actions, values to 1e-4
        P = cfg.num_pi_trajs
        np.testing.assert_array_equal(st["actions"][e, 0, :, P:].cpu().numpy(), wst["actions"][0][:, P:].numpy())
        np.testing.assert_allclose(st["actions"][e, 0, :, :P].cpu().numpy(), wst["actions"][0][:, :P].numpy(), atol=2e-5, rtol=0)
        worst["value"] = max(worst["value"], value_err(st["value"][e, 0].cpu().numpy(), wst["value"][0].numpy()))
        same = all(set(st["elite_idx"][e, it].cpu().tolist()) == set(wst["elite_idx"][it].tolist()) for it in range(I))
        if not same:  # top-k is discontinuous: count, and require that the reference's own boundary is that close
            swaps += 1
            continue
        worst["mean"] = max(worst["mean"], (st["mean"][e].cpu() - wst["mean"]).abs().max().item())
        worst["action"] = max(worst["action"], (a[e].cpu() - wa).abs().max().item())
        worst["prev_mean"] = max(worst["prev_mean"], (prev[e].cpu() - wpm).abs().max().item())
    print(f"[c2_i6 E=256 philox] worst {worst}, elite-boundary swaps {swaps}")
    record_parity("c2_i6/fused/split/philox_E256_benched_geometry", value_rel=worst["value"], mean_abs=worst["mean"],
                  action_abs=worst["action"], prev_mean_abs=worst["prev_mean"], elite_swaps=int(swaps), plans=8)
    assert worst["value"] < 1e-4
    assert swaps <= 1
    assert worst["action"] < ACT_ATOL and worst["prev_mean"] < ACT_ATOL and worst["mean"] < ACT_ATOL
    planner.close()


@pytest.mark.parametrize("name,E,envs", [("c3", 30, (0, 13, 29)), ("c4", 8, (0, 7))], ids=["c3-E30", "c4-E8"])
def test_benched_layered_geometry_matches_the_oracle_on_its_own_draws(name, E, envs):
    """The layered legs of the bench line at THEIR geometry: c3 (mt30 48M) with E = 30 plans per call and c4 (mt80 317M, H5 N1024)
    with E = 8, I = 6, tape = NULL -- hidden layers on g_gemm_w (256 x 256 tiles, XCD-local / row-major order), two chains in
    flight, in-kernel Philox.  The draws of two or three of the plans replay through the oracle's plan() (a 317M plan is about
    a minute of CPU): values of every iteration, elite sets, mean / std, action, new _prev_mean."""
    from oracle import cases
    from oracle import planner_oracle as po
    from tdmpc2_amd.native import NativePlanner
    from tests.gpu_common import dev

    c = cases.build_case("c3_x4" if name == "c3" else "c4_x2")  # (the 6-iteration cases of those models)
    cfg, I = c["cfg"], c["iterations"]
    assert I == 6
    model = po.OracleModel(cfg, {k: torch.as_tensor(v) for k, v in c["sd"].items()})
    planner = NativePlanner(cfg, I, dev(), max_envs=E, path=2)
    planner.bind_state_dict(model.sd)
    inp = _many_env_inputs(c, model, E, seed=41)
    seed = 20260925
    call = planner.call_counter()
    a, prev, st = _plan(planner, inp, None, seed)
    assert planner.take_fault() == 0
    worst = dict(value=0.0, mean=0.0, action=0.0, prev_mean=0.0)
    swaps = 0
    for e in envs:
        tp = planner.export_noise(seed, call, 1, env_first=e)
        tp = {k: v[0].cpu() for k, v in tp.items()}
        wa, wpm, wst = po.plan(model, z0=torch.as_tensor(inp["z0_np"][e:e + 1]), tape=tp, prev_mean=torch.as_tensor(inp["prev_np"][e]),
                               t0=bool(inp["t0_np"][e]), eval_mode=False, task=inp["tasks"][e], discount=inp["discounts"][e], iterations=I)
        P = cfg.num_pi_trajs
        np.testing.assert_array_equal(st["actions"][e, 0, :, P:].cpu().numpy(), wst["actions"][0][:, P:].numpy())
        worst["value"] = max(worst["value"], value_err(st["value"][e, 0].cpu().numpy(), wst["value"][0].numpy()))
        same = all(set(st["elite_idx"][e, it].cpu().tolist()) == set(wst["elite_idx"][it].tolist()) for it in range(I))
        if not same:  # top-k is discontinuous: count, and require that the reference's own boundary is that close
            swaps += 1
            continue
        for it in range(I):
            worst["value"] = max(worst["value"], value_err(st["value"][e, it].cpu().numpy(), wst["value"][it].numpy()))
        worst["mean"] = max(worst["mean"], (st["mean"][e].cpu() - wst["mean"]).abs().max().item())
        worst["action"] = max(worst["action"], (a[e].cpu() - wa).abs().max().item())
        worst["prev_mean"] = max(worst["prev_mean"], (prev[e].cpu() - wpm).abs().max().item())
    print(f"[{name} E={E} philox, layered] worst {worst}, elite-boundary swaps {swaps}")
    record_parity(f"{name}/layered/split/philox_E{E}_benched_geometry", value_rel=worst["value"], mean_abs=worst["mean"],
                  action_abs=worst["action"], prev_mean_abs=worst["prev_mean"], elite_swaps=int(swaps), plans=len(envs))
    assert worst["value"] < 1e-4
    assert swaps <= 1
    assert worst["action"] < ACT_ATOL and worst["prev_mean"] < ACT_ATOL and worst["mean"] < ACT_ATOL
    planner.close()


def _normal_checks(x, what):
    """x: 1-D float64 sample that should be N(0, 1)."""
    from scipy import stats

    n = x.size
    assert np.isfinite(x).all(), what
    se = 1.0 / np.sqrt(n)
    assert abs(x.mean()) < 5 * se, (what, "mean", x.mean())
    assert abs(x.var() - 1.0) < 5 * np.sqrt(2.0) * se, (what, "var", x.var())
    assert abs(stats.skew(x)) < 5 * np.sqrt(6.0) * se, (what, "skew")
    assert abs(stats.kurtosis(x)) < 5 * np.sqrt(24.0) * se, (what, "kurtosis")
    for k, p in ((1.0, 0.31731050786), (2.0, 0.04550026390), (3.0, 2.6997960633e-3), (4.0, 6.334248367e-5)):
        got = float((np.abs(x) > k).mean())
        assert abs(got - p) < 5 * np.sqrt(p * (1 - p) / n) + 2.0 / n, (what, f"P(|x| > {k})", got, p)
    sub = x[:: max(1, n // 200_000)]
    ks = stats.kstest(sub, "norm")
    assert ks.pvalue > 1e-4, (what, "KS", ks)
    assert np.abs(x).max() < 7.0, (what, "largest draw", np.abs(x).max())  # 24-bit uniforms: |x| <= sqrt(2 ln 2^25) = 5.9


def test_draws_have_the_reference_distributions():
    """The six sites, in bulk, from the generator the kernels use (c2 dims: A = 38 -> the 48-column pair layout)."""
    from scipy import stats
    from tdmpc2_amd.config import named_config
    from tdmpc2_amd.native import NativePlanner
    from tests.gpu_common import dev

    cfg = named_config("c2")
    I, E = 8, 16
    planner = NativePlanner(cfg, I, dev(), max_envs=1)  # (no weights needed: the generator is a function of the key only)
    t = planner.export_noise(seed=77, call=3, n_envs=E)
    se = t["sample_eps"].double().cpu().numpy()           # [E, I, H, N - P, A]: 8.5 M normals (Box-Muller pairs, fast log / sin / cos)
    _normal_checks(se.reshape(-1), "sample_eps")
    _normal_checks(se[..., 0::2].reshape(-1), "sample_eps cos branch")
    _normal_checks(se[..., 1::2].reshape(-1), "sample_eps sin branch")
    # the two branches of a pair are independent normals, and so are neighbouring pairs / iterations / plans
    a, b = se[..., 0::2].reshape(-1), se[..., 1::2].reshape(-1)
    assert abs(np.corrcoef(a, b)[0, 1]) < 5 / np.sqrt(a.size)
    assert abs(np.corrcoef(a ** 2, b ** 2)[0, 1]) < 5 / np.sqrt(a.size)
    assert abs(np.corrcoef(se[0].reshape(-1), se[1].reshape(-1))[0, 1]) < 5 / np.sqrt(se[0].size)           # plan 0 vs plan 1
    assert abs(np.corrcoef(se[:, 0].reshape(-1), se[:, 1].reshape(-1))[0, 1]) < 5 / np.sqrt(se[:, 0].size)  # iteration 0 vs 1
    _normal_checks(t["pi_eps"].double().cpu().numpy().reshape(-1), "pi_eps")           # [E, I, N, A]: 2.5 M
    # small sites: pool several keys
    pt, fe, ge, qi = [], [], [], []
    for s in range(40):
        u = planner.export_noise(seed=1000 + s, call=s, n_envs=64, fields=("pi_traj_eps", "final_eps", "gumbel_exp", "qidx"))
        pt.append(u["pi_traj_eps"].double().cpu().numpy().reshape(-1))
        fe.append(u["final_eps"].double().cpu().numpy().reshape(-1))
        ge.append(u["gumbel_exp"].double().cpu().numpy().reshape(-1))
        qi.append(u["qidx"].cpu().numpy().reshape(-1, 2))
    _normal_checks(np.concatenate(pt), "pi_traj_eps")
    _normal_checks(np.concatenate(fe), "final_eps")
    g = np.concatenate(ge)                                 # Exp(1): what torch's exponential_() draws (math.py:90)
    assert (g > 0).all() and np.isfinite(g).all()
    n = g.size
    assert abs(g.mean() - 1.0) < 5 / np.sqrt(n) and abs(g.var() - 1.0) < 5 * np.sqrt(8.0 / n)
    for k in (0.1, 1.0, 3.0, 6.0):
        p = np.exp(-k)
        assert abs(float((g > k).mean()) - p) < 5 * np.sqrt(p * (1 - p) / n) + 2.0 / n, k
    assert stats.kstest(g[:200_000], "expon").pvalue > 1e-4
    q = np.concatenate(qi)                                 # randperm(nq)[:2]: uniform over the nq (nq - 1) ordered pairs
    nq = cfg.num_q
    assert (q[:, 0] != q[:, 1]).all() and q.min() == 0 and q.max() == nq - 1
    counts = np.zeros((nq, nq))
    np.add.at(counts, (q[:, 0], q[:, 1]), 1)
    obs = counts[~np.eye(nq, dtype=bool)]
    chi = stats.chisquare(obs)
    assert chi.pvalue > 1e-4, (chi, counts)
    planner.close()


def test_streams_of_different_plans_do_not_collide():
    """E = 256 plans of one call (and consecutive calls, and neighbouring seeds) draw distinct, uncorrelated streams."""
    from tdmpc2_amd.config import named_config
    from tdmpc2_amd.native import NativePlanner
    from tests.gpu_common import dev

    cfg = named_config("c1")
    planner = NativePlanner(cfg, 6, dev(), max_envs=1)
    heads = []
    for seed, call in ((5, 0), (5, 1), (6, 0), ((1 << 32) | 5, 0)):
        t = planner.export_noise(seed=seed, call=call, n_envs=256, fields=("final_eps", "gumbel_exp", "pi_traj_eps"))
        heads.append(torch.cat([t["final_eps"], t["gumbel_exp"], t["pi_traj_eps"].reshape(256, -1)[:, :64]], dim=1).cpu().numpy())
    allrows = np.concatenate(heads)                        # 1 024 streams x 134 leading draws
    assert len({r.tobytes() for r in allrows}) == allrows.shape[0], "two plans share a noise stream"
    # no value of one stream's head reappears at the same position in another (shifted-counter collisions)
    first = allrows[:, 0]
    assert len(np.unique(first)) > 0.99 * first.size
    c = np.corrcoef(allrows[:, 70:134])                    # columns: final_eps [0, 6) | gumbel [6, 70) | pi_traj normals [70, 134)
    off = c[~np.eye(c.shape[0], dtype=bool)]
    assert np.abs(off).max() < 0.75 and abs(off.mean()) < 0.01  # 64 samples per row: |r| ~ 0.125 typical
    planner.close()
