"""-m gpu: how much room is there under the 1e-4 gate -- on the models the bench line is quoted on, with weights that look like
TRAINED ones (VERDICT r5 next #4a).

Until round 6 the closest case to the gate was the headline model itself (c2: std 8.5e-5, mean 7.9e-5 of 1e-4) and everything
about the f16x2-split arithmetic's margin had been shown on `synth.make_state_dict`'s kind weights (trunc-normal matrices,
LayerNorm gains 1 +- 0.1) or on the small models.  Here:

* goldens minted by the reference's own planner (oracle/make_golden.py) on `synth.trained_like` weights -- LayerNorm gains
  log-uniform in [0.2, 5], biases N(0, 0.3), one weight in a thousand at 20 sigma, head weights scaled for values of a few
  hundred -- for c2 (the benched I = 6), c3 (48M) and c4 (317M);
* end-to-end action statistics over independently seeded plans for c2 (32 seeds) and c3 (16; 8 with the fp64 twin),
  plain and trained-like (the round-5 statistic was c1 only).

What the first run of these cases showed (r6k / r6l / r6m, profiles/README.md): the KERNELS are as close to an fp64 evaluation of the
network as torch's fp32 is, on trained-like weights too (`_estimate_value` with given actions: 1.9-2.6e-5 against 2.7e-5) -- but a
whole plan is a CHAIN: with values of several hundred, score_k = exp(temperature (v_k - v_max)) turns a relative value error of
1e-5 into per-cent changes of the elite weights, the refitted mean / std move by more than 1e-4, the next iteration samples other
actions, and its values differ in the third digit.  The reference's own fp32 plan is that far from an fp64 plan of the same network.
So the trained-like cases are gated the way the exact-fp32 mode's chain always was -- values of iterations >= 1 within 2e-4,
mean / std / `_prev_mean` / action within min(1e-4 + 4 temperature eps_v, 2.5e-4), and the action additionally within max(5e-5,
3 x the reference fp32 plan's own distance from an fp64 plan) -- and the seeded statistics measure the reference arithmetic's own
conditioning next to ours (median and worst |HIP - fp64| of the same order as |torch fp32 - fp64| over the seeds).  Iteration-0
values of the sampled rows -- a statement about the kernels alone -- are held against fp64 the same way (no further than three times
torch's fp32); plain weights keep the flat 1e-4 gate on everything with the factor of two on the action."""
import numpy as np
import pytest
import torch

from tests.helpers import boundary_gap, elite_sets_equal, load_golden, record_parity
from tests.test_gpu_planner import _compare_stages, _run_native

pytestmark = pytest.mark.gpu

ACTION_MARGIN = 5e-5  # half the gate


@pytest.mark.parametrize("name", ["c2_tl", "c3_tl", "c4_tl"])
def test_trained_like_weights_reproduce_the_reference_golden_with_margin(name):
    from tests.gpu_common import case_on_gpu

    c, model, planner = case_on_gpu(name, 0, 2)  # family by size (c2: fused; c3 / c4: layered), default arithmetic
    g = load_golden(name)
    got = _run_native(c, model, planner)
    assert np.isfinite(got["action"]).all() and np.abs(got["action"]).max() <= 1.0
    vmax = float(np.abs(g["value"]).max())
    worst = _compare_stages(name, c, got, g, g["action"], g["prev_mean_out"], tag="/trained_like/golden", conditioned=True)
    print(f"[{name}] reference values reach |v| = {vmax:.1f}; action error {worst['action']:.2e} = {ACTION_MARGIN / max(worst['action'], 1e-12):.1f}x under 5e-5")
    record_parity(f"{name}/trained_like/golden", value_abs_max=vmax)
    ref_dev = None
    if c["cfg"].latent_dim <= 1024:  # the reference arithmetic's own distance from an fp64 plan of the same network (a 317M fp64 plan: minutes)
        from oracle import planner_oracle as po

        m64 = po.OracleModel(c["cfg"], {k: torch.as_tensor(v) for k, v in c["sd"].items()}, dtype=torch.float64)
        ref_dev = 0.0
        for e in range(c["n_envs"]):
            d = c["discounts"][e]
            a64, _, _ = po.plan(m64, z0=torch.as_tensor(c["z0"][e:e + 1]).double(), tape=po.env_tape(c["tape"], e),
                                prev_mean=torch.as_tensor(c["prev_mean"][e]).double(), t0=bool(c["t0"][e]), eval_mode=c["eval_mode"],
                                task=None if c["tasks"] is None else c["tasks"][e], discount=d.double() if torch.is_tensor(d) else d,
                                iterations=c["iterations"])
            ref_dev = max(ref_dev, float(np.abs(g["action"][e].astype(np.float64) - a64.numpy()).max()))
        print(f"[{name}] the reference's own fp32 plan against an fp64 plan: action {ref_dev:.2e}")
        record_parity(f"{name}/trained_like/golden", torch_fp32_vs_fp64=ref_dev)
    # the factor of two under the gate -- or, where the chain is ill-conditioned, no further from the reference than three times
    # the reference arithmetic's own distance from fp64
    assert worst["action"] < max(ACTION_MARGIN, 3.0 * (ref_dev or 0.0)), (worst, ref_dev)
    assert planner.take_fault() == 0


@pytest.mark.parametrize("cfg_name,E,trained,head_std,overrides", [
    ("c2", 32, False, 0.06, dict(iterations=4)), ("c2", 32, True, 0.015, dict(iterations=4)),
    ("c3", 16, False, 0.03, {}), ("c3", 8, True, 0.02, {}),  # (a 48M oracle plan is seconds of CPU, its fp64 twin more)
], ids=["c2", "c2_trained_like", "c3", "c3_trained_like"])
def test_action_statistics_over_seeds_on_the_benched_models(cfg_name, E, trained, head_std, overrides):
    """One batched call of E environments -- each with its own latent, warm-start mean, noise tape (and task) -- against E
    sequential oracle plans.  top-k makes the map discontinuous: plans whose oracle k-th / (k+1)-th values are closer than 1e-4
    at some iteration may legitimately diverge (counted); everything else must agree to 1e-4, with the factor of two on top."""
    from oracle import cases
    from oracle import planner_oracle as po
    from tdmpc2_amd import synth
    from tdmpc2_amd.config import named_config
    from tdmpc2_amd.native import NativePlanner
    from tests.gpu_common import dev

    cfg = named_config(cfg_name, **overrides)
    c = cases.build_custom(cfg, E, head_std=head_std, name="c3" if cfg.multitask else "")
    if trained:
        c["sd"] = synth.trained_like(c["sd"], seed=0)
    model = po.OracleModel(cfg, {k: torch.as_tensor(v) for k, v in c["sd"].items()})
    planner = NativePlanner(cfg, c["iterations"], dev(), max_envs=E)
    planner.bind_state_dict(model.sd)
    got = _run_native(c, model, planner)
    assert planner.take_fault() == 0
    planner.close()
    K, I = cfg.num_elites, c["iterations"]
    clean, boundary = [], 0
    worst_v0 = 0.0
    hip_rows, ref_rows = [], []
    want_a = np.zeros_like(got["action"])
    ref64_dev = np.zeros(E)
    hip64_dev = np.zeros(E)
    model64 = po.OracleModel(cfg, {k: torch.as_tensor(v) for k, v in c["sd"].items()}, dtype=torch.float64) if trained else None
    for e in range(E):  # one oracle plan at a time (a 48M plan is a few seconds of CPU)
        task = None if c["tasks"] is None else c["tasks"][e]
        kw = dict(tape=po.env_tape(c["tape"], e), t0=bool(c["t0"][e]), eval_mode=False, task=task, discount=c["discounts"][e], iterations=I)
        a, pm, st = po.plan(model, z0=torch.as_tensor(c["z0"][e:e + 1]), prev_mean=torch.as_tensor(c["prev_mean"][e]), **kw)
        want_a[e] = a.numpy()
        # iteration 0 is sampled from the same (mean, std) and the same draws: its values are a statement about the kernels alone
        # (rows >= P: the sampled actions are bit-identical; the first P rows roll out the policy prior, computed by either side)
        P = cfg.num_pi_trajs
        v0 = st["value"][0].numpy().astype(np.float64)[P:]
        worst_v0 = max(worst_v0, float(np.max(np.abs(got["value"][e, 0].astype(np.float64)[P:] - v0) / np.maximum(1.0, np.abs(v0)))))
        same = all(elite_sets_equal(got["elite_idx"][e, it], st["elite_idx"][it].numpy()) for it in range(I))
        if trained:  # the reference arithmetic's own conditioning: the same plan in fp64
            d = c["discounts"][e]
            kw64 = dict(kw, discount=d.double() if torch.is_tensor(d) else d)
            a64, _, st64 = po.plan(model64, z0=torch.as_tensor(c["z0"][e:e + 1]).double(), prev_mean=torch.as_tensor(c["prev_mean"][e]).double(), **kw64)
            v64 = st64["value"][0].numpy()[P:]
            hip_rows.append(np.abs(got["value"][e, 0].astype(np.float64)[P:] - v64) / np.maximum(1.0, np.abs(v64)))
            ref_rows.append(np.abs(v0 - v64) / np.maximum(1.0, np.abs(v64)))
            if same and all(elite_sets_equal(st64["elite_idx"][it].numpy(), st["elite_idx"][it].numpy()) for it in range(I)):
                ref64_dev[e] = float((a.double() - a64).abs().max())
                hip64_dev[e] = float(np.abs(got["action"][e].astype(np.float64) - a64.numpy()).max())
            else:
                same = False  # (an elite swap between any two of the three arithmetics: this plan says nothing about margins)
                if all(elite_sets_equal(got["elite_idx"][e, it], st["elite_idx"][it].numpy()) for it in range(I)):
                    boundary += 1
                    continue
        if not same:
            assert min(boundary_gap(st["value"][it].numpy(), K) for it in range(I)) < (1e-3 if trained else 1e-4), e
            boundary += 1
            continue
        clean.append(e)
    d = got["action"][clean].astype(np.float64) - want_a[clean].astype(np.float64)
    mse, worst = float((d ** 2).mean()), float(np.abs(d).max())
    tag = f"{cfg_name}x{E}seeds{'_trained_like' if trained else ''}"
    print(f"[{tag}] action MSE {mse:.3e}, max |diff| {worst:.3e}, iteration-0 values {worst_v0:.2e}, {boundary} plans at an elite boundary")
    record_parity(f"{tag}/oracle", action_abs=worst, action_mse=mse, value_rel=worst_v0, elite_swaps=int(boundary), plans=E)
    assert len(clean) >= E - max(2, E // 4)
    if trained:
        # iteration-0 values of the sampled rows (identical actions on both sides): a statement about the kernels alone.  A trajectory
        # value is a SUM of rewards of several hundred with either sign -- rows where they cancel to |v| of order 1 carry the
        # components' rounding at full size -- and a reward is a softmax over 101 bins dominated by two or three sharp logits: torch's
        # own fp32 is 2e-4 from fp64 on the worst of 8 000 rows of the 48M model.  The kernels' fp32 accumulation (one chain over
        # the K = 1792 contraction per output; torch's GEMM sums in blocks) measured 2.3x torch's error at the 99.9 % quantile and
        # 4.4x on the worst row -- the exact-fp32 MFMA mode MORE (one fmaf chain), so it is the summation order, not the f16x2
        # split (profiles/r6r_c3_tl_paths.txt).  Gate: the same order of magnitude as the reference arithmetic, population-wise.
        hr, rr = np.concatenate(hip_rows), np.concatenate(ref_rows)
        hip_q, ref_q = float(np.quantile(hr, 0.999)), float(np.quantile(rr, 0.999))
        print(f"[{tag}] iteration-0 values of the {hr.size} sampled rows against fp64: HIP 99.9 % {hip_q:.2e} max {hr.max():.2e}; "
              f"torch fp32 99.9 % {ref_q:.2e} max {rr.max():.2e}")
        record_parity(f"{tag}/vs_fp64", value_hip_vs_fp64=float(hr.max()), value_torch_fp32_vs_fp64=float(rr.max()),
                      value_q999_hip_vs_fp64=hip_q, value_q999_torch_fp32_vs_fp64=ref_q)
        assert hip_q <= max(1e-4, 3.0 * ref_q) and hr.max() <= max(1e-4, 6.0 * rr.max())
        print(f"[{tag}] per plan |torch fp32 - fp64| on the action: median {np.median(ref64_dev[clean]):.2e} max {ref64_dev[clean].max():.2e};  "
              f"|HIP - fp64|: median {np.median(hip64_dev[clean]):.2e} max {hip64_dev[clean].max():.2e}")
        record_parity(f"{tag}/vs_fp64", hip_vs_fp64=float(hip64_dev[clean].max()), torch_fp32_vs_fp64=float(ref64_dev[clean].max()))
        # a chain this sensitive amplifies ANY rounding difference by a plan-specific, sign-random factor: the statement is about the
        # population, not about a plan -- typical and worst distance from fp64 of the same order as the reference arithmetic's own
        assert np.median(hip64_dev[clean]) <= 2.0 * np.median(ref64_dev[clean]) + 1e-5
        assert hip64_dev[clean].max() <= 3.0 * ref64_dev[clean].max() + ACTION_MARGIN
    else:
        assert worst_v0 < 1e-4 and worst < ACTION_MARGIN and mse < 1e-9
