"""-m gpu: the f16x2-split arithmetic on ADVERSARIAL weights: heavy-tailed Student-t(3) matrices, one
100x outlier column per matrix, LayerNorm gains in [0.1, 10] and biases in [-3, 3] -- both kernel families, against an
fp64 evaluation of the same network, with the torch-CPU fp32 arithmetic the reference runs as the yardstick.  And the
saturation guard: a LayerNorm gain large enough to overflow a fixed 2^5 operand scale must not turn into NaN -> 0."""
import numpy as np
import pytest
import torch

from tests.helpers import record_parity

pytestmark = pytest.mark.gpu


def adversarial_state_dict(cfg, seed=0, gain_hi=10.0):
    from tdmpc2_amd import synth

    sd = synth.make_state_dict(cfg, seed=seed)
    rng = np.random.default_rng(seed + 101)
    out = {}
    for k, v in sd.items():
        if k.endswith(".ln.weight"):
            out[k] = np.exp(rng.uniform(np.log(0.1), np.log(gain_hi), v.shape)).astype(np.float32)
        elif k.endswith(".ln.bias"):
            out[k] = rng.uniform(-3.0, 3.0, v.shape).astype(np.float32)
        elif k.endswith(".weight") and v.ndim >= 2 and not k.startswith("_task_emb"):
            w = (rng.standard_t(3, v.shape) * float(v.std())).astype(np.float32)
            col = int(rng.integers(0, v.shape[-1]))
            w[..., col] *= 100.0  # one outlier input column per matrix (per ensemble member)
            out[k] = w
        else:
            out[k] = v
    return out


def _value_errors(cfg, sd, path, prec, E=2, seed=7, fewrow=None):
    from oracle import cases
    from oracle import planner_oracle as po
    from tdmpc2_amd.native import NativePlanner
    from tests.gpu_common import dev, plan_inputs

    c = cases.build_custom(cfg, E)
    c["sd"] = sd
    tsd = {k: torch.as_tensor(v) for k, v in sd.items()}
    model = po.OracleModel(cfg, tsd)
    model64 = po.OracleModel(cfg, tsd, dtype=torch.float64)
    planner = NativePlanner(cfg, c["iterations"], dev(), max_envs=E, path=path, precision=prec)
    planner.bind_state_dict(tsd)
    if fewrow is not None:
        planner.set_fewrow(fewrow)
    inp = plan_inputs(c, model)
    H, N, A = cfg.horizon, cfg.num_samples, cfg.action_dim
    g = torch.Generator().manual_seed(seed)
    actions = torch.rand(E, H, N, A, generator=g) * 2 - 1
    eps = torch.randn(E, N, A, generator=g)
    qidx = torch.tensor([[0, 2], [1, 0]][:E], dtype=torch.int32)
    if cfg.multitask:
        actions = actions * model.sd["_action_masks"][torch.tensor(c["tasks"])].view(E, 1, 1, A)
    got = planner.estimate_value(inp["z0"], inp["disc_pow"], actions.to(dev()).contiguous(), eps.to(dev()).contiguous(),
                                 qidx.to(dev()).contiguous(), task_emb=inp["task_emb"], act_mask=inp["act_mask"]).cpu().double()
    planner.close()
    worst_hip = worst_ref = 0.0
    for e in range(E):
        task = None if c["tasks"] is None else c["tasks"][e]
        z = torch.as_tensor(c["z0"][e:e + 1]).repeat(N, 1)
        v32 = po.estimate_value(model, z, actions[e], task, c["discounts"][e], eps[e], qidx[e]).squeeze(1).double()
        d64 = c["discounts"][e].double() if torch.is_tensor(c["discounts"][e]) else c["discounts"][e]
        v64 = po.estimate_value(model64, z.double(), actions[e].double(), task, d64, eps[e].double(), qidx[e]).squeeze(1)
        scale = v64.abs().clamp_min(1.0)
        worst_hip = max(worst_hip, ((got[e] - v64).abs() / scale).max().item())
        worst_ref = max(worst_ref, ((v32 - v64).abs() / scale).max().item())
    assert torch.isfinite(got).all()
    return worst_hip, worst_ref


@pytest.mark.parametrize("name,path", [("c1", 1), ("mt5", 1), ("small", 2), ("small_mt", 2)])
def test_split_arithmetic_on_heavy_tailed_weights(name, path):
    from tdmpc2_amd.config import named_config

    cfg = named_config("small", task="mt30") if name == "small_mt" else named_config(name)
    if cfg.multitask:
        cfg.action_dims = [cfg.action_dim - (i % 3) for i in range(len(cfg.tasks))]
    sd = adversarial_state_dict(cfg)
    hip, ref = _value_errors(cfg, sd, path, 2)
    print(f"[{name} path {path}] adversarial weights: |HIP split - fp64| {hip:.3e}   |torch fp32 - fp64| {ref:.3e}")
    record_parity(f"{name}/{'fused' if path == 1 else 'layered'}/split/adversarial_vs_fp64", hip_vs_fp64=hip, torch_fp32_vs_fp64=ref)
    if path == 2:  # the layered family's two routes for a call this small: the few-row path (default) and the per-layer tiles
        hip_t, _ = _value_errors(cfg, sd, path, 2, fewrow=0)
        print(f"[{name} path {path}] ... on the per-layer tiles (TDMPC2_TUNE_FEWROW = 0): {hip_t:.3e}")
        record_parity(f"{name}/layered/split/adversarial_vs_fp64/per_layer_tiles", hip_vs_fp64=hip_t, torch_fp32_vs_fp64=ref)
        assert hip_t < 3 * ref + 1e-6 and hip_t < max(1e-4, 1.5 * ref), (hip_t, ref)
    # no further from exact arithmetic than 3x the fp32 arithmetic the reference itself runs, and inside the 1e-4 bar
    # wherever that arithmetic itself is.  Heavy tails cost torch's own fp32 2.9e-4 on the 64-wide model: an ill-conditioned
    # case where every rounding sequence lands somewhere in 1e-4 .. 4e-4 (the fused-LayerNorm tiles 1.1e-4, the few-row path's
    # two-pass LayerNorm 3.6e-4, r6b) -- the bar there is 1.5x the reference arithmetic's own error, not a lucky draw below it
    assert hip < 3 * ref + 1e-6 and hip < max(1e-4, 1.5 * ref), (hip, ref)


@pytest.mark.parametrize("name,path", [("c1", 1), ("small", 2)])
def test_huge_layernorm_gain_does_not_saturate(name, path):
    """gamma = 300 on the hidden layers: |Mish(LayerNorm)| reaches ~6e3, 32x that overflows f16 (65504).  The bind-time
    activation scale (k_ascale) lowers the operand scale of exactly those layers; values stay finite and fp32-accurate."""
    from tdmpc2_amd import synth
    from tdmpc2_amd.config import named_config

    cfg = named_config(name)
    sd = synth.make_state_dict(cfg, seed=0)
    for k in list(sd):
        if k.endswith(".ln.weight") and not k.startswith(("_dynamics.2", "_encoder")):
            sd[k] = (sd[k] * 300.0).astype(np.float32)
    hip, ref = _value_errors(cfg, sd, path, 2)
    print(f"[{name} path {path}] gamma x300: |HIP split - fp64| {hip:.3e}   |torch fp32 - fp64| {ref:.3e}")
    record_parity(f"{name}/{'fused' if path == 1 else 'layered'}/split/gamma300_vs_fp64", hip_vs_fp64=hip, torch_fp32_vs_fp64=ref)
    assert hip < 10 * ref + 1e-5, (hip, ref)
