"""-m gpu: layer-level parity.  Every fused phase of the rollout kernel (each NormedLinear of
reward / dynamics / pi / Q, SimNorm latents, two-hot heads, the policy sample) is dumped through
tdmpc2_plan_estimate_value_trace and compared with its unfused torch counterpart at its own scale."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SLOT_NAMES_T = ["reward.h1", "reward.h2", "dynamics.h1", "dynamics.h2", "z_next"]
SLOT_NAMES_END = ["pi.h1", "pi.h2", "z_H", "Qa.h1", "Qa.h2", "Qb.h1", "Qb.h2"]


@pytest.mark.parametrize("prec", [1, 2], ids=["fp32", "split"])
@pytest.mark.parametrize("name", ["c1", "c2", "mt5"])
def test_every_fused_phase_matches_unfused_torch(name, prec):
    from oracle import planner_oracle as po
    from tests.gpu_common import case_on_gpu, dev, plan_inputs

    c, model, planner = case_on_gpu(name, 1, prec)
    cfg = c["cfg"]
    inp = plan_inputs(c, model)
    E, H, N, A = c["n_envs"], cfg.horizon, cfg.num_samples, cfg.action_dim
    g = torch.Generator().manual_seed(3)
    actions = (torch.rand(E, H, N, A, generator=g) * 2 - 1)
    if cfg.multitask:
        actions = actions * model.sd["_action_masks"][torch.tensor(c["tasks"])].view(E, 1, 1, A)
    eps = torch.randn(E, N, A, generator=g)
    qidx = torch.tensor([[1, 4], [3, 0], [2, 1]][:E], dtype=torch.int32)
    value, tiles, scalars = planner.estimate_value(inp["z0"], inp["disc_pow"], actions.to(dev()).contiguous(),
                                                   eps.to(dev()).contiguous(), qidx.to(dev()).contiguous(),
                                                   task_emb=inp["task_emb"], act_mask=inp["act_mask"], trace=True)
    torch.cuda.synchronize()
    tiles = tiles.cpu().view(E, N // 64, 5 * H + 7, 64, cfg.latent_dim).permute(0, 2, 1, 3, 4).reshape(E, 5 * H + 7, N, -1)
    scalars, value = scalars.cpu(), value.cpu()
    names = [f"t{t}.{s}" for t in range(H) for s in SLOT_NAMES_T] + SLOT_NAMES_END
    report = []
    for e in range(E):
        task = None if c["tasks"] is None else c["tasks"][e]
        wv, wt, ws = po.trace_estimate_value(model, torch.as_tensor(c["z0"][e:e + 1]).repeat(N, 1), actions[e], task,
                                             c["discounts"][e], eps[e], qidx[e])
        for s, nm in enumerate(names):
            d = (tiles[e, s] - wt[s]).abs().max().item()
            report.append((e, nm, d))
        ds = (scalars[e] - ws).abs() / ws.abs().clamp_min(1.0)
        report.append((e, "r_t / Q / a_H (rel)", ds.max().item()))
        dv = ((value[e] - wv).abs() / wv.abs().clamp_min(1.0)).max().item()
        report.append((e, "value (rel)", dv))
    for e, nm, d in report:
        print(f"[{name}] env {e} {nm:24s} max err {d:.3e}")
    # hidden activations are O(1) after LayerNorm+Mish; SimNorm latents are in [0,1]
    assert max(d for _, nm, d in report if "(rel)" not in nm) < 2e-5
    assert max(d for _, nm, d in report if "(rel)" in nm) < 1e-4
