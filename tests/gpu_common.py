"""Helpers shared by the -m gpu parity tests (HIP planner vs oracle on the same seeded inputs)."""
import numpy as np
import torch

from oracle import cases
from oracle import planner_oracle as po

_cache = {}


def dev():
    return torch.device("cuda", 0)


def case_on_gpu(name, path=0, precision=0):
    """(case dict, oracle model, NativePlanner with the case's weights bound).
    path: 0 auto, 1 fused, 2 layered; precision: 0 auto, 1 exact-fp32 MFMA, 2 f16x2 split."""
    key = (name, path, precision)
    if key in _cache:
        return _cache[key]
    from tdmpc2_amd.native import NativePlanner

    for k, (c, model, _) in list(_cache.items()):
        if k[0] == name:
            break
    else:
        c = cases.build_case(name)
        model = po.OracleModel(c["cfg"], {k: torch.as_tensor(v) for k, v in c["sd"].items()})
    if c["cfg"].latent_dim > 1024:  # 317M-class weights: keep one such case resident at a time
        for k in [k for k in _cache if _cache[k][0]["cfg"].latent_dim > 1024 and k[0] != name]:
            del _cache[k]
    planner = NativePlanner(c["cfg"], c["iterations"], dev(), max_envs=max(2, c["n_envs"]), path=path, precision=precision)
    planner.bind_state_dict(model.sd)
    _cache[key] = (c, model, planner)
    return _cache[key]


def plan_inputs(c, model):
    """Device tensors for NativePlanner.plan from a case dict."""
    cfg, E = c["cfg"], c["n_envs"]
    d = dev()
    z0 = torch.as_tensor(c["z0"]).to(d)
    prev = torch.as_tensor(c["prev_mean"]).to(d).clone()
    t0 = torch.as_tensor(c["t0"].astype(np.uint8)).to(d)
    tape = {k: torch.as_tensor(v).to(d).contiguous() for k, v in c["tape"].items()}
    emb = mask = None
    if cfg.multitask:
        embs = []
        for t in c["tasks"]:
            e = model.sd["_task_emb.weight"][t]
            n = e.norm(2)
            embs.append(e * (1.0 / (n + 1e-7)) if n > 1.0 else e)  # nn.Embedding(max_norm=1)
        emb = torch.stack(embs).to(d).contiguous()
        mask = model.sd["_action_masks"][torch.tensor(c["tasks"])].to(d).contiguous()
    disc = disc_pow(cfg, c["discounts"]).to(d)
    return dict(z0=z0, prev_mean=prev, t0=t0, tape=tape, task_emb=emb, act_mask=mask, disc_pow=disc)


def disc_pow(cfg, discounts):
    """discount^0..^H the way tdmpc2/tdmpc2.py:126,130-132 accumulates it."""
    rows = []
    for g in discounts:
        if torch.is_tensor(g):
            d = torch.ones((), dtype=torch.float32)
            vals = [d]
            for _ in range(cfg.horizon):
                d = d * g
                vals.append(d)
            rows.append(torch.stack(vals))
        else:
            d, vals = 1, []
            for _ in range(cfg.horizon + 1):
                vals.append(float(d))
                d = d * g
            rows.append(torch.tensor(vals, dtype=torch.float32))
    return torch.stack(rows).contiguous()
