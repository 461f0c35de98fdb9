"""Generate `tests/golden/*.npz` by running the REFERENCE's planner code
verbatim (oracle/ref_runner.py).  Run in the build container only:

    python -m oracle.make_golden            # all cases
    python -m oracle.make_golden tiny c1    # some

TEST INFRASTRUCTURE ONLY.  The fixtures hold reference OUTPUTS (per-iteration
values, elite indices, scores, mean/std, final action, new prev_mean); inputs
are rebuilt from seeds by `oracle/cases.py`.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from oracle import cases, ref_runner
from oracle.planner_oracle import env_tape

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def generate(name: str):
    c = cases.build_case(name)
    cfg = c["cfg"]
    sd = {k: torch.as_tensor(v) for k, v in c["sd"].items()}
    out = {k: [] for k in ("value", "elite_idx", "score", "mean", "std", "action", "prev_mean_out")}
    for e in range(c["n_envs"]):
        a, pm, st = ref_runner.run_reference_plan(
            cfg, sd, z0=c["z0"][e], tape=env_tape(c["tape"], e), prev_mean=c["prev_mean"][e], t0=bool(c["t0"][e]),
            eval_mode=c["eval_mode"], task=None if c["tasks"] is None else c["tasks"][e],
            discount=_ref_discount(c, e), iterations=c["iterations"])
        out["value"].append(st["value"].numpy())
        out["elite_idx"].append(st["elite_idx"].numpy().astype(np.int32))
        out["score"].append(st["score"].numpy())
        out["mean"].append(st["mean"].numpy())
        out["std"].append(st["std"].numpy())
        out["action"].append(a.numpy())
        out["prev_mean_out"].append(pm.numpy())
    # encoder golden: reference encode() on synthetic observations
    from tdmpc2_amd import synth
    obs = synth.make_obs(cfg, c["n_envs"], seed=3)
    agent = ref_runner.build_agent(cfg, sd, 0.99)
    with torch.no_grad():
        zs = [agent.model.encode(torch.as_tensor(obs[e:e + 1]),
                                 None if c["tasks"] is None else torch.tensor([c["tasks"][e]]))[0].numpy()
              for e in range(c["n_envs"])]
    arrs = {k: np.stack(v) for k, v in out.items()}
    arrs["encode_z"] = np.stack(zs)
    # TDMPC2._td_target on a synthetic [H, B] batch (inputs rebuilt by cases.td_batch)
    arrs["td_target"] = _td_target_reference(c, cfg, sd)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, f"{name}.npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: wrote {path} ({os.path.getsize(path)/1024:.1f} KiB)  value range "
          f"[{arrs['value'].min():.3f}, {arrs['value'].max():.3f}]")


def add_td_target(name: str):
    """Add the `td_target` array (reference `TDMPC2._td_target` on cases.td_batch) to an existing fixture."""
    c = cases.build_case(name)
    cfg = c["cfg"]
    sd = {k: torch.as_tensor(v) for k, v in c["sd"].items()}
    path = os.path.join(GOLDEN_DIR, f"{name}.npz")
    arrs = dict(np.load(path))
    arrs["td_target"] = _td_target_reference(c, cfg, sd)
    np.savez_compressed(path, **arrs)
    print(f"{name}: td_target {arrs['td_target'].shape} range [{arrs['td_target'].min():.3f}, {arrs['td_target'].max():.3f}]")


def _td_target_reference(c, cfg, sd):
    """The reference's `_td_target` (tdmpc2.py:239-254) on cases.td_batch; multitask: one task per batch column and the
    per-task discount vector the reference holds in self.discount (tdmpc2.py:35-37)."""
    from tdmpc2_amd.config import get_discount

    tb = cases.td_batch(cfg)
    if cfg.multitask:
        task = torch.as_tensor(tb["tasks"])
        discount = torch.tensor([get_discount(cfg, ln) for ln in cfg.episode_lengths])
    else:
        task, discount = None, _ref_discount(c, 0)
    return ref_runner.run_reference_td_target(cfg, sd, next_z=tb["next_z"], reward=tb["reward"], terminated=tb["terminated"],
                                              task=task, discount=discount, pi_eps=tb["pi_eps"], qidx=tb["qidx"]).numpy()


def _ref_discount(c, e):
    """What the reference holds in `self.discount` (tdmpc2/tdmpc2.py:35-37): a
    python float (single task) or a tensor over tasks (multitask)."""
    cfg = c["cfg"]
    if cfg.multitask:
        from tdmpc2_amd.config import get_discount
        return torch.tensor([get_discount(cfg, L) for L in cfg.episode_lengths])
    return c["discounts"][e]


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--td-only":
    for n in sys.argv[2:]:
        add_td_target(n)
    sys.exit(0)

if __name__ == "__main__":
    names = sys.argv[1:] or list(cases.CASES)
    for n in names:
        generate(n)
