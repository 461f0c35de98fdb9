"""Run the REFERENCE's own planner code, verbatim, on CPU.  TEST INFRASTRUCTURE ONLY.

Works only where `/root/reference` exists (the build container).  Nothing is
copied: the reference's modules are imported from where they lie, with a stub
`tensordict` in `sys.modules` (the real package is absent and is used by the
planner only for parameter stacking).  Recipe: SURVEY.md Appendix A.

Executed verbatim from the reference: `TDMPC2._plan`, `TDMPC2._estimate_value`
(tdmpc2/tdmpc2.py:122-206), `WorldModel.{task_emb,encode,next,reward,
termination,pi,Q}` (tdmpc2/common/world_model.py:88-216), `layers.mlp /
NormedLinear / SimNorm / enc` (tdmpc2/common/layers.py:74-164) and
`common.math` (tdmpc2/common/math.py).  The only restated piece is the Q
`Ensemble` container (layers.py:8-33 needs real tensordict): the same vmap over
stacked parameters via `torch.func`.

RNG: the six draw sites are served from a noise tape by
patching `torch.randn`, `torch.randn_like`, `torch.randperm` and
`Tensor.exponential_` while `_plan` runs — reference code untouched.
Per-iteration locals (value, elite idx, score, mean, std) are read with
`sys.settrace`, again without touching the reference.
"""
from __future__ import annotations

import copy
import inspect
import os
import sys
import types
from typing import Dict, Optional

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("TDMPC2_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "tdmpc2", "common"))


_ref = None


def _import_reference():
    global _ref
    if _ref is not None:
        return _ref
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    td = types.ModuleType("tensordict")
    td.from_modules = None
    td.TensorDict = dict  # pi() builds TensorDict({...}); a dict is enough
    tdn = types.ModuleType("tensordict.nn")
    tdn.TensorDictParams = object
    sys.modules.setdefault("tensordict", td)
    sys.modules.setdefault("tensordict.nn", tdn)
    path = os.path.join(REF_ROOT, "tdmpc2")
    saved = list(sys.path)
    saved_mods = {k: sys.modules.get(k) for k in ("common", "tdmpc2")}
    sys.path.insert(0, path)
    try:
        for k in [m for m in sys.modules if m == "common" or m.startswith("common.")]:
            del sys.modules[k]
        import importlib

        layers = importlib.import_module("common.layers")
        rmath = importlib.import_module("common.math")
        wm = importlib.import_module("common.world_model")
        # `tdmpc2` here is the reference's tdmpc2/tdmpc2.py module
        spec = importlib.util.spec_from_file_location("_ref_tdmpc2_module", os.path.join(path, "tdmpc2.py"))
        # tdmpc2.py imports common.scale (hard-codes cuda in its ctor only; import is fine)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path[:] = saved
    _ref = types.SimpleNamespace(layers=layers, math=rmath, WorldModel=wm.WorldModel, TDMPC2=mod.TDMPC2)
    return _ref


class _Ens(nn.Module):
    """Stand-in for layers.Ensemble (layers.py:8-33): vmap over stacked params."""

    def __init__(self, mods):
        super().__init__()
        from torch.func import stack_module_state

        p, _ = stack_module_state(mods)
        self.base = copy.deepcopy(mods[0]).to("meta")
        self.p = nn.ParameterDict({k.replace(".", "__"): nn.Parameter(v) for k, v in p.items()})

    def forward(self, x):
        from torch.func import functional_call

        P = {k.replace("__", "."): v for k, v in self.p.items()}
        return torch.vmap(lambda p, x: functional_call(self.base, (p,), (x,)), (0, None), randomness="different")(P, x)


def build_agent(cfg, state_dict: Dict[str, torch.Tensor], discount):
    """Duck-typed host objects carrying the reference's unbound methods."""
    ref = _import_reference()
    layers, WorldModel, TDMPC2 = ref.layers, ref.WorldModel, ref.TDMPC2

    class WM(nn.Module):
        def __init__(s):
            super().__init__()
            s.cfg = cfg
            D = cfg.latent_dim + cfg.action_dim + cfg.task_dim
            if cfg.multitask:
                s._task_emb = nn.Embedding(len(cfg.tasks), cfg.task_dim, max_norm=1)
                s.register_buffer("_action_masks", torch.zeros(len(cfg.tasks), cfg.action_dim))
            s._encoder = layers.enc(cfg, out={})
            s._dynamics = layers.mlp(D, 2 * [cfg.mlp_dim], cfg.latent_dim, act=layers.SimNorm(cfg))
            s._reward = layers.mlp(D, 2 * [cfg.mlp_dim], max(cfg.num_bins, 1))
            s._termination = layers.mlp(cfg.latent_dim + cfg.task_dim, 2 * [cfg.mlp_dim], 1) if cfg.episodic else None
            s._pi = layers.mlp(cfg.latent_dim + cfg.task_dim, 2 * [cfg.mlp_dim], 2 * cfg.action_dim)
            s._Qs = _Ens([layers.mlp(D, 2 * [cfg.mlp_dim], max(cfg.num_bins, 1), dropout=cfg.dropout)
                          for _ in range(cfg.num_q)])
            # the target / detached ensembles of world_model.py:38-53 (there: deep copies of _Qs holding separate
            # parameter sets); WorldModel.Q picks one by its `target` / `detach` flags (world_model.py:201-206)
            s._target_Qs = _Ens([layers.mlp(D, 2 * [cfg.mlp_dim], max(cfg.num_bins, 1), dropout=cfg.dropout)
                                 for _ in range(cfg.num_q)])
            s._detach_Qs = s._Qs
            s.register_buffer("log_std_min", torch.tensor(float(cfg.log_std_min)))
            s.register_buffer("log_std_dif", torch.tensor(float(cfg.log_std_max)) - s.log_std_min)

    for m in ["task_emb", "encode", "next", "reward", "termination", "pi", "Q"]:
        setattr(WM, m, getattr(WorldModel, m))

    class Agent(nn.Module):
        def __init__(s):
            super().__init__()
            s.cfg = cfg
            s.device = torch.device("cpu")
            s.model = WM().eval()
            s.discount = discount
            s._prev_mean = torch.nn.Buffer(torch.zeros(cfg.horizon, cfg.action_dim))

    Agent._plan = TDMPC2._plan
    Agent._estimate_value = TDMPC2._estimate_value
    Agent._td_target = TDMPC2._td_target
    agent = Agent()
    # load weights (checkpoint keys -> host modules)
    own = agent.model.state_dict()
    with torch.no_grad():
        for k, v in state_dict.items():
            if k.startswith("_Qs.params."):
                kk = "_Qs.p." + k[len("_Qs.params."):].replace(".", "__")
            elif k.startswith("_target_Qs_params."):
                kk = "_target_Qs.p." + k[len("_target_Qs_params."):].replace(".", "__")
            elif k.startswith("_detach_Qs_params."):
                continue  # an alias of _Qs.params in the reference (world_model.py:40)
            else:
                kk = k
            if kk not in own:
                raise KeyError(f"{k} -> {kk} not in reference host model")
            own[kk].copy_(torch.as_tensor(v))
    return agent


class TapePlayer:
    """Serves the reference's RNG calls from a per-env tape, in call order."""

    def __init__(self, cfg, tape: Dict[str, torch.Tensor], iterations: int, eval_mode: bool):
        self.q = []
        H = cfg.horizon
        if cfg.num_pi_trajs > 0:
            for t in range(H):
                self.q.append(("randn_like", tape["pi_traj_eps"][t]))
        for it in range(iterations):
            self.q.append(("randn", tape["sample_eps"][it]))
            self.q.append(("randn_like", tape["pi_eps"][it]))
            self.q.append(("randperm", tape["qidx"][it]))
        self.q.append(("exponential_", tape["gumbel_exp"]))
        if not eval_mode:
            self.q.append(("randn", tape["final_eps"]))
        self.pos = 0
        self.num_q = cfg.num_q

    def _next(self, kind, shape=None):
        k, v = self.q[self.pos]
        assert k == kind, f"RNG call #{self.pos}: reference called {kind}, tape expected {k}"
        if shape is not None:
            assert tuple(v.shape) == tuple(shape), f"RNG call #{self.pos} ({kind}): shape {tuple(shape)} vs tape {tuple(v.shape)}"
        self.pos += 1
        return torch.as_tensor(v).clone()

    def __enter__(self):
        self._saved = (torch.randn, torch.randn_like, torch.randperm, torch.Tensor.exponential_)
        player = self

        def randn(*size, **kw):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            return player._next("randn", size).to(torch.float32)

        def randn_like(x, **kw):
            return player._next("randn_like", x.shape).to(x.dtype)

        def randperm(n, **kw):
            first = player._next("randperm").long()
            rest = torch.tensor([i for i in range(n) if i not in first.tolist()], dtype=torch.long)
            return torch.cat([first, rest])

        def exponential_(self_t, *a, **kw):
            self_t.copy_(player._next("exponential_", self_t.shape))
            return self_t

        torch.randn, torch.randn_like, torch.randperm = randn, randn_like, randperm
        torch.Tensor.exponential_ = exponential_
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like, torch.randperm, torch.Tensor.exponential_ = self._saved
        if exc[0] is None:
            assert self.pos == len(self.q), f"reference consumed {self.pos} of {len(self.q)} tape entries"
        return False


def run_reference_td_target(cfg, state_dict, *, next_z, reward, terminated, task, discount, pi_eps, qidx):
    """Call the reference's `TDMPC2._td_target` (tdmpc2/tdmpc2.py:239-254) once with its two RNG draws (the policy's
    randn_like, world_model.py:156, and Q's randperm, world_model.py:212) served from the arguments.
    next_z [H, B, L], reward / terminated [H, B, 1]; `discount` = what the reference holds in self.discount."""
    agent = build_agent(cfg, state_dict, discount)
    saved = (torch.randn_like, torch.randperm)

    def randn_like(x, **kw):
        assert tuple(x.shape) == tuple(pi_eps.shape)
        return torch.as_tensor(pi_eps).to(x.dtype).clone()

    def randperm(n, **kw):
        first = torch.as_tensor(qidx).long()
        rest = torch.tensor([i for i in range(n) if i not in first.tolist()], dtype=torch.long)
        return torch.cat([first, rest])

    torch.randn_like, torch.randperm = randn_like, randperm
    try:
        with torch.no_grad():
            task_t = None if task is None else torch.as_tensor(task)
            return agent._td_target(torch.as_tensor(next_z), torch.as_tensor(reward), torch.as_tensor(terminated), task_t)
    finally:
        torch.randn_like, torch.randperm = saved


def run_reference_plan(cfg, state_dict, *, obs=None, z0=None, tape, prev_mean, t0, eval_mode, task, discount,
                       iterations):
    """Call the reference's `_plan` once.  If `z0` is given, `encode` is
    bypassed (the planner is benchmarked on synthetic latents) by handing the
    model an `encode` that returns it.  Returns (action, new_prev_mean, stages)."""
    agent = build_agent(cfg, state_dict, discount)
    cfg_run = copy.copy(cfg)
    cfg_run.iterations = iterations
    agent.cfg = cfg_run
    agent.model.cfg = cfg_run
    with torch.no_grad():
        agent._prev_mean.copy_(torch.as_tensor(prev_mean))
    if z0 is not None:
        zz = torch.as_tensor(z0).reshape(1, -1).clone()
        agent.model.encode = lambda o, tk: zz
        obs = torch.zeros(1, 1)
    else:
        obs = torch.as_tensor(obs).reshape(1, -1)
    task_t = None if task is None else torch.tensor([task])

    plan_code = agent._plan.__func__.__wrapped__.__code__ if hasattr(agent._plan.__func__, "__wrapped__") \
        else agent._plan.__func__.__code__
    snaps = []

    def tracer(frame, event, arg):
        if frame.f_code is not plan_code:
            return None

        def local(frame, event, arg):
            if event == "line":
                loc = frame.f_locals
                src_line = lines[frame.f_lineno - first].strip()
                # `std` is final for an iteration when control is back at the loop
                # header (next iteration / exit check) or at the first line after the loop.
                at_header = src_line.startswith("for _ in range")
                at_exit = src_line.startswith("rand_idx")
                if "score" in loc and "elite_idxs" in loc and (at_header or (at_exit and len(snaps) < iterations)):
                    snaps.append({k: loc[k].detach().clone() for k in
                                  ("value", "elite_idxs", "score", "mean", "std", "actions")})
            return local

        return local

    src, first = inspect.getsourcelines(ref_plan_function())
    lines = src
    with TapePlayer(cfg_run, tape, iterations, eval_mode):
        sys.settrace(tracer)
        try:
            with torch.no_grad():
                a = agent._plan(obs, t0=t0, eval_mode=eval_mode, task=task_t)
        finally:
            sys.settrace(None)
    assert len(snaps) == iterations, f"captured {len(snaps)} iteration snapshots, expected {iterations}"
    stages = {
        "value": torch.stack([s["value"].squeeze(1) for s in snaps]),
        "elite_idx": torch.stack([s["elite_idxs"] for s in snaps]),
        "score": torch.stack([s["score"].squeeze(1) for s in snaps]),
        "mean": torch.stack([s["mean"] for s in snaps]),
        "std": torch.stack([s["std"] for s in snaps]),
        "actions": torch.stack([s["actions"] for s in snaps]),
    }
    return a.detach().clone(), agent._prev_mean.detach().clone(), stages


def time_reference_plan(cfg, state_dict, *, z0, tape, task, discount, iterations, budget_s=6.0, min_plans=3):
    """Wall time per call of the reference's `_plan` run verbatim (no line tracer, the agent built once, the recorded noise
    replayed per call) -- bench.py's measured `port_vs_reference` when the reference tree is on the machine.
    Returns (ms per plan, plans timed)."""
    import time

    agent = build_agent(cfg, state_dict, discount)
    cfg_run = copy.copy(cfg)
    cfg_run.iterations = iterations
    agent.cfg = cfg_run
    agent.model.cfg = cfg_run
    zz = torch.as_tensor(z0).reshape(1, -1).clone()
    agent.model.encode = lambda o, tk: zz
    obs = torch.zeros(1, 1)
    task_t = None if task is None else torch.tensor([task])

    def one(t0):
        with TapePlayer(cfg_run, tape, iterations, False):
            with torch.no_grad():
                agent._plan(obs, t0=t0, eval_mode=False, task=task_t)

    one(True)
    n, t_start = 0, time.perf_counter()
    while n < min_plans or time.perf_counter() - t_start < budget_s:
        one(False)
        n += 1
    return 1e3 * (time.perf_counter() - t_start) / n, n


def ref_plan_function():
    ref = _import_reference()
    f = ref.TDMPC2._plan
    return getattr(f, "__wrapped__", f)
