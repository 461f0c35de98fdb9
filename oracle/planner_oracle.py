"""CPU oracle for TD-MPC2's MPPI/CEM planner.  TEST INFRASTRUCTURE ONLY.

This file restates, with plain torch CPU ops, the algorithm of the reference's
planner hot path.  It exists to CHECK the HIP implementation; nothing in the
product path (`tdmpc2_amd/`) imports it.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may use it.

Pinning: the reference ships no golden vectors or tests for this path
(SURVEY.md section 4).  The oracle is therefore pinned against outputs of the
reference's OWN planner code executed verbatim in the build container
(`oracle/ref_runner.py` -> `oracle/make_golden.py` -> `tests/golden/*.npz`);
`tests/test_oracle_golden.py` asserts agreement.

Each function cites the reference lines it follows (paths relative to the
reference repo root).  Arithmetic is fp32 like the reference (`dtype=torch.float64`
is available to attribute error between two fp32 implementations).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- layers
def simnorm(x: torch.Tensor, g: int) -> torch.Tensor:
    """tdmpc2/common/layers.py:84-88 — softmax over contiguous groups of `g`."""
    shp = x.shape
    x = x.view(*shp[:-1], -1, g)
    x = F.softmax(x, dim=-1)
    return x.view(*shp)


def normed_linear(x, w, b, g, beta, act):
    """tdmpc2/common/layers.py:107-112 — act(LayerNorm(Linear(x))); dropout is
    inactive in eval mode (tdmpc2/tdmpc2.py:32)."""
    x = F.linear(x, w, b)
    x = F.layer_norm(x, (w.shape[0],), g, beta, 1e-5)
    return act(x)


def mlp_forward(sd: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, simnorm_dim: Optional[int] = None):
    """tdmpc2/common/layers.py:121-133 — 2x NormedLinear(Mish) + (NormedLinear(SimNorm) | Linear)."""
    i = 0
    while f"{prefix}.{i + 1}.weight" in sd:
        x = normed_linear(x, sd[f"{prefix}.{i}.weight"], sd[f"{prefix}.{i}.bias"],
                          sd[f"{prefix}.{i}.ln.weight"], sd[f"{prefix}.{i}.ln.bias"], F.mish)
        i += 1
    if f"{prefix}.{i}.ln.weight" in sd:
        return normed_linear(x, sd[f"{prefix}.{i}.weight"], sd[f"{prefix}.{i}.bias"],
                             sd[f"{prefix}.{i}.ln.weight"], sd[f"{prefix}.{i}.ln.bias"],
                             lambda t: simnorm(t, simnorm_dim))
    return F.linear(x, sd[f"{prefix}.{i}.weight"], sd[f"{prefix}.{i}.bias"])


def ensemble_forward(sd, prefix, x):
    """tdmpc2/common/layers.py:24-30 — vmap of the same mlp over stacked params
    (leading dim num_q), i.e. batched matmuls.  x: [N, in] -> [nq, N, out]."""
    nq = sd[f"{prefix}.0.weight"].shape[0]
    h = x.unsqueeze(0).expand(nq, *x.shape)
    for i in (0, 1):
        w, b = sd[f"{prefix}.{i}.weight"], sd[f"{prefix}.{i}.bias"]
        h = torch.baddbmm(b.unsqueeze(1), h, w.transpose(1, 2))
        g, beta = sd[f"{prefix}.{i}.ln.weight"], sd[f"{prefix}.{i}.ln.bias"]
        h = F.layer_norm(h, (w.shape[1],), None, None, 1e-5) * g.unsqueeze(1) + beta.unsqueeze(1)
        h = F.mish(h)
    w, b = sd[f"{prefix}.2.weight"], sd[f"{prefix}.2.bias"]
    return torch.baddbmm(b.unsqueeze(1), h, w.transpose(1, 2))


# --------------------------------------------------------------------------- math
def symexp(x):
    """tdmpc2/common/math.py:50-55."""
    return torch.sign(x) * (torch.exp(torch.abs(x)) - 1)


def two_hot_inv(x, cfg):
    """tdmpc2/common/math.py:74-83."""
    if cfg.num_bins == 0:
        return x
    if cfg.num_bins == 1:
        return symexp(x)
    bins = torch.linspace(cfg.vmin, cfg.vmax, cfg.num_bins, dtype=x.dtype, device=x.device)
    x = F.softmax(x, dim=-1)
    x = torch.sum(x * bins, dim=-1, keepdim=True)
    return symexp(x)


def log_std_fn(x, low, dif):
    """tdmpc2/common/math.py:12-13."""
    return low + 0.5 * dif * (torch.tanh(x) + 1)


# --------------------------------------------------------------------------- world model
class OracleModel:
    """Functional restatement of WorldModel.{task_emb,encode,next,reward,pi,Q}
    (tdmpc2/common/world_model.py:88-216) over a state dict in the reference's
    checkpoint key layout."""

    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], dtype=torch.float32, device=None):
        """`device` (default: where the tensors are, i.e. CPU) lets bench.py time the same restatement as stock
        PyTorch-ROCm eager ops on the GPU."""
        self.cfg = cfg
        self.dtype = dtype
        self.sd = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v)
                   for k, v in state_dict.items() if torch.is_tensor(v)}
        if device is not None:
            self.sd = {k: v.to(device) for k, v in self.sd.items()}

    def task_emb(self, x, task: int):
        """world_model.py:88-101; nn.Embedding(max_norm=1) (world_model.py:21)
        rescales a looked-up row with ||row|| > 1 by 1/(||row|| + 1e-7)."""
        if torch.is_tensor(task) and task.numel() > 1:  # one task per row (training batches, world_model.py:95-97)
            emb = self.sd["_task_emb.weight"][task.long()]
            norm = emb.norm(2, dim=-1, keepdim=True)
            emb = torch.where(norm > 1.0, emb * (1.0 / (norm + 1e-7)), emb)
            return torch.cat([x, emb], dim=-1)
        task = int(task)
        emb = self.sd["_task_emb.weight"][task]
        norm = emb.norm(2)
        if norm > 1.0:
            emb = emb * (1.0 / (norm + 1e-7))
        emb = emb.unsqueeze(0).repeat(x.shape[0], 1)
        return torch.cat([x, emb], dim=-1)

    def encode(self, obs, task):
        """world_model.py:103-112 (state observations)."""
        if self.cfg.multitask:
            obs = self.task_emb(obs, task)
        return mlp_forward(self.sd, "_encoder.state", obs, self.cfg.simnorm_dim)

    def next(self, z, a, task):
        """world_model.py:114-121 — concat order is [z, emb, a]."""
        if self.cfg.multitask:
            z = self.task_emb(z, task)
        return mlp_forward(self.sd, "_dynamics", torch.cat([z, a], dim=-1), self.cfg.simnorm_dim)

    def reward(self, z, a, task):
        """world_model.py:123-130."""
        if self.cfg.multitask:
            z = self.task_emb(z, task)
        return mlp_forward(self.sd, "_reward", torch.cat([z, a], dim=-1))

    def termination(self, z, task):
        """world_model.py:132-141."""
        assert task is None
        return torch.sigmoid(mlp_forward(self.sd, "_termination", z))

    def pi(self, z, task, eps):
        """world_model.py:144-184 with the randn_like draw (line 156) supplied
        as `eps`; only the action is returned (the planner ignores `info`)."""
        if self.cfg.multitask:
            z = self.task_emb(z, task)
        mean, log_std = mlp_forward(self.sd, "_pi", z).chunk(2, dim=-1)
        log_std = log_std_fn(log_std, self.sd["log_std_min"], self.sd["log_std_dif"])
        if self.cfg.multitask:
            mask = self.sd["_action_masks"][task.long() if torch.is_tensor(task) else task]
            mean = mean * mask
            log_std = log_std * mask
            eps = eps * mask
        action = mean + eps * log_std.exp()
        return torch.tanh(action)  # math.squash, math.py:23-29

    def Q_avg(self, z, a, task, qidx):
        """world_model.py:186-216, return_type='avg', with randperm(num_q)[:2]
        (line 212) supplied as `qidx`.  All num_q heads are evaluated, as written."""
        if self.cfg.multitask:
            z = self.task_emb(z, task)
        out = ensemble_forward(self.sd, "_Qs.params", torch.cat([z, a], dim=-1))
        Q = two_hot_inv(out[qidx.long()], self.cfg)
        return Q.sum(0) / 2

    def Q_pair(self, z, a, task, qidx, return_type="min", target=False):
        """world_model.py:186-216 for return_type 'min' / 'avg' on the online (`_Qs.params`), or with `target` the
        target (`_target_Qs_params`, world_model.py:38-53) ensemble; `qidx` = randperm(num_q)[:2] (line 212)."""
        if self.cfg.multitask:
            z = self.task_emb(z, task)
        out = ensemble_forward(self.sd, "_target_Qs_params" if target else "_Qs.params", torch.cat([z, a], dim=-1))
        Q = two_hot_inv(out[qidx.long()], self.cfg)
        return Q.min(0).values if return_type == "min" else Q.sum(0) / 2


# --------------------------------------------------------------------------- planner
def estimate_value(model: OracleModel, z, actions, task, discount, pi_eps, qidx):
    """tdmpc2/tdmpc2.py:122-136.  `discount` is the python float (single task)
    or the 0-dim fp32 tensor discount[task] (multitask)."""
    cfg = model.cfg
    G, disc = 0, 1
    termination = torch.zeros(cfg.num_samples, 1, dtype=z.dtype, device=z.device)
    for t in range(cfg.horizon):
        reward = two_hot_inv(model.reward(z, actions[t], task), cfg)
        z = model.next(z, actions[t], task)
        G = G + disc * (1 - termination) * reward
        disc = disc * discount
        if cfg.episodic:
            termination = torch.clip(termination + (model.termination(z, task) > 0.5).to(z.dtype), max=1.0)
    action = model.pi(z, task, pi_eps)
    return G + disc * (1 - termination) * model.Q_avg(z, action, task, qidx)


def td_target(model: OracleModel, next_z, reward, terminated, task, discount, pi_eps, qidx):
    """tdmpc2/tdmpc2.py:239-254 (`TDMPC2._td_target`): next_z [..., L], reward / terminated [..., 1]; the randn_like draw
    of `pi` and the randperm of `Q` are supplied."""
    lead = next_z.shape[:-1]
    z2, e2 = next_z.reshape(-1, next_z.shape[-1]), pi_eps.reshape(-1, pi_eps.shape[-1])
    if torch.is_tensor(task) and task.numel() > 1:  # task [B] for next_z [H, B, L]: one task per row of the flattened batch
        task = task.repeat(next_z.shape[0]) if next_z.dim() == 3 else task
    action = model.pi(z2, task, e2)
    q = model.Q_pair(z2, action, task, qidx, "min", target=True).reshape(*lead, 1)
    return reward + discount * (1 - terminated) * q


def policy_value(model: OracleModel, zs, task, pi_eps, qidx):
    """The forward half of tdmpc2/tdmpc2.py:208-225 (`TDMPC2.update_pi`): action = pi(zs), qs = Q(zs, action, 'avg') on
    the online (detached) ensemble, before the running-scale normalisation."""
    lead = zs.shape[:-1]
    z2, e2 = zs.reshape(-1, zs.shape[-1]), pi_eps.reshape(-1, pi_eps.shape[-1])
    action = model.pi(z2, task, e2)
    q = model.Q_pair(z2, action, task, qidx, "avg", target=False)
    return action.reshape(*lead, -1), q.reshape(*lead, 1)


def refit(cfg, value, actions, action_mask=None):
    """tdmpc2/tdmpc2.py:184-197: nan_to_num, top-k, softmax-weighted mean/std."""
    value = value.nan_to_num(0)
    elite_idxs = torch.topk(value.squeeze(1), cfg.num_elites, dim=0).indices
    elite_value, elite_actions = value[elite_idxs], actions[:, elite_idxs]
    max_value = elite_value.max(0).values
    score = torch.exp(cfg.temperature * (elite_value - max_value))
    score = score / score.sum(0)
    mean = (score.unsqueeze(0) * elite_actions).sum(dim=1) / (score.sum(0) + 1e-9)
    std = ((score.unsqueeze(0) * (elite_actions - mean.unsqueeze(1)) ** 2).sum(dim=1) / (score.sum(0) + 1e-9)).sqrt()
    std = std.clamp(cfg.min_std, cfg.max_std)
    if action_mask is not None:
        mean = mean * action_mask
        std = std * action_mask
    return value, elite_idxs, score, elite_actions, mean, std


def gumbel_select(score, exp_draws):
    """tdmpc2/common/math.py:86-94 with the exponential_() draw supplied."""
    logits = score.log()
    gumbels = -exp_draws.log()
    gumbels = (logits + gumbels) / 1.0
    y_soft = gumbels.softmax(0)
    return y_soft.argmax(-1)


def plan(model: OracleModel, *, z0=None, obs=None, tape: Dict[str, torch.Tensor], prev_mean, t0: bool,
         eval_mode: bool, task: Optional[int], discount, iterations: int):
    """tdmpc2/tdmpc2.py:138-206 for ONE environment, with the RNG draws played
    from `tape` (per-env slices of synth.make_noise_tape).

    Either `obs` [1, obs_dim] (encoded here, tdmpc2.py:153) or the latent
    `z0` [1, L] is given.  Returns (action[A], new_prev_mean[H,A], stages) where
    `stages` holds per-iteration values / elite idx / score / mean / std.
    """
    cfg = model.cfg
    dt = model.dtype
    H, N, P, A = cfg.horizon, cfg.num_samples, cfg.num_pi_trajs, cfg.action_dim
    z = model.encode(obs.to(dt), task) if z0 is None else z0.to(dt)
    mask = model.sd["_action_masks"][task].unsqueeze(0) if cfg.multitask else None
    # policy trajectories, tdmpc2.py:154-160
    pi_actions = torch.empty(H, P, A, dtype=dt, device=z.device)
    if P > 0:
        _z = z.repeat(P, 1)
        for t in range(H - 1):
            pi_actions[t] = model.pi(_z, task, tape["pi_traj_eps"][t].to(dt))
            _z = model.next(_z, pi_actions[t], task)
        pi_actions[-1] = model.pi(_z, task, tape["pi_traj_eps"][H - 1].to(dt))
    # init, tdmpc2.py:163-170
    z = z.repeat(N, 1)
    mean = torch.zeros(H, A, dtype=dt, device=z.device)
    std = torch.full((H, A), cfg.max_std, dtype=dt, device=z.device)
    if not t0:
        mean[:-1] = prev_mean[1:].to(dt)
    actions = torch.empty(H, N, A, dtype=dt, device=z.device)
    if P > 0:
        actions[:, :P] = pi_actions
    stages = {"value": [], "elite_idx": [], "score": [], "mean": [], "std": [], "actions": []}
    for it in range(iterations):  # tdmpc2.py:173-197
        r = tape["sample_eps"][it].to(dt)
        actions_sample = (mean.unsqueeze(1) + std.unsqueeze(1) * r).clamp(-1, 1)
        actions[:, P:] = actions_sample
        if cfg.multitask:
            actions = actions * mask
        value = estimate_value(model, z, actions, task, discount, tape["pi_eps"][it].to(dt), tape["qidx"][it])
        value, elite_idxs, score, elite_actions, mean, std = refit(cfg, value, actions, mask)
        stages["value"].append(value.squeeze(1).clone())
        stages["elite_idx"].append(elite_idxs.clone())
        stages["score"].append(score.squeeze(1).clone())
        stages["mean"].append(mean.clone())
        stages["std"].append(std.clone())
        stages["actions"].append(actions.clone())
    # select, tdmpc2.py:199-206
    rand_idx = gumbel_select(score.squeeze(1), tape["gumbel_exp"].to(dt))
    a = elite_actions[0, rand_idx]
    if not eval_mode:
        a = a + std[0] * tape["final_eps"].to(dt)
    stages = {k: torch.stack(v) for k, v in stages.items()}
    stages["rand_idx"] = rand_idx
    return a.clamp(-1, 1), mean.clone(), stages


def env_tape(tape: Dict, e: int) -> Dict[str, torch.Tensor]:
    """Slice env `e` out of a batched tape (numpy or torch) -> torch tensors."""
    out = {}
    for k, v in tape.items():
        t = torch.as_tensor(v[e])
        out[k] = t
    return out


def plan_batch(model: OracleModel, z0, tape, prev_mean, t0, eval_mode, tasks, discounts, iterations):
    """E independent plans, executed one at a time (the reference has no
    batch-of-envs API: tdmpc2.py:111,163).  z0 [E,L]; prev_mean [E,H,A];
    t0 [E] bool; tasks list[int]|None; discounts list (float | 0-dim tensor)."""
    E = z0.shape[0]
    acts, means, stages = [], [], []
    for e in range(E):
        a, m, st = plan(model, z0=torch.as_tensor(z0[e:e + 1]), tape=env_tape(tape, e),
                        prev_mean=torch.as_tensor(prev_mean[e]), t0=bool(t0[e]), eval_mode=eval_mode,
                        task=None if tasks is None else int(tasks[e]), discount=discounts[e],
                        iterations=iterations)
        acts.append(a)
        means.append(m)
        stages.append(st)
    st = {k: torch.stack([s[k] for s in stages]) for k in stages[0]}
    return torch.stack(acts), torch.stack(means), st


# --------------------------------------------------------------------------- layer-level trace
def _mlp_hidden(sd, prefix, x):
    """The two NormedLinear(Mish) hidden activations of a reference `mlp` (layers.py:121-133)."""
    hs = []
    for i in (0, 1):
        x = normed_linear(x, sd[f"{prefix}.{i}.weight"], sd[f"{prefix}.{i}.bias"], sd[f"{prefix}.{i}.ln.weight"],
                          sd[f"{prefix}.{i}.ln.bias"], F.mish)
        hs.append(x)
    return hs


def trace_estimate_value(model: OracleModel, z, actions, task, discount, pi_eps, qidx):
    """_estimate_value (tdmpc2/tdmpc2.py:122-136) with every intermediate the fused kernel can dump
    (tdmpc2_plan_estimate_value_trace).  Returns (value [N], tiles [5H+7, N, L], scalars [N, H+2+A])."""
    cfg, sd = model.cfg, model.sd
    H = cfg.horizon
    tiles, rs = [], []
    G, disc = 0, 1
    emb = (lambda x: model.task_emb(x, task)) if cfg.multitask else (lambda x: x)
    for t in range(H):
        x = torch.cat([emb(z), actions[t]], dim=-1)
        rh = _mlp_hidden(sd, "_reward", x)
        r = two_hot_inv(F.linear(rh[1], sd["_reward.2.weight"], sd["_reward.2.bias"]), cfg)
        dh = _mlp_hidden(sd, "_dynamics", x)
        z = normed_linear(dh[1], sd["_dynamics.2.weight"], sd["_dynamics.2.bias"], sd["_dynamics.2.ln.weight"],
                          sd["_dynamics.2.ln.bias"], lambda u: simnorm(u, cfg.simnorm_dim))
        tiles += [rh[0], rh[1], dh[0], dh[1], z]
        rs.append(r)
        G = G + disc * r
        disc = disc * discount
    ph = _mlp_hidden(sd, "_pi", emb(z))
    a = model.pi(z, task, pi_eps)
    x = torch.cat([emb(z), a], dim=-1)
    qh, qv = [], []
    for q in qidx.tolist():
        hsd = {k.replace("_Qs.params", "_q"): v[q] for k, v in sd.items() if k.startswith("_Qs.params.")}
        hh = _mlp_hidden(hsd, "_q", x)
        qh += hh
        qv.append(two_hot_inv(F.linear(hh[1], hsd["_q.2.weight"], hsd["_q.2.bias"]), cfg))
    tiles += [ph[0], ph[1], z] + qh
    value = G + disc * ((qv[0] + qv[1]) / 2)
    scalars = torch.cat(rs + qv + [a], dim=-1)
    same = all(t.shape == tiles[0].shape for t in tiles)  # latent_dim == mlp_dim (the fused size class)
    return value.squeeze(1), (torch.stack(tiles) if same else tiles), scalars
