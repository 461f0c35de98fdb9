"""TDMPC2: the drop-in boundary (reference tdmpc2/tdmpc2.py:11-120,138-206).

Same constructor argument (`cfg`), same `act(obs, t0, eval_mode, task)` /
`plan(obs, t0, eval_mode, task)` / `load(fp)` / `save(fp)` signatures, same
`model` attribute names and checkpoint layout, same `_prev_mean` buffer — so
the reference's `evaluate.py:57-59,80` works unchanged with this class.  The
planning itself (everything of `_plan` after `encode`) runs in the HIP library;
there is no PyTorch or CPU fallback for it.

Extension over the reference (whose planner is hard-wired to one environment,
tdmpc2.py:111,163): `act_batch` / `plan_batch` plan E independent environments
in one call — the vectorised-env case the north star shards across GPUs.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import checkpoint
from .config import get_discount
from .native import NativePlanner
from .world_model import WorldModel


def _rank() -> int:
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return int(torch.distributed.get_rank())
    return int(os.environ.get("RANK", 0))


class TDMPC2(torch.nn.Module):
    def __init__(self, cfg, device: Optional[torch.device] = None, max_envs: int = 1):
        super().__init__()
        self.cfg = cfg
        if device is None:
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))  # reference: cuda:0
        self.device = torch.device(device)
        self.model = WorldModel(cfg).to(self.device)
        self.model.eval()
        self.cfg.iterations += 2 * int(cfg.action_dim >= 20)  # reference tdmpc2.py:34
        if cfg.multitask:  # reference tdmpc2.py:35-37
            self.discount = torch.tensor([get_discount(cfg, ep_len) for ep_len in cfg.episode_lengths],
                                         device=self.device)
        else:
            self.discount = get_discount(cfg, cfg.episode_length)
        self._prev_mean = torch.nn.Buffer(torch.zeros(cfg.horizon, cfg.action_dim, device=self.device))
        self.max_envs = int(max_envs)
        self.native_encoder = True  # False: encode with the PyTorch-ROCm module (the parity tests compare both)
        self._planner: Optional[NativePlanner] = None
        self._planner_log_std = None
        self._prev_mean_batch = None
        # Philox stream of the in-library noise: cfg.seed in the low word, the rank in the high word, so that env-sharded
        # ranks built from one cfg (tdmpc2_amd/dist.py) do not draw identical exploration noise; a per-call counter is
        # added on top.  (torch.manual_seed does not reach the planner: its RNG lives in the kernels.)
        self._seed = (_rank() << 32) ^ (int(getattr(cfg, "seed", 0)) & 0xFFFFFFFF)
        self.noise_tape = None  # optional dict of device tensors (tdmpc2_noise) for reproducible plans
        self._one = torch.ones(1, dtype=torch.uint8, device=self.device) if self.device.type == "cuda" else None
        self._zero = torch.zeros(1, dtype=torch.uint8, device=self.device) if self.device.type == "cuda" else None

    # ------------------------------------------------------------------ checkpoint I/O
    def save(self, fp):
        """reference tdmpc2.py:72-79."""
        torch.save({"model": self.model.state_dict()}, fp)

    def load(self, fp):
        """reference tdmpc2.py:81-95: path or dict; old- and new-format Q keys."""
        if isinstance(fp, dict):
            state_dict = fp
        else:
            state_dict = torch.load(fp, map_location=self.device, weights_only=False)
        state_dict = state_dict["model"] if "model" in state_dict else state_dict
        # key conversion (old API -> new, tensordict meta entries dropped, buffers an old file lacks filled from this
        # model as reference layers.py:167-221 does) happens in WorldModel's load_state_dict pre-hook
        self.model.load_state_dict(dict(state_dict))
        self.sync_planner_weights()

    # ------------------------------------------------------------------ native planner
    def planner(self) -> NativePlanner:
        # (log_std_min / log_std_dif are constants of the handle; they can only change in load() / sync_planner_weights(),
        # which drop a stale handle -- no device read, hence no stream sync, on the planning path)
        if self._planner is None:
            self._planner_log_std = self._log_std()
            self._planner = NativePlanner(self.cfg, self.cfg.iterations, self.device, max_envs=self.max_envs,
                                          log_std_min=self._planner_log_std[0], log_std_dif=self._planner_log_std[1])
            self._planner.bind_state_dict(self.model.planner_state_dict())
            self._bind_encoder()
        return self._planner

    def _log_std(self):
        """(log_std_min, log_std_dif) as host floats: two device reads -- only where the weights can change."""
        return float(self.model.log_std_min), float(self.model.log_std_dif)

    def _bind_encoder(self):
        # state observations: WorldModel.encode runs inside the library as well (include/tdmpc2_plan.h,
        # tdmpc2_plan_run_obs); pixel observations are encoded by the PyTorch-ROCm conv module (layers.conv) and enter
        # the library as latents (tdmpc2_plan_run)
        if self.native_encoder and self.cfg.obs == "state":
            sd = {k: v for k, v in self.model.state_dict().items() if torch.is_tensor(v) and k.startswith("_encoder.state.")}
            self._planner.bind_encoder(sd)

    def sync_planner_weights(self):
        """Re-pack the model's current weights into the planner (after load / a training step)."""
        if self._planner is not None and self._planner_log_std != self._log_std():
            self._planner.close()
            self._planner = None  # rebuilt (with the new constants and weights) by the next planner() call
        if self._planner is not None:
            self._planner.bind_state_dict(self.model.planner_state_dict())
            self._bind_encoder()

    def _disc_pow(self, tasks):
        """discount^0..discount^H exactly as tdmpc2.py:126,130-132 accumulates it: python-float
        products (single task) or fp32 tensor products (multitask)."""
        H = self.cfg.horizon
        if self.cfg.multitask:
            g = self.discount[tasks.long()].to(torch.float32)  # [E]
            cols = [torch.ones_like(g)]
            for _ in range(H):
                cols.append(cols[-1] * g)
            return torch.stack(cols, dim=1).contiguous()
        d, vals = 1, []
        for _ in range(H + 1):
            vals.append(float(d))
            d = d * self.discount
        return torch.tensor(vals, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ reference API
    @property
    def plan(self):
        """reference tdmpc2.py:45-55 (there: optionally torch.compile'd; here: the HIP planner)."""
        return self._plan

    @torch.no_grad()
    def act(self, obs, t0=False, eval_mode=False, task=None):
        """reference tdmpc2.py:97-120."""
        obs = obs.to(self.device, non_blocking=True).unsqueeze(0)
        if task is not None:
            task = torch.tensor([task], device=self.device)
        if self.cfg.mpc:
            a = self.plan(obs, t0=t0, eval_mode=eval_mode, task=task).cpu()
            if self._planner is not None and self._planner.take_fault():
                # a cluster hand-over of THIS plan gave up (another process / kernel held the compute units): the library
                # returned NaN and left _prev_mean alone; it has switched to the path without hand-overs -- plan again
                a = self.plan(obs, t0=t0, eval_mode=eval_mode, task=task).cpu()
            return a
        z = self.model.encode(obs, task)
        action, info = self.model.pi(z, task)
        if eval_mode:
            action = info["mean"]
        return action[0].cpu()

    @torch.no_grad()
    def _plan(self, obs, t0=False, eval_mode=False, task=None):
        """reference tdmpc2.py:138-206.  obs [1, obs_dim] on device; task int64[1] or None."""
        t0 = self._one if t0 else self._zero
        prev = self._prev_mean.view(1, *self._prev_mean.shape)
        if self.native_encoder and self.cfg.obs == "state":
            return self._plan_obs(obs.to(torch.float32).contiguous(), t0, eval_mode, task, prev)[0]
        z = self.model.encode(obs, task)  # PyTorch-ROCm encoder (pixels, or native_encoder = False)
        return self._plan_latent(z.contiguous(), t0, eval_mode, task, prev)[0]

    # ------------------------------------------------------------------ vectorised extension
    @torch.no_grad()
    def plan_batch(self, obs, t0, eval_mode=False, tasks=None):
        """E environments at once.  obs [E, obs_dim]; t0 bool[E] (or bool); tasks int64[E] or None."""
        obs = obs.to(self.device)
        E = obs.shape[0]
        tasks = torch.as_tensor(tasks, device=self.device).long() if self.cfg.multitask else None
        if self._prev_mean_batch is None or self._prev_mean_batch.shape[0] != E:
            self._prev_mean_batch = torch.zeros(E, self.cfg.horizon, self.cfg.action_dim, device=self.device)
        if isinstance(t0, bool):
            t0 = torch.full((E,), int(t0), dtype=torch.uint8, device=self.device)
        else:
            t0 = torch.as_tensor(t0, device=self.device).to(torch.uint8)
        if self.native_encoder and self.cfg.obs == "state":
            return self._plan_obs(obs.to(torch.float32).contiguous(), t0, eval_mode, tasks, self._prev_mean_batch)
        if self.cfg.multitask:
            emb = self.model._task_emb(tasks)  # max_norm renorm happens inside the lookup
            z = self.model._encoder[self.cfg.obs](torch.cat([obs, emb], dim=-1))
        else:
            z = self.model._encoder[self.cfg.obs](obs)
        return self._plan_latent(z.contiguous(), t0, eval_mode, tasks, self._prev_mean_batch)

    @torch.no_grad()
    def act_batch(self, obs, t0, eval_mode=False, tasks=None):
        a = self.plan_batch(obs, t0, eval_mode, tasks).cpu()
        if self._planner is not None and self._planner.take_fault():  # see act()
            a = self.plan_batch(obs, t0, eval_mode, tasks).cpu()
        return a

    def _plan_inputs(self, E, tasks):
        planner = self.planner()
        if E > planner.max_envs:
            raise ValueError(f"{E} environments exceed max_envs={planner.max_envs} given at construction")
        emb = mask = None
        if self.cfg.multitask:
            emb = self.model._task_emb(tasks.long()).to(torch.float32).contiguous()
            mask = self.model._action_masks[tasks.long()].contiguous()
            disc = self._disc_pow(tasks)
        else:
            disc = self._disc_pow(None).unsqueeze(0).repeat(E, 1).contiguous()
        self._seed += 1
        return planner, emb, mask, disc

    def _plan_obs(self, obs, t0, eval_mode, tasks, prev_mean):
        """encode + plan inside the library (tdmpc2_plan_run_obs)."""
        planner, emb, mask, disc = self._plan_inputs(obs.shape[0], tasks)
        return planner.plan_obs(obs, disc, prev_mean, t0, eval_mode=eval_mode, task_emb=emb, act_mask=mask,
                                tape=self.noise_tape, seed=self._seed)

    def _plan_latent(self, z, t0, eval_mode, tasks, prev_mean):
        planner, emb, mask, disc = self._plan_inputs(z.shape[0], tasks)
        return planner.plan(z.to(torch.float32), disc, prev_mean, t0, eval_mode=eval_mode, task_emb=emb,
                            act_mask=mask, tape=self.noise_tape, seed=self._seed)

    # ------------------------------------------------------------------ training-side forward
    def _task_tables(self):
        """The per-task tables a training batch indexes with `task` (world_model.py:88-101; tdmpc2.py:35-37)."""
        w = self.model._task_emb.weight.detach().to(torch.float32)
        n = w.norm(2, dim=-1, keepdim=True)
        emb = torch.where(n > 1.0, w * (1.0 / (n + 1e-7)), w).contiguous()  # nn.Embedding(max_norm=1) at lookup
        return emb, self.model._action_masks.to(torch.float32).contiguous(), self.discount.to(torch.float32).contiguous()

    @torch.no_grad()
    def _td_target(self, next_z, reward, terminated, task=None, pi_eps=None, qidx=None):
        """reference tdmpc2.py:239-254 (already `@torch.no_grad()` there): the TD target of a training batch on the planner's
        kernels.  next_z [H, B, L] (or [R, L]), reward / terminated [H, B, 1]; task int64 [B] for multitask models (the
        reference broadcasts it over the H leading rows).  `pi_eps` / `qidx` pin the policy noise and the two target heads
        (parity tests); by default they are drawn inside the library."""
        lead = next_z.shape[:-1]
        z2 = next_z.reshape(-1, next_z.shape[-1]).to(self.device, torch.float32).contiguous()
        r2 = reward.reshape(-1).to(self.device, torch.float32).contiguous()
        t2 = terminated.reshape(-1).to(self.device, torch.float32).contiguous()
        kw = {}
        if self.cfg.multitask:
            task = torch.as_tensor(task, device=self.device)
            if next_z.dim() == 3 and task.numel() == next_z.shape[1]:
                task = task.repeat(next_z.shape[0])
            emb, mask, disc = self._task_tables()
            kw = dict(task_ids=task.to(torch.int32).contiguous(), task_emb_table=emb, act_mask_table=mask)
            discount = disc
        else:
            discount = self.discount
        if pi_eps is not None:
            pi_eps = pi_eps.reshape(-1, pi_eps.shape[-1]).to(self.device, torch.float32).contiguous()
        self._seed += 1
        td = self.planner().td_target(z2, r2, t2, discount, pi_eps=pi_eps, qidx=qidx, seed=self._seed, **kw)
        return td.reshape(*lead, 1)

    @torch.no_grad()
    def policy_value(self, zs, task=None, reduce="avg", target=False, pi_eps=None, qidx=None):
        """The forward half of `update_pi` (reference tdmpc2.py:208-225): action = pi(zs), q = Q(zs, action, 'avg') on the
        online ensemble -- returned as (action [..., A], q [..., 1]); gradients are outside this package's scope."""
        lead = zs.shape[:-1]
        z2 = zs.reshape(-1, zs.shape[-1]).to(self.device, torch.float32).contiguous()
        kw = {}
        if self.cfg.multitask:
            task = torch.as_tensor(task, device=self.device)
            if zs.dim() == 3 and task.numel() == zs.shape[1]:
                task = task.repeat(zs.shape[0])
            emb, mask, _ = self._task_tables()
            kw = dict(task_ids=task.to(torch.int32).contiguous(), task_emb_table=emb, act_mask_table=mask)
        if pi_eps is not None:
            pi_eps = pi_eps.reshape(-1, pi_eps.shape[-1]).to(self.device, torch.float32).contiguous()
        self._seed += 1
        a, q = self.planner().policy_value(z2, use_target=target, reduce=reduce, pi_eps=pi_eps, qidx=qidx, seed=self._seed, **kw)
        return a.reshape(*lead, -1), q.reshape(*lead, 1)
