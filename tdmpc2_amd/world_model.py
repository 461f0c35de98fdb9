"""WorldModel: host-side container with the reference's attribute names and
checkpoint key layout (tdmpc2/common/world_model.py:11-216).

What lives here: parameters (so `load_state_dict` of a reference checkpoint
works), the observation encoder (`encode`, PyTorch-ROCm, as the north star
keeps it), and plain-torch `next/reward/pi/Q` used only outside planning
(`act()` with `mpc=False`).  The planner reads these parameters once, through
`NativePlanner.bind_state_dict`, and runs in HIP.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import checkpoint, layers


def _symexp(x):
    return torch.sign(x) * (torch.exp(torch.abs(x)) - 1)


def two_hot_inv(x, cfg):
    """reference tdmpc2/common/math.py:74-83."""
    if cfg.num_bins == 0:
        return x
    if cfg.num_bins == 1:
        return _symexp(x)
    bins = torch.linspace(cfg.vmin, cfg.vmax, cfg.num_bins, device=x.device, dtype=x.dtype)
    return _symexp(torch.sum(F.softmax(x, dim=-1) * bins, dim=-1, keepdim=True))


class WorldModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        L, M, A, T = cfg.latent_dim, cfg.mlp_dim, cfg.action_dim, cfg.task_dim
        if cfg.multitask:
            self._task_emb = nn.Embedding(len(cfg.tasks), T, max_norm=1)
            self.register_buffer("_action_masks", torch.zeros(len(cfg.tasks), A))
            for i in range(len(cfg.tasks)):
                self._action_masks[i, : cfg.action_dims[i]] = 1.0
        self._encoder = layers.encoders(cfg)
        self._dynamics = layers.mlp(L + A + T, 2 * [M], L, act=layers.SimNorm(cfg.simnorm_dim))
        self._reward = layers.mlp(L + A + T, 2 * [M], max(cfg.num_bins, 1))
        self._termination = layers.mlp(L + T, 2 * [M], 1) if cfg.episodic else None
        self._pi = layers.mlp(L + T, 2 * [M], 2 * A)
        self._Qs = layers.QEnsemble(cfg.num_q, L + A + T, M, max(cfg.num_bins, 1))
        # the reference keeps a detached alias and a target copy of the Q parameters in the
        # state dict (world_model.py:38-53); they are training state, kept here for key parity
        self._detach_Qs_params = layers.StackedMLPParams(cfg.num_q, L + A + T, M, max(cfg.num_bins, 1), as_buffer=True)
        self._target_Qs_params = layers.StackedMLPParams(cfg.num_q, L + A + T, M, max(cfg.num_bins, 1), as_buffer=True)
        self.register_buffer("log_std_min", torch.tensor(float(cfg.log_std_min)))
        self.register_buffer("log_std_dif", torch.tensor(float(cfg.log_std_max)) - self.log_std_min)
        self.apply(self._weight_init)
        self._register_state_dict_hook(self._add_meta_keys)
        self._register_load_state_dict_pre_hook(self._convert_incoming, with_module=True)

    # reference common/init.py:4-11
    @staticmethod
    def _weight_init(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.Embedding):
            nn.init.uniform_(m.weight, -0.02, 0.02)
        elif isinstance(m, layers._StackedLayer):
            nn.init.trunc_normal_(m.weight, std=0.02)

    # ---- state-dict layout ------------------------------------------------------------
    @staticmethod
    def _add_meta_keys(module, state_dict, prefix, local_metadata):
        dev = module.log_std_min.device
        for p in ("_Qs.params.", "_detach_Qs_params.", "_target_Qs_params."):
            state_dict[prefix + p + "__batch_size"] = torch.Size([module.cfg.num_q])
            state_dict[prefix + p + "__device"] = dev
        return state_dict

    @staticmethod
    def _convert_incoming(module, state_dict, prefix, *args):
        incoming = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        conv = checkpoint.convert_state_dict(incoming)
        if checkpoint.is_old_format(incoming):
            # released (old-API) checkpoints predate these buffers; the reference's loader takes them from the freshly
            # constructed model (layers.py:211-215) -- and overwrites any value the file might carry, as done here
            conv["log_std_min"] = module.log_std_min.detach().clone()
            conv["log_std_dif"] = module.log_std_dif.detach().clone()
            if hasattr(module, "_action_masks"):
                conv["_action_masks"] = module._action_masks.detach().clone()
        for k in [k for k in state_dict if k.startswith(prefix)]:
            del state_dict[k]
        for k, v in conv.items():
            state_dict[prefix + k] = v

    def planner_state_dict(self):
        """The tensors the HIP planner binds (new-format keys)."""
        sd = {k: v for k, v in self.state_dict().items() if torch.is_tensor(v)}
        return {k: v for k, v in sd.items()
                if k.startswith(("_dynamics.", "_reward.", "_pi.", "_Qs.params.", "_termination.", "_target_Qs_params."))}

    # ---- forward pieces (reference world_model.py:88-216) ------------------------------
    def task_emb(self, x, task):
        if isinstance(task, int):
            task = torch.tensor([task], device=x.device)
        emb = self._task_emb(task.long())
        if x.ndim == 3:
            emb = emb.unsqueeze(0).repeat(x.shape[0], 1, 1)
        elif emb.shape[0] == 1:
            emb = emb.repeat(x.shape[0], 1)
        return torch.cat([x, emb], dim=-1)

    def encode(self, obs, task):
        """reference world_model.py:103-112, including the [T, B, C, H, W] pixel-sequence branch."""
        if self.cfg.multitask:
            obs = self.task_emb(obs, task)
        if self.cfg.obs == "rgb" and obs.ndim == 5:
            return torch.stack([self._encoder[self.cfg.obs](o) for o in obs])
        return self._encoder[self.cfg.obs](obs)

    def next(self, z, a, task):
        if self.cfg.multitask:
            z = self.task_emb(z, task)
        return self._dynamics(torch.cat([z, a], dim=-1))

    def reward(self, z, a, task):
        if self.cfg.multitask:
            z = self.task_emb(z, task)
        return self._reward(torch.cat([z, a], dim=-1))

    def pi(self, z, task):
        """Returns (action, info) with info['mean'] like the reference (world_model.py:144-184)."""
        if self.cfg.multitask:
            z = self.task_emb(z, task)
        mean, log_std = self._pi(z).chunk(2, dim=-1)
        log_std = self.log_std_min + 0.5 * self.log_std_dif * (torch.tanh(log_std) + 1)
        eps = torch.randn_like(mean)
        if self.cfg.multitask:
            m = self._action_masks[task]
            mean, log_std, eps = mean * m, log_std * m, eps * m
        action = mean + eps * log_std.exp()
        return torch.tanh(action), {"mean": torch.tanh(mean), "log_std": log_std}

    def Q(self, z, a, task, return_type="min"):
        assert return_type in {"min", "avg", "all"}
        if self.cfg.multitask:
            z = self.task_emb(z, task)
        out = self._Qs(torch.cat([z, a], dim=-1))
        if return_type == "all":
            return out
        qidx = torch.randperm(self.cfg.num_q, device=out.device)[:2]
        Q = two_hot_inv(out[qidx], self.cfg)
        return Q.min(0).values if return_type == "min" else Q.sum(0) / 2
