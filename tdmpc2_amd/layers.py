"""Host-side (PyTorch) building blocks of the world model.

These exist for what stays on the host side of the boundary: the observation
encoder, checkpoint I/O in the reference's key layout, and the non-planning
`act()` path.  The planner itself never runs through them — it runs in the HIP
library (tdmpc2_amd/csrc).  Behaviour follows tdmpc2/common/layers.py:74-164 of
the reference.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class SimNorm(nn.Module):
    """Softmax over contiguous groups of `dim` features (reference layers.py:74-91)."""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        shp = x.shape
        return F.softmax(x.view(*shp[:-1], -1, self.dim), dim=-1).view(*shp)

    def __repr__(self):
        return f"SimNorm(dim={self.dim})"


class NormedLinear(nn.Linear):
    """Linear -> LayerNorm -> activation (Mish unless given); reference layers.py:94-118.
    Dropout only matters in training and is kept for state-dict-free parity of behaviour."""

    def __init__(self, in_features, out_features, dropout: float = 0.0, act=None):
        super().__init__(in_features, out_features)
        self.ln = nn.LayerNorm(out_features)
        self.act = act if act is not None else nn.Mish()
        self.dropout = nn.Dropout(dropout) if dropout else None

    def forward(self, x):
        x = super().forward(x)
        if self.dropout is not None:
            x = self.dropout(x)
        return self.act(self.ln(x))


def mlp(in_dim, mlp_dims, out_dim, act=None, dropout=0.0):
    """Reference layers.py:121-133: NormedLinear stack, last layer plain Linear unless `act`."""
    if isinstance(mlp_dims, int):
        mlp_dims = [mlp_dims]
    dims = [in_dim] + list(mlp_dims) + [out_dim]
    mods = [NormedLinear(dims[i], dims[i + 1], dropout=dropout * (i == 0)) for i in range(len(dims) - 2)]
    mods.append(NormedLinear(dims[-2], dims[-1], act=act) if act is not None else nn.Linear(dims[-2], dims[-1]))
    return nn.Sequential(*mods)


def state_encoder(cfg):
    """Reference layers.py:153-164, 'state' branch only (pixel encoders are outside the hot-path scope)."""
    out = {}
    for k, shape in cfg.obs_shape.items():
        if k != "state":
            raise NotImplementedError(f"encoder for observation type {k!r} is outside this package's scope")
        out[k] = mlp(shape[0] + cfg.task_dim, max(cfg.num_enc_layers - 1, 1) * [cfg.enc_dim], cfg.latent_dim,
                     act=SimNorm(cfg.simnorm_dim))
    return nn.ModuleDict(out)


class _StackedLayer(nn.Module):
    def __init__(self, n, out_f, in_f, ln: bool, as_buffer: bool):
        super().__init__()

        def reg(mod, name, t):
            if as_buffer:
                mod.register_buffer(name, t)
            else:
                mod.register_parameter(name, nn.Parameter(t))

        reg(self, "weight", torch.zeros(n, out_f, in_f))
        reg(self, "bias", torch.zeros(n, out_f))
        if ln:
            self.ln = nn.Module()
            reg(self.ln, "weight", torch.ones(n, out_f))
            reg(self.ln, "bias", torch.zeros(n, out_f))


class StackedMLPParams(nn.Module):
    """Parameters of `n` identically shaped 3-layer mlps stacked on dim 0, with the
    state-dict keys tensordict gives the reference's Ensemble ("<i>.weight",
    "<i>.ln.weight", ...; reference layers.py:8-33, 167-199)."""

    def __init__(self, n, in_dim, hidden, out_dim, as_buffer=False):
        super().__init__()
        dims = [in_dim, hidden, hidden, out_dim]
        for i in range(3):
            self.add_module(str(i), _StackedLayer(n, dims[i + 1], dims[i], ln=(i < 2), as_buffer=as_buffer))
        self.n = n

    def layer(self, i):
        return getattr(self, str(i))


class QEnsemble(nn.Module):
    """Host-side Q ensemble (reference layers.Ensemble): batched matmuls over stacked params."""

    def __init__(self, n, in_dim, hidden, out_dim):
        super().__init__()
        self.params = StackedMLPParams(n, in_dim, hidden, out_dim)
        self._n = n

    def __len__(self):
        return self._n

    @staticmethod
    def apply_params(p: StackedMLPParams, x):
        h = x.unsqueeze(0).expand(p.n, *x.shape)
        for i in (0, 1):
            l = p.layer(i)
            h = torch.baddbmm(l.bias.unsqueeze(1), h, l.weight.transpose(1, 2))
            h = F.layer_norm(h, (h.shape[-1],), None, None, 1e-5) * l.ln.weight.unsqueeze(1) + l.ln.bias.unsqueeze(1)
            h = F.mish(h)
        l = p.layer(2)
        return torch.baddbmm(l.bias.unsqueeze(1), h, l.weight.transpose(1, 2))

    def forward(self, x):
        return self.apply_params(self.params, x)
