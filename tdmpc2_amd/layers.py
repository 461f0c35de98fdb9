"""Host-side (PyTorch) building blocks of the world model.

These exist for what stays on the host side of the boundary: the pixel
observation encoder (state observations are also encoded inside the library),
checkpoint I/O in the reference's key layout, and the non-planning
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


class ShiftAug(nn.Module):
    """Random +-`pad` pixel shift (reference layers.py:36-59: replicate-pad, then bilinear resampling on a grid shifted
    by an integer number of pixels per image).  The integer shift comes from `torch.randint` exactly as in the reference,
    so a seeded run draws the same shifts."""

    def __init__(self, pad: int = 3):
        super().__init__()
        self.pad = pad

    def forward(self, x):
        x = x.float()
        n, _, h, w = x.shape
        if h != w:
            raise ValueError(f"ShiftAug expects square images, got {h}x{w}")
        p = self.pad
        x = F.pad(x, (p, p, p, p), mode="replicate")
        full = h + 2 * p
        eps = 1.0 / full
        lin = torch.linspace(-1.0 + eps, 1.0 - eps, full, device=x.device, dtype=x.dtype)[:h]
        gx = lin.view(1, h, 1).expand(h, h, 1)          # x coordinate varies along the width
        grid = torch.cat([gx, gx.transpose(0, 1)], dim=2).unsqueeze(0).expand(n, h, h, 2)
        shift = torch.randint(0, 2 * p + 1, size=(n, 1, 1, 2), device=x.device, dtype=x.dtype) * (2.0 / full)
        return F.grid_sample(x, grid + shift, padding_mode="zeros", align_corners=False)


class PixelPreprocess(nn.Module):
    """uint8-range pixels -> [-0.5, 0.5] (reference layers.py:62-71)."""

    def forward(self, x):
        return x.div(255.0).sub(0.5)


def conv(in_shape, num_channels: int, act=None):
    """Pixel encoder with the reference's module indices, hence its checkpoint keys `_encoder.rgb.{2,4,6,8}.*`
    (reference layers.py:136-150): ShiftAug, PixelPreprocess, four Conv2d (7/2, 5/2, 3/2, 3/1) with ReLU between,
    Flatten, then the optional activation (SimNorm).  64 x 64 inputs shrink 64 -> 29 -> 13 -> 6 -> 4, so Flatten yields
    16 * num_channels features (512 for the default 32 channels = the 5M model's latent_dim).  Host-side PyTorch-ROCm
    (MIOpen convolutions): the planner is handed the latent (`tdmpc2_plan_run`)."""
    if in_shape[-1] != 64:
        raise ValueError(f"the pixel encoder is laid out for 64x64 observations (got {tuple(in_shape)})")
    mods = [ShiftAug(), PixelPreprocess(),
            nn.Conv2d(in_shape[0], num_channels, 7, stride=2), nn.ReLU(inplace=False),
            nn.Conv2d(num_channels, num_channels, 5, stride=2), nn.ReLU(inplace=False),
            nn.Conv2d(num_channels, num_channels, 3, stride=2), nn.ReLU(inplace=False),
            nn.Conv2d(num_channels, num_channels, 3, stride=1), nn.Flatten()]
    if act is not None:
        mods.append(act)
    return nn.Sequential(*mods)


def encoders(cfg):
    """One encoder per observation key (reference layers.py:153-164): 'state' -> NormedLinear stack with a SimNorm output,
    'rgb' -> `conv`.  A fresh dict per call (the reference's mutable default argument shares encoders between models)."""
    out = {}
    for k, shape in cfg.obs_shape.items():
        if k == "state":
            out[k] = mlp(shape[0] + cfg.task_dim, max(cfg.num_enc_layers - 1, 1) * [cfg.enc_dim], cfg.latent_dim,
                         act=SimNorm(cfg.simnorm_dim))
        elif k == "rgb":
            out[k] = conv(shape, cfg.num_channels, act=SimNorm(cfg.simnorm_dim))
        else:
            raise NotImplementedError(f"Encoder for observation type {k} not implemented.")
    return nn.ModuleDict(out)


state_encoder = encoders  # name used by round-1 callers


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
