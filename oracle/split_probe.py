"""CPU emulation of candidate MFMA arithmetic modes on the oracle network.  TEST INFRASTRUCTURE ONLY.

    python -m oracle.split_probe c1 c2

Each fp32 product of every nn.Linear in `_estimate_value` is replaced by an emulation of a split-precision
matrix-pipe scheme (operands rounded to f16 / bf16 pieces exactly as the HIP kernels do, partial products
accumulated in fp32 by torch's sgemm) and the resulting trajectory values are compared with an fp64
evaluation of the same network.  This is how the f16x2-split mode of `fused_kernels.cuh` was chosen:

    fp32        plain torch fp32                                   error vs fp64 ~ 4e-6 (c1)
    f16x2_3     x = hi + lo in f16, 3 products (hh, hl, lh)        ~ 4e-6   <- same class as fp32: CHOSEN
    bf16x2_3    the same with bf16 pieces (8 + 8 bits)             ~ 5e-5 ... 1e-4: rejected
    bf16x3_6    three bf16 pieces, 6 products                      ~ 2e-6 (twice the MFMA work of f16x2_3)
    f16x2_2a / f16x2_2w   two products (weights resp. activations rounded to one f16 piece)   2e-3 ... 4e-3
    f16_1       one product (plain f16 operands, fp32 accumulate)  3e-3 ... 4e-3: 500 x the tolerance class
"""
from __future__ import annotations

import sys

import torch
import torch.nn.functional as F

from oracle import cases
from oracle import planner_oracle as po

MODES = ["fp32", "f16x2_3", "f16x2_4", "bf16x3_6", "bf16x2_3", "f16x2_2a", "f16x2_2w", "f16_1"]
_orig_linear = F.linear
_mode = {"m": "fp32"}


def _split16(x):
    h = x.half().float()
    return h, (x - h).half().float()


def _linear(x, w, b=None):
    m = _mode["m"]
    if m == "fp32" or x.dtype != torch.float32:
        return _orig_linear(x, w, b)
    if m in ("f16x2_2a", "f16x2_2w", "f16_1"):  # cheaper schemes, measured to show why three products are needed
        sw = 2.0 ** (13 - torch.floor(torch.log2(w.abs().max())).item())
        wh, wl = _split16(w * sw)
        xh, xl = _split16(x * 32.0)
        acc = _orig_linear(xh, wh)
        if m == "f16x2_2a":
            acc = acc + _orig_linear(xl, wh)
        elif m == "f16x2_2w":
            acc = acc + _orig_linear(xh, wl)
        out = acc / (sw * 32.0)
        return out + b if b is not None else out
    if m in ("f16x2_3", "f16x2_4"):
        # the kernels' scaling: weights by 2^kw with max|W| 2^kw in [2^13, 2^14), activations by 2^5
        sw = 2.0 ** (13 - torch.floor(torch.log2(w.abs().max())).item())
        sx = 32.0
        wh, wl = _split16(w * sw)
        xh, xl = _split16(x * sx)
        acc = _orig_linear(xh, wh) + (_orig_linear(xh, wl) + _orig_linear(xl, wh))
        if m == "f16x2_4":
            acc = acc + _orig_linear(xl, wl)
        out = acc / (sw * sx)
        return out + b if b is not None else out

    def pieces(t, n):
        out, r = [], t
        for _ in range(n):
            p = r.bfloat16().float()
            out.append(p)
            r = r - p
        return out

    if m == "bf16x3_6":
        xa, xb, xc = pieces(x, 3)
        wa, wb, wc = pieces(w, 3)
        acc = _orig_linear(xa, wa) + (_orig_linear(xa, wb) + _orig_linear(xb, wa)) + \
            (_orig_linear(xa, wc) + _orig_linear(xc, wa) + _orig_linear(xb, wb))
    elif m == "bf16x2_3":
        xa, xb = pieces(x, 2)
        wa, wb = pieces(w, 2)
        acc = _orig_linear(xa, wa) + (_orig_linear(xa, wb) + _orig_linear(xb, wa))
    else:
        raise KeyError(m)
    return acc + b if b is not None else acc


def _ensemble(sd, prefix, x):
    """planner_oracle.ensemble_forward with per-head F.linear (so the emulation applies)."""
    outs = []
    for q in range(sd[f"{prefix}.0.weight"].shape[0]):
        h = x
        for i in (0, 1):
            h = po.normed_linear(h, sd[f"{prefix}.{i}.weight"][q], sd[f"{prefix}.{i}.bias"][q],
                                 sd[f"{prefix}.{i}.ln.weight"][q], sd[f"{prefix}.{i}.ln.bias"][q], F.mish)
        outs.append(F.linear(h, sd[f"{prefix}.2.weight"][q], sd[f"{prefix}.2.bias"][q]))
    return torch.stack(outs)


def value_errors(name: str, modes=MODES):
    """{mode: max relative error of _estimate_value vs fp64} for env 0 of a golden case."""
    c = cases.build_case(name)
    cfg = c["cfg"]
    sd = {k: torch.as_tensor(v) for k, v in c["sd"].items()}
    model, model64 = po.OracleModel(cfg, sd), po.OracleModel(cfg, sd, dtype=torch.float64)
    H, N, A = cfg.horizon, cfg.num_samples, cfg.action_dim
    g = torch.Generator().manual_seed(7)
    actions = torch.rand(H, N, A, generator=g) * 2 - 1
    eps = torch.randn(N, A, generator=g)
    qidx = torch.tensor([0, 2])
    z = torch.as_tensor(c["z0"][0:1]).repeat(N, 1)
    task = None if c["tasks"] is None else c["tasks"][0]
    saved = (F.linear, po.ensemble_forward)
    F.linear, po.ensemble_forward = _linear, _ensemble
    try:
        with torch.no_grad():
            _mode["m"] = "fp32"
            v64 = po.estimate_value(model64, z.double(), actions.double(), task, c["discounts"][0], eps.double(), qidx).squeeze(1)
            scale = v64.abs().clamp_min(1.0)
            out = {}
            for m in modes:
                _mode["m"] = m
                v = po.estimate_value(model, z, actions, task, c["discounts"][0], eps, qidx).squeeze(1).double()
                out[m] = float(((v - v64).abs() / scale).max())
    finally:
        F.linear, po.ensemble_forward = saved
        _mode["m"] = "fp32"
    return out


if __name__ == "__main__":
    for name in sys.argv[1:] or ["c1"]:
        errs = value_errors(name)
        print(name, {k: f"{v:.3e}" for k, v in errs.items()})
