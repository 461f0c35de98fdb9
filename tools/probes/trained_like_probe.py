"""Where does the error on trained-like weights (synth.trained_like) come from?  estimate_value of the c2 model on both kernel
families and both arithmetics against an fp64 evaluation of the same network, torch's fp32 as the yardstick; optionally with one
ingredient of the recipe at a time (PROBE_PARTS=gain,bias,outliers)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tdmpc2_amd import synth  # noqa: E402
from tdmpc2_amd.config import named_config  # noqa: E402
from tests.test_gpu_adversarial import _value_errors  # noqa: E402


def recipe(sd, parts, seed=0):
    full = synth.trained_like(sd, seed)
    out = {}
    for k, v in sd.items():
        use = (k.endswith("ln.weight") and "gain" in parts) or (k.endswith("ln.bias") and "bias" in parts) or \
              (k.endswith(".weight") and v.ndim >= 2 and not k.endswith("ln.weight") and "outliers" in parts)
        out[k] = full[k] if use else v
    return out


name = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = named_config(name, iterations=4) if name == "c2" else named_config(name)
if cfg.multitask:
    cfg.action_dims = [cfg.action_dim - (i % 3) for i in range(len(cfg.tasks))]
    cfg.episode_lengths = [500 if i % 2 == 0 else 100 for i in range(len(cfg.tasks))]
hs = float(os.environ.get("PROBE_HEAD_STD", "0.015"))
base = synth.make_state_dict(cfg, seed=0, head_std=hs)
for parts in (os.environ.get("PROBE_PARTS", "gain,bias,outliers|gain|bias|outliers|none").split("|")):
    sd = recipe(base, parts.split(","))
    for path, prec in ((1, 1), (1, 2), (2, 1), (2, 2)):
        if path == 1 and cfg.latent_dim != 512:
            continue
        hip, ref = _value_errors(cfg, sd, path, prec, E=2)
        print(f"[{name} {parts:22s}] {'fused' if path == 1 else 'layered':8s} {'fp32 ' if prec == 1 else 'split'}: |HIP - fp64| {hip:.3e}   |torch fp32 - fp64| {ref:.3e}", flush=True)
