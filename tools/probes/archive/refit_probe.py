"""Phase cycle counts of the elite refit (a -DREFIT_TIMING build returns them through the score output):
TDMPC2_PLAN_LIB=build/ablate/lib_rt48.so python tools/probes/refit_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tdmpc2_amd import synth  # noqa: E402
from tdmpc2_amd.config import named_config  # noqa: E402
from tdmpc2_amd.native import NativePlanner  # noqa: E402

cfg = named_config("c2")
dev = torch.device("cuda", 0)
pl = NativePlanner(cfg, 6, dev, max_envs=256)
for E in (1, 256):
    g = torch.Generator(device=dev).manual_seed(0)
    value = torch.randn(E, cfg.num_samples, device=dev, generator=g)
    actions = torch.rand(E, cfg.horizon, cfg.num_samples, cfg.action_dim, device=dev, generator=g) * 2 - 1
    for _ in range(3):
        mean, std, score, idx = pl.refit(value.clone(), actions)
    torch.cuda.synchronize()
    names = ["load+nan_to_num", "top-k", "score sums", "stage elites", "mean/std", "outputs", "final"]
    cyc = score[:, :6].mean(0).tolist()
    print(f"E={E}: " + ", ".join(f"{n} {c:.0f}" for n, c in zip(names, cyc)) + f"  (total {sum(cyc):.0f} cycles)")
    want = torch.topk(value, cfg.num_elites, dim=1).indices
    print("   elite order equals torch.topk:", bool(torch.equal(idx.long(), want)))
