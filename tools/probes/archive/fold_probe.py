"""E = 1 planning: where the elite refit runs (inside the rollout launch or as k_refit) and what it costs.
Run under `rocprofv3 --kernel-trace` (tools/gpu_run7.sh); prints wall ms per plan for both modes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tdmpc2_amd import synth  # noqa: E402
from tdmpc2_amd.config import named_config  # noqa: E402
from tdmpc2_amd.native import NativePlanner  # noqa: E402

cfg = named_config("c2")
dev = torch.device("cuda", 0)
sd = {k: torch.as_tensor(v).to(dev) for k, v in synth.make_state_dict(cfg, seed=0).items()}
pl = NativePlanner(cfg, 6, dev, max_envs=1)
pl.bind_state_dict(sd)
z = torch.as_tensor(synth.make_latents(cfg, 1, seed=1)).to(dev)
disc = torch.tensor([[0.99 ** k for k in range(cfg.horizon + 1)]], dtype=torch.float32, device=dev)
pm = torch.zeros(1, cfg.horizon, cfg.action_dim, device=dev)
t0 = torch.zeros(1, dtype=torch.uint8, device=dev)
out = torch.empty(1, cfg.action_dim, device=dev)
for mode in (1, 0, 1, 0):
    pl.set_fold_refit(mode)
    for i in range(3):
        pl.plan(z, disc, pm, t0, seed=i, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(20):
        pl.plan(z, disc, pm, t0, seed=10 + i, out=out)
    torch.cuda.synchronize()
    print(f"fold={mode}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per plan", flush=True)
