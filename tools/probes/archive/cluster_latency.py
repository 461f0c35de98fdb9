"""Single-plan latency with and without the cluster path (tdmpc2_amd/csrc/cluster_kernels.cuh), c1 and c2, E = 1 and 2.
Prints wall ms per plan (20 plans back to back after 3 warm-up plans)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tdmpc2_amd import synth  # noqa: E402
from tdmpc2_amd.config import named_config  # noqa: E402
from tdmpc2_amd.native import NativePlanner  # noqa: E402

dev = torch.device("cuda", 0)
for name in sys.argv[1:] or ["c2", "c1"]:
    cfg = named_config(name)
    sd = {k: torch.as_tensor(v).to(dev) for k, v in synth.make_state_dict(cfg, seed=0).items()}
    for E in [int(m) for m in os.environ.get("CLUSTER_ENVS", "1,2").split(",")]:
        pl = NativePlanner(cfg, 6, dev, max_envs=E)
        pl.bind_state_dict(sd)
        z = torch.as_tensor(synth.make_latents(cfg, E, seed=1)).to(dev)
        disc = torch.tensor([[0.99 ** k for k in range(cfg.horizon + 1)]] * E, dtype=torch.float32, device=dev)
        pm = torch.zeros(E, cfg.horizon, cfg.action_dim, device=dev)
        t0 = torch.zeros(E, dtype=torch.uint8, device=dev)
        out = torch.empty(E, cfg.action_dim, device=dev)
        for mode in [int(m) for m in os.environ.get("CLUSTER_MODES", "1,0,1,0").split(",")]:
            pl.set_cluster(mode)
            for i in range(3):
                pl.plan(z, disc, pm, t0, seed=i, out=out)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(20):
                pl.plan(z, disc, pm, t0, seed=10 + i, out=out)
            torch.cuda.synchronize()
            print(f"{name} E={E} cluster={mode}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per call", flush=True)
        pl.close()
