#!/bin/bash
# episodic models on the cluster path: planner / edge / boundary tests, single-plan latency of an episodic 5M model
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_planner.py tests/test_gpu_edge.py tests/test_gpu_boundary.py tests/test_gpu_dist.py -q -m gpu --timeout 600 > gpurun_out/r03e_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03e_pytest.log; tail -15 gpurun_out/r03e_pytest.log
python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from tdmpc2_amd import synth
from tdmpc2_amd.config import named_config
from tdmpc2_amd.native import NativePlanner
dev = torch.device("cuda", 0)
cfg = named_config("c2"); cfg.episodic = True
sd = {k: torch.as_tensor(v).to(dev) for k, v in synth.make_state_dict(cfg, seed=0).items()}
pl = NativePlanner(cfg, 6, dev, max_envs=1); pl.bind_state_dict(sd)
z = torch.as_tensor(synth.make_latents(cfg, 1, seed=1)).to(dev)
disc = torch.tensor([[0.99 ** k for k in range(cfg.horizon + 1)]], dtype=torch.float32, device=dev)
pm = torch.zeros(1, cfg.horizon, cfg.action_dim, device=dev); t0 = torch.zeros(1, dtype=torch.uint8, device=dev)
out = torch.empty(1, cfg.action_dim, device=dev)
for mode in (1, 0, 1, 0):
    pl.set_cluster(mode)
    for i in range(3): pl.plan(z, disc, pm, t0, seed=i, out=out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(20): pl.plan(z, disc, pm, t0, seed=10 + i, out=out)
    torch.cuda.synchronize(); print(f"c2 episodic E=1 cluster={mode}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per plan", flush=True)
PY
