"""Does a stage of the layered family lose to tile quantisation?  G planners of E / G plans each on G torch streams against one
planner of E plans (c3: 420 tiles of 256 x 256 on 256 CUs = 1.64 rounds per GEMM).  usage: python tools/probes/split_batch_probe.py c3 30 2 3 5"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from tdmpc2_amd import synth
from tdmpc2_amd.config import named_config
from tdmpc2_amd.native import NativePlanner
from bench import disc_pow_rows

name, E = sys.argv[1], int(sys.argv[2])
groups = [int(g) for g in sys.argv[3:]] or [2]
dev = torch.device("cuda", 0)
cfg = named_config(name)
I = cfg.iterations + 2 * int(cfg.action_dim >= 20)
sd_np = synth.make_state_dict(cfg, seed=0)
sd = {k: torch.as_tensor(v).to(dev) for k, v in sd_np.items() if not k.startswith("_encoder.")}


def inputs(e0, n):
    tasks = (torch.arange(n) + e0) % len(cfg.tasks)
    w = torch.as_tensor(sd_np["_task_emb.weight"])[tasks]
    nn = w.norm(dim=1, keepdim=True)
    emb = torch.where(nn > 1.0, w / (nn + 1e-7), w).to(dev).contiguous()
    mask = torch.as_tensor(sd_np["_action_masks"])[tasks].to(dev).contiguous()
    z0 = torch.as_tensor(synth.make_latents(cfg, E, seed=2000))[e0:e0 + n].to(dev).contiguous()
    return dict(z0=z0, disc=disc_pow_rows(cfg, n, dev), prev=torch.zeros(n, cfg.horizon, cfg.action_dim, device=dev),
                warm=torch.zeros(n, dtype=torch.uint8, device=dev), out=torch.empty(n, cfg.action_dim, device=dev), emb=emb, mask=mask)


def run(G, steps=6):
    per = E // G
    ps, ins, streams = [], [], []
    for g in range(G):
        p = NativePlanner(cfg, I, dev, max_envs=per)
        p.bind_state_dict(sd)
        if os.environ.get("PROBE_UNFUSED"):  # GEMM + row kernel instead of the NormedLinear epilogue with its cross-workgroup wait
            p.set_fuse_ln(0)
        ps.append(p)
        ins.append(inputs(g * per, per))
        streams.append(torch.cuda.Stream(device=dev))
    def step(i):
        for g in range(G):
            with torch.cuda.stream(streams[g]):
                x = ins[g]
                ps[g].plan(x["z0"], x["disc"], x["prev"], x["warm"], task_emb=x["emb"], act_mask=x["mask"], seed=100 + i, out=x["out"])
    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        step(10 + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    faults = sum(p.take_fault() for p in ps)
    for p in ps:
        p.close()
    print(f"{name} E={E} groups={G} ({per} plans each): {G * per * steps / el:.1f} plans/s, {1e3 * el / steps:.2f} ms per step, faults {faults}", flush=True)


run(1)
for G in groups:
    run(G)
run(1)
