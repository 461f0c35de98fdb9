import time, torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tdmpc2_amd import synth
from tdmpc2_amd.config import named_config, get_discount
from tdmpc2_amd.native import NativePlanner
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = named_config(name); I = 6
sd = {k: torch.as_tensor(v).to(dev) for k, v in synth.make_state_dict(cfg, seed=0).items()}
pl = NativePlanner(cfg, I, dev, max_envs=1); pl.bind_state_dict(sd)
emb = mask = None
if cfg.multitask:
    w = sd["_task_emb.weight"][:1]; n = w.norm(dim=1, keepdim=True)
    emb = torch.where(n > 1.0, w / (n + 1e-7), w).contiguous(); mask = sd["_action_masks"][:1].contiguous()
z = torch.as_tensor(synth.make_latents(cfg, 1, seed=1)).to(dev)
g = get_discount(cfg, 500); disc = torch.tensor([[g**k for k in range(cfg.horizon+1)]], device=dev)
prev = torch.zeros(1, cfg.horizon, cfg.action_dim, device=dev); t0 = torch.zeros(1, dtype=torch.uint8, device=dev)
out = torch.empty(1, cfg.action_dim, device=dev)
for i in range(3): pl.plan(z, disc, prev, t0, seed=i, out=out, task_emb=emb, act_mask=mask)
torch.cuda.synchronize()
t=time.perf_counter()
for i in range(20): pl.plan(z, disc, prev, t0, seed=10+i, out=out, task_emb=emb, act_mask=mask)
torch.cuda.synchronize(); print("eager launches: %.3f ms/plan" % ((time.perf_counter()-t)/20*1e3))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    pl.plan(z, disc, prev, t0, seed=1, out=out, task_emb=emb, act_mask=mask)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=s):
    pl.plan(z, disc, prev, t0, seed=1, out=out, task_emb=emb, act_mask=mask)
for i in range(3): gr.replay()
torch.cuda.synchronize()
t=time.perf_counter()
for i in range(20): gr.replay()
torch.cuda.synchronize(); print("graph replay: %.3f ms/plan" % ((time.perf_counter()-t)/20*1e3))
print(out)
