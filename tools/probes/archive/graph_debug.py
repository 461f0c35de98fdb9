"""Debug: eager vs hipGraph replay of a plan after other handles have been created / destroyed (allocator reuse)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import cases  # noqa: E402
from oracle import planner_oracle as po  # noqa: E402
from tdmpc2_amd.native import NativePlanner  # noqa: E402
from tests.gpu_common import dev, plan_inputs  # noqa: E402
from tests.helpers import load_golden  # noqa: E402

pre = sys.argv[1] if len(sys.argv) > 1 else "none"
c = cases.build_case("c1")
model = po.OracleModel(c["cfg"], {k: torch.as_tensor(v) for k, v in c["sd"].items()})
g = load_golden("c1")
if pre in ("fused", "both"):
    p0 = NativePlanner(c["cfg"], c["iterations"], dev(), max_envs=2, path=1)
    p0.bind_state_dict(model.sd)
    i0 = plan_inputs(c, model)
    p0.plan(i0["z0"], i0["disc_pow"], i0["prev_mean"].clone(), i0["t0"], tape=i0["tape"])
    torch.cuda.synchronize()
    p0.close()
if pre in ("layered", "both"):
    cs = cases.build_case("small")
    ms = po.OracleModel(cs["cfg"], {k: torch.as_tensor(v) for k, v in cs["sd"].items()})
    p1 = NativePlanner(cs["cfg"], cs["iterations"], dev(), max_envs=3, path=2)
    p1.bind_state_dict(ms.sd)
    i1 = plan_inputs(cs, ms)
    if os.environ.get("DBG_EV"):
        cfgs = cs["cfg"]
        gg = torch.Generator().manual_seed(7)
        acts = (torch.rand(3, cfgs.horizon, cfgs.num_samples, cfgs.action_dim, generator=gg) * 2 - 1).to(dev()).contiguous()
        eps = torch.randn(3, cfgs.num_samples, cfgs.action_dim, generator=gg).to(dev()).contiguous()
        qi = torch.tensor([[0, 2], [1, 0], [2, 1]], dtype=torch.int32).to(dev())
        v = p1.estimate_value(i1["z0"], i1["disc_pow"], acts, eps, qi).cpu()
        print("layered estimate_value finite", bool(torch.isfinite(v).all()))
    else:
        p1.plan(i1["z0"], i1["disc_pow"], i1["prev_mean"].clone(), i1["t0"], tape=i1["tape"])
        torch.cuda.synchronize()
    p1.close()
planner = NativePlanner(c["cfg"], c["iterations"], dev(), max_envs=2)
planner.bind_state_dict(model.sd)
for mode in (sys.argv[2:] or ["default"]):
    if mode == "nocluster":
        planner.set_cluster(0)
    if mode == "nofold":
        planner.set_fold_refit(0)
    inp = plan_inputs(c, model)
    kw = dict(tape=inp["tape"])
    ref = torch.as_tensor(g["action"]).to(dev())
    for i in range(2):
        a = planner.plan(inp["z0"], inp["disc_pow"], inp["prev_mean"].clone(), inp["t0"], **kw)
        print(mode, "eager", i, float((a - ref).abs().max()))
    pm_static, out = inp["prev_mean"].clone(), torch.empty_like(a)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        planner.plan(inp["z0"], inp["disc_pow"], pm_static.clone(), inp["t0"], out=out, **kw)
    torch.cuda.synchronize()
    print(mode, "side-stream eager", float((out - ref).abs().max()))
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        planner.plan(inp["z0"], inp["disc_pow"], pm_static, inp["t0"], out=out, **kw)
    for i in range(3):
        pm_static.copy_(inp["prev_mean"])
        out.zero_()
        gr.replay()
        torch.cuda.synchronize()
        print(mode, "replay", i, float((out - ref).abs().max()), "faults", planner.take_fault())
