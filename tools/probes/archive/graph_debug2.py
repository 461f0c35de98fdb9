"""Debug: call the pytest functions in sequence, print eager / replay vs golden."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_gpu_adversarial as adv  # noqa: E402
from tests.gpu_common import case_on_gpu, plan_inputs  # noqa: E402
from tests.helpers import load_golden  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "small"
if which != "none":
    adv.test_split_arithmetic_on_heavy_tailed_weights(which, 1 if which in ("c1", "mt5") else 2)
c, model, planner = case_on_gpu("c1")
g = torch.as_tensor(load_golden("c1")["action"]).cuda()
inp = plan_inputs(c, model)
kw = dict(task_emb=inp["task_emb"], act_mask=inp["act_mask"], tape=inp["tape"])
pm_eager = inp["prev_mean"].clone()
a_eager = planner.plan(inp["z0"], inp["disc_pow"], pm_eager, inp["t0"], **kw).clone()
print("eager vs golden", float((a_eager - g).abs().max()))
pm_static, out = inp["prev_mean"].clone(), torch.empty_like(a_eager)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    planner.plan(inp["z0"], inp["disc_pow"], pm_static.clone(), inp["t0"], out=out, **kw)
torch.cuda.synchronize()
print("side eager vs golden", float((out - g).abs().max()))
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=s):
    planner.plan(inp["z0"], inp["disc_pow"], pm_static, inp["t0"], out=out, **kw)
for i in range(3):
    pm_static.copy_(inp["prev_mean"])
    out.zero_()
    gr.replay()
    torch.cuda.synchronize()
    print("replay", i, "vs golden", float((out - g).abs().max()), "pm equal", bool(torch.equal(pm_static, pm_eager)), "faults", planner.take_fault())
a2 = planner.plan(inp["z0"], inp["disc_pow"], inp["prev_mean"].clone(), inp["t0"], **kw)
print("eager again vs golden", float((a2 - g).abs().max()))
