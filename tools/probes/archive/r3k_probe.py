"""r03k hunt (profiles/README.md): after a layered handle has worked (two streams), the first eager plan of a fresh fused handle
on the NULL stream comes back wrong under pytest (tools/gpu_r4o.sh).  The same sequence outside pytest, one variant per
process:   python tools/probes/r3k_probe.py <variant>
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from tdmpc2_amd import synth  # noqa: E402
from tdmpc2_amd.native import PATH_LAYERED, NativePlanner  # noqa: E402
from tests.gpu_common import case_on_gpu, dev, plan_inputs  # noqa: E402
from tests.helpers import load_golden  # noqa: E402

V = sys.argv[1] if len(sys.argv) > 1 else "base"
opts = set(V.split("+"))


def layered_work():
    c, model, ref = case_on_gpu("small", PATH_LAYERED, 2)
    cfg = c["cfg"]
    if "fault" in opts:
        os.environ["TDMPC2_CLUSTER_FAULT"] = "1"
    planner = NativePlanner(cfg, c["iterations"], dev(), max_envs=c["n_envs"], path=PATH_LAYERED, precision=2)
    os.environ.pop("TDMPC2_CLUSTER_FAULT", None)
    planner.bind_state_dict(model.sd)
    if "nowork" not in opts:
        if "plan" in opts:
            inp = plan_inputs(c, model)
            planner.plan(inp["z0"], inp["disc_pow"], inp["prev_mean"].clone(), inp["t0"], eval_mode=c["eval_mode"],
                         task_emb=inp["task_emb"], act_mask=inp["act_mask"], tape=inp["tape"])
        else:
            R = 256
            z = torch.as_tensor(synth.make_latents(cfg, R, seed=9)).to(dev())
            rw, tm = torch.randn(R, device=dev()), torch.zeros(R, device=dev())
            for _ in range(2):
                planner.td_target(z, rw, tm, 0.99, seed=1)
                torch.cuda.synchronize()
                planner.take_fault()
    torch.cuda.synchronize()
    if "noclose" not in opts:
        planner.close()
    return planner


def fused_plan(pre=None):
    c, model, planner = pre if pre else case_on_gpu("c1")
    if "nocluster" in opts:
        planner.set_cluster(0)
    inp = plan_inputs(c, model)
    kw = dict(task_emb=inp["task_emb"], act_mask=inp["act_mask"], tape=inp["tape"])
    want = torch.as_tensor(load_golden("c1")["action"]).to(dev())
    errs = []
    s = torch.cuda.Stream() if "sidestream" in opts else None
    for i in range(3):
        pm = inp["prev_mean"].clone()
        if s is not None:
            torch.cuda.synchronize()
            with torch.cuda.stream(s):
                a = planner.plan(inp["z0"], inp["disc_pow"], pm, inp["t0"], **kw).clone()
        else:
            a = planner.plan(inp["z0"], inp["disc_pow"], pm, inp["t0"], **kw).clone()
        torch.cuda.synchronize()
        errs.append((float((a - want).abs().max()), planner.take_fault()))
    return errs


pre = case_on_gpu("c1") if "c1first" in opts else None
if "nolayered" not in opts:
    keep = layered_work()
if "sleep" in opts:
    torch.cuda.synchronize()
    time.sleep(1.0)
print(f"{V:40s} (max |action - golden|, fault) of three eager c1 plans: {fused_plan(pre)}", flush=True)
