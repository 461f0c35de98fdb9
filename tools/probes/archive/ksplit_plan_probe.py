"""Whole plans (recorded tape) with the K-split default against the same plans with every tile whole, iteration by iteration.
usage: python tools/probes/ksplit_plan_probe.py <golden case> [repeats]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests.gpu_common import case_on_gpu, plan_inputs  # noqa: E402

name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
c, model, planner = case_on_gpu(name, 2, 2)
inp = plan_inputs(c, model)
kw = dict(eval_mode=c["eval_mode"], task_emb=inp["task_emb"], act_mask=inp["act_mask"], tape=inp["tape"], debug=True)


def run():
    pm = inp["prev_mean"].clone()
    a, st = planner.plan(inp["z0"], inp["disc_pow"], pm, inp["t0"], **kw)
    torch.cuda.synchronize()
    return a.cpu().numpy(), st["value"].cpu().numpy()


planner.set_ksplit(0)
a0, v0 = run()
a0b, v0b = run()
print(f"{name}: whole tiles repeatable {np.array_equal(v0, v0b)}; faults {planner.fault_info()['faults_total']}")
planner.set_ksplit(int(os.environ.get('PROBE_KSPLIT', '2')))
for r in range(reps):
    a, v = run()
    err = np.abs(v - v0) / np.maximum(1.0, np.abs(v0))
    per_it = [(float(err[:, it].max()), int((err[:, it] > 1e-3).sum())) for it in range(v.shape[1])]
    print(f"  split plan {r}: per iteration (max rel err, rows > 1e-3): {[(f'{m:.1e}', n) for m, n in per_it]}  faults {planner.fault_info()['faults_total']}")
    bad = np.argwhere(err > 1e-3)
    if len(bad):
        rows = sorted(set(int(b[2]) for b in bad if b[1] == bad[0][1] and b[0] == bad[0][0]))
        print(f"     first bad (env, iteration) = {tuple(int(x) for x in bad[0][:2])}; bad rows there: {rows[:12]} ... {rows[-4:]} ({len(rows)} rows)")
planner.close()
