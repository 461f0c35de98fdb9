"""How fast is the oracle port next to the reference's OWN `_plan` (tdmpc2/tdmpc2.py:138-206, run verbatim by oracle/ref_runner.py)?

    python -m oracle.time_port_vs_reference            # build container only (needs /root/reference); writes profiles/port_vs_reference.json

TEST INFRASTRUCTURE ONLY.  bench.py's `cpu_baseline` is the PORT (`kind: "port"`): the reference is Python and may not travel to
the GPU box in any form, so its own file cannot be timed on those host cores.  What can be done is done here: both are timed in
the one place where both exist, on the same inputs and thread count, and the committed record gives `cpu_baseline.port_vs_reference`
a measured value with its provenance instead of a bare constant."""
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import planner_oracle as po, ref_runner  # noqa: E402
from tdmpc2_amd import synth  # noqa: E402
from tdmpc2_amd.config import get_discount, named_config  # noqa: E402


def time_port(cfg, sd, z0, tape, disc, iterations, budget_s):
    model = po.OracleModel(cfg, sd)
    prev = torch.zeros(cfg.horizon, cfg.action_dim)

    def one(t0):
        nonlocal prev
        _, prev, _ = po.plan(model, z0=torch.as_tensor(z0), tape=tape, prev_mean=prev, t0=t0, eval_mode=False, task=None, discount=disc,
                             iterations=iterations)

    with torch.no_grad():
        one(True)
        n, t0 = 0, time.perf_counter()
        while n < 3 or time.perf_counter() - t0 < budget_s:
            one(False)
            n += 1
    return 1e3 * (time.perf_counter() - t0) / n, n


def main():
    threads = len(os.sched_getaffinity(0))
    torch.set_num_threads(threads)
    out = {"threads": threads, "cpu": platform.processor() or platform.machine(), "torch": torch.__version__, "cases": {}}
    for name, iters in (("c1", 6), ("c2", 6)):
        cfg = named_config(name)
        sd = {k: torch.as_tensor(v) for k, v in synth.make_state_dict(cfg, seed=0).items()}
        z0 = synth.make_latents(cfg, 1, seed=1)
        tape = po.env_tape(synth.make_noise_tape(cfg, 1, iters, seed=2), 0)
        disc = get_discount(cfg, cfg.episode_length)
        # interleaved twice, so that neither side gets the warmer machine
        port, ref = [], []
        for _ in range(2):
            port.append(time_port(cfg, sd, z0, tape, disc, iters, 8.0))
            ref.append(ref_runner.time_reference_plan(cfg, sd, z0=z0[0], tape=tape, task=None, discount=disc, iterations=iters, budget_s=8.0))
        pm, rm = min(p[0] for p in port), min(r[0] for r in ref)
        out["cases"][name] = {"iterations": iters, "port_ms_per_plan": round(pm, 1), "reference_ms_per_plan": round(rm, 1),
                              "port_vs_reference": round(pm / rm, 3), "plans_timed": [p[1] for p in port] + [r[1] for r in ref]}
        print(name, out["cases"][name], flush=True)
    path = os.path.join(ROOT, "profiles", "port_vs_reference.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
