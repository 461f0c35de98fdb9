"""c3 with trained-like weights, 16 seeded plans: iteration-0 values of the sampled rows against an fp64 evaluation, per kernel route
(which route carries the 8e-4 of r6q?)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cases  # noqa: E402
from oracle import planner_oracle as po  # noqa: E402
from tdmpc2_amd import synth  # noqa: E402
from tdmpc2_amd.config import named_config  # noqa: E402
from tdmpc2_amd.native import NativePlanner  # noqa: E402
from tests.gpu_common import dev  # noqa: E402
from tests.test_gpu_planner import _run_native  # noqa: E402

E = int(os.environ.get("PROBE_E", "16"))
cfg = named_config("c3")
c = cases.build_custom(cfg, E, head_std=0.02, name="c3")
c["sd"] = synth.trained_like(c["sd"], seed=0)
c["iterations"] = 1
for k in ("sample_eps", "pi_eps", "qidx"):
    c["tape"][k] = np.ascontiguousarray(c["tape"][k][:, :1])
tsd = {k: torch.as_tensor(v) for k, v in c["sd"].items()}
m32, m64 = po.OracleModel(cfg, tsd), po.OracleModel(cfg, tsd, dtype=torch.float64)
P = cfg.num_pi_trajs
v32, v64 = [], []
for e in range(E):
    kw = dict(tape=po.env_tape(c["tape"], e), t0=bool(c["t0"][e]), eval_mode=False, task=c["tasks"][e], iterations=1)
    _, _, st = po.plan(m32, z0=torch.as_tensor(c["z0"][e:e + 1]), prev_mean=torch.as_tensor(c["prev_mean"][e]), discount=c["discounts"][e], **kw)
    _, _, s64 = po.plan(m64, z0=torch.as_tensor(c["z0"][e:e + 1]).double(), prev_mean=torch.as_tensor(c["prev_mean"][e]).double(),
                        discount=c["discounts"][e].double(), **kw)
    v32.append(st["value"][0].numpy().astype(np.float64)[P:])
    v64.append(s64["value"][0].numpy()[P:])
v32, v64 = np.stack(v32), np.stack(v64)
rel = lambda a: np.abs(a - v64) / np.maximum(1.0, np.abs(v64))  # noqa: E731
r = rel(v32)
print(f"torch fp32 vs fp64: max {r.max():.2e} at env {r.max(1).argmax()}, |v| there {abs(v64.flat[r.argmax()]):.1f}; 99.9 % {np.quantile(r, 0.999):.2e}", flush=True)
for label, env, prec in (("default", {}, 2), ("FUSE_LN=0", {"TDMPC2_FUSE_LN": "0"}, 2), ("Z0_SHARED_OFF", {"TDMPC2_Z0_SHARED_OFF": "1"}, 2),
                         ("no wide tile", {"TDMPC2_GEMM_W256_MIN": "-1"}, 2), ("one stream", {"TDMPC2_ONE_STREAM": "1"}, 2), ("exact fp32", {}, 1)):
    os.environ.update(env)
    pl = NativePlanner(cfg, 1, dev(), max_envs=E, precision=prec)
    for k in env:
        del os.environ[k]
    pl.bind_state_dict(m32.sd)
    got = _run_native(c, m32, pl)
    pl.close()
    h = got["value"][:, 0].astype(np.float64)[:, P:]
    rh = rel(h)
    print(f"{label:14s}: HIP vs fp64 max {rh.max():.2e} at env {rh.max(1).argmax()} (|v| {abs(v64.flat[rh.argmax()]):.1f}, torch there {r.flat[rh.argmax()]:.2e}); "
          f"99.9 % {np.quantile(rh, 0.999):.2e}; vs torch fp32 max {(np.abs(h - v32) / np.maximum(1, np.abs(v32))).max():.2e}", flush=True)
