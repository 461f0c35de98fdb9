"""One process, one planner, the tile order of the fused launches switched between calls (TDMPC2_GEMM_XCD_COLS is read per
call): c4 (317M, E = 8 by default) plans/s with a row block on 8 XCDs (row-major, the default for 16 column blocks), on 2
and on 4; the action digests must agree.   python tools/probes/xcd_cols_probe.py [config] [envs] [plans per setting]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
cfg = bench.named_config(name)
I = cfg.iterations + 2 * int(cfg.action_dim >= 20)
cfg, planner, x = bench._planner_for(name, E, I, dev)
prev = torch.zeros(E, cfg.horizon, cfg.action_dim, device=dev)
warm = torch.zeros(E, dtype=torch.uint8, device=dev)
out = torch.empty(E, cfg.action_dim, device=dev)


def run(setting):
    if setting:
        os.environ["TDMPC2_GEMM_XCD_COLS"] = setting
    else:
        os.environ.pop("TDMPC2_GEMM_XCD_COLS", None)
    planner.plan(x["z0"], x["disc"], prev.clone(), warm, task_emb=x["emb"], act_mask=x["mask"], seed=1, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(K):
        planner.plan(x["z0"], x["disc"], prev.clone(), warm, task_emb=x["emb"], act_mask=x["mask"], seed=10 + i, out=out)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    return E * K / el, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]


for rep in range(2):
    for s in ("", "2", "4"):
        v, h = run(s)
        print(f"{name} E={E} XCD_COLS={s or '-'}: {v:8.2f} plans/s  sha {h}  faults {planner.take_fault()}", flush=True)
