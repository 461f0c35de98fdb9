"""Latency of dist.sharded_plan with two PROCESSES on one MI355X over gloo (RCCL refuses two ranks on one device): the round-4
protocol (a broadcast + all_reduce + .item() to agree on the Philox stream, a synchronize + all_reduce + .item() for the
verdict, beside the per-iteration value gathers) against the round-5 one (both folded into the value gathers: one host look
per plan).  usage: python tools/probes/shard_latency.py <case> [plans] [old dist.py]     (prints ms per plan, rank 0)"""
import importlib.util
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)


def worker(rank, world, port, name, plans, old_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cases
    from oracle import planner_oracle as po
    from tdmpc2_amd import dist as new_dist
    from tdmpc2_amd.native import NativePlanner
    from tests.gpu_common import plan_inputs

    mods = {"round 5 (folded)": new_dist}
    if old_path and os.path.exists(old_path):
        spec = importlib.util.spec_from_file_location("dist_r4", old_path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods = {"round 4 (separate agreement + verdict collectives)": m, **mods}
    c = cases.build_case(name)
    cfg = c["cfg"]
    model = po.OracleModel(cfg, {k: torch.as_tensor(v) for k, v in c["sd"].items()})
    planner = NativePlanner(cfg, c["iterations"], torch.device("cuda", 0), max_envs=max(c["n_envs"], 2))
    planner.bind_state_dict(model.sd)
    if not hasattr(planner, "tuned_cluster"):
        planner.tuned_cluster, planner.tuned_fuse_ln = 2, 1  # (attributes the round-4 dist.py reads)
    inp = plan_inputs(c, model)
    kw = dict(task_emb=inp["task_emb"], act_mask=inp["act_mask"])
    for rep in range(2):
        for label, m in mods.items():
            pm = inp["prev_mean"].clone()
            for i in range(3):
                m.sharded_plan(planner, inp["z0"], inp["disc_pow"], pm, inp["t0"], tape=None, seed=11 + i, **kw)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            retries = 0
            for i in range(plans):
                m.sharded_plan(planner, inp["z0"], inp["disc_pow"], pm, inp["t0"], tape=None, seed=100 + i, **kw)
                retries += int(planner.last_shard_retries)
            torch.cuda.synchronize()
            dist.barrier()
            ms = (time.perf_counter() - t0) / plans * 1e3
            if rank == 0:
                print(f"{name} E={c['n_envs']} I={c['iterations']} 2 processes on one GPU (gloo): {label}: {ms:.3f} ms per sharded plan "
                      f"({retries} re-plans in {plans})", flush=True)
    planner.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "c1"
    plans = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    old = sys.argv[3] if len(sys.argv) > 3 else ""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(2, port, name, plans, old), nprocs=2, join=True)
