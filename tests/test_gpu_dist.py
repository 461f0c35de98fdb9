"""-m gpu: the multi-GPU host path on the hardware at hand (one GPU): `torch.distributed` with backend "nccl" (= RCCL)
at world size 1, the HIP planner as the compute leg -- weight broadcast, env sharding + action gather, and one plan
sharded over "ranks" through the shard_* entry points against tdmpc2_plan_run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from tests.gpu_common import case_on_gpu, dev, plan_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev())
    yield
    dist.destroy_process_group()


def test_env_sharded_planning_over_rccl(rccl_group):
    """dist.py's env sharding with the HIP planner: broadcast the weights (bucketed), plan the local envs, gather actions."""
    from tdmpc2_amd.dist import broadcast_state_dict, gather_actions, shard_range
    from tdmpc2_amd.native import NativePlanner

    c, model, ref = case_on_gpu("c1", 1, 2)
    sd = {k: v.to(dev()) for k, v in model.sd.items()}
    broadcast_state_dict(sd, src=0, bucket_bytes=8 << 20)  # 20 MB of weights -> several buckets over RCCL
    assert all(torch.equal(sd[k].cpu(), model.sd[k]) for k in sd)
    a0, a1 = shard_range(c["n_envs"], dist.get_world_size(), dist.get_rank())
    planner = NativePlanner(c["cfg"], c["iterations"], dev(), max_envs=a1 - a0)
    planner.bind_state_dict(sd)
    inp = plan_inputs(c, model)
    local = planner.plan(inp["z0"][a0:a1].contiguous(), inp["disc_pow"][a0:a1].contiguous(), inp["prev_mean"][a0:a1].clone(),
                         inp["t0"][a0:a1].contiguous(), tape={k: v[a0:a1].contiguous() for k, v in inp["tape"].items()})
    full = gather_actions(local, c["n_envs"])
    want = ref.plan(inp["z0"], inp["disc_pow"], inp["prev_mean"].clone(), inp["t0"], tape=inp["tape"])
    assert torch.equal(full, want)
    planner.close()


@pytest.mark.parametrize("name,path,prec", [("c1", 1, 2), ("c1", 1, 1), ("mt5", 1, 2), ("small_ep", 2, 2), ("small_mt", 2, 1)])
def test_sharded_plan_entry_points_reproduce_run(name, path, prec, rccl_group):
    """shard_begin / shard_values / shard_refit with one rank and the full row range compute what tdmpc2_plan_run computes
    (same tape): values, elite sets, mean / std of every iteration and the final action."""
    from tdmpc2_amd.dist import sharded_plan

    c, model, planner = case_on_gpu(name, path, prec)
    inp = plan_inputs(c, model)
    kw = dict(task_emb=inp["task_emb"], act_mask=inp["act_mask"], tape=inp["tape"])
    pm_a = inp["prev_mean"].clone()
    if path == 1:  # the shard entry points run one workgroup per row tile: compare with the same kernel (not the cluster path)
        planner.set_cluster(0)
    try:
        a, st = planner.plan(inp["z0"], inp["disc_pow"], pm_a, inp["t0"], eval_mode=c["eval_mode"], debug=True, **kw)
    finally:
        if path == 1:
            planner.set_cluster(2)
    pm_b = inp["prev_mean"].clone()
    stages = planner.debug_buffers(c["n_envs"])
    b = sharded_plan(planner, inp["z0"], inp["disc_pow"], pm_b, inp["t0"], eval_mode=c["eval_mode"], stages=stages, **kw)
    torch.cuda.synchronize()
    assert torch.equal(st["elite_idx"], stages["elite_idx"])
    for k in ("value", "mean", "std"):
        assert torch.allclose(st[k], stages[k], rtol=1e-6, atol=1e-6), k
    assert torch.allclose(a, b, atol=1e-6) and torch.allclose(pm_a, pm_b, atol=1e-6)


@pytest.mark.parametrize("name,path", [("c1", 1), ("small", 2)])
def test_row_ranges_compose(name, path, rccl_group):
    """Two half ranges written one after the other give the values of one full-range call bit for bit (what two ranks
    would all-gather)."""
    c, model, planner = case_on_gpu(name, path, 2)
    cfg = c["cfg"]
    inp = plan_inputs(c, model)
    E, N = c["n_envs"], cfg.num_samples
    if N // 2 % planner.shard_granularity:
        pytest.skip("num_samples does not split in two aligned halves for this family")
    planner.shard_begin(inp["z0"], inp["prev_mean"].clone(), inp["t0"], tape=inp["tape"])
    full = torch.zeros(E, N, device=dev())
    planner.shard_values(0, 0, N, inp["z0"], inp["disc_pow"], full)
    halves = torch.zeros(E, N, device=dev())
    planner.shard_values(0, 0, N // 2, inp["z0"], inp["disc_pow"], halves)
    planner.shard_values(0, N // 2, N, inp["z0"], inp["disc_pow"], halves)
    torch.cuda.synchronize()
    assert torch.equal(full, halves)
