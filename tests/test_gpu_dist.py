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


# ---------------------------------------------------------------- two PROCESSES sharing the one MI355X
# RCCL refuses two ranks on one device, so the ranks talk over gloo (value slices and actions staged through host memory,
# tdmpc2_amd/dist.py) -- everything else is the real thing: two processes, two library handles on the same GPU, the HIP
# planner as the compute leg of dist.sharded_plan and of the env-sharded batch path.
def _two_proc_worker(rank, world, port, out_dir, name, path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import cases
        from oracle import planner_oracle as po
        from tdmpc2_amd.dist import gather_actions, shard_range, sharded_plan
        from tdmpc2_amd.native import NativePlanner

        d = torch.device("cuda", 0)
        c = cases.build_case(name)
        cfg, E = c["cfg"], c["n_envs"]
        model = po.OracleModel(cfg, {k: torch.as_tensor(v) for k, v in c["sd"].items()})
        planner = NativePlanner(cfg, c["iterations"], d, max_envs=max(E, 2), path=path)
        planner.bind_state_dict(model.sd)
        if path == 2:
            planner.set_ksplit(0)  # every GEMM tile whole: the bits of a row do not depend on how many rows share the call
        inp = plan_inputs(c, model)
        kw = dict(task_emb=inp["task_emb"], act_mask=inp["act_mask"])
        # (1) ONE plan, its sample rows split over the two processes; no tape: in-kernel Philox.  Rank 1's handle has planned
        # on its own before (its call counter ran ahead): sharded_plan must re-align the streams.
        if rank == 1:
            planner.plan(inp["z0"], inp["disc_pow"], inp["prev_mean"].clone(), inp["t0"], seed=5, **kw)
        pm = inp["prev_mean"].clone()
        stages = planner.debug_buffers(E)
        a = sharded_plan(planner, inp["z0"], inp["disc_pow"], pm, inp["t0"], eval_mode=c["eval_mode"], tape=None, seed=77,
                         stages=stages, **kw)
        torch.cuda.synchronize()
        torch.save({"action": a.cpu(), "prev_mean": pm.cpu(), "value": stages["value"].cpu(), "elite_idx": stages["elite_idx"].cpu()},
                   os.path.join(out_dir, f"shard{rank}.pt"))
        # did sharded_plan have to re-plan (a bounded inter-workgroup wait gave up while the two processes competed for the
        # compute units; the verdict is collective, so both ranks report the same number)?
        retries = int(planner.last_shard_retries)
        # (2) env-sharded batch: each process plans its share of the environments (recorded tape), actions gathered
        a0, a1 = shard_range(E, world, rank)
        replans = 0
        if a1 > a0:
            def own_share():
                return planner.plan(inp["z0"][a0:a1].contiguous(), inp["disc_pow"][a0:a1].contiguous(), inp["prev_mean"][a0:a1].clone(),
                                    inp["t0"][a0:a1].contiguous(), eval_mode=c["eval_mode"],
                                    task_emb=None if inp["task_emb"] is None else inp["task_emb"][a0:a1].contiguous(),
                                    act_mask=None if inp["act_mask"] is None else inp["act_mask"][a0:a1].contiguous(),
                                    tape={k: v[a0:a1].contiguous() for k, v in inp["tape"].items()})
            loc = own_share()
            torch.cuda.synchronize()
            if planner.take_fault():  # what TDMPC2.act() does: the plan came back NaN, the handle has switched kernels -- plan again
                replans = 1
                loc = own_share()
                torch.cuda.synchronize()
        else:
            loc = torch.empty(0, cfg.action_dim, device=d)
        full = gather_actions(loc, E)
        torch.save(full.cpu(), os.path.join(out_dir, f"envs{rank}.pt"))
        torch.save(torch.tensor([retries, replans]), os.path.join(out_dir, f"faults{rank}.pt"))
        planner.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,path", [("c1", 1), ("c4", 2)])
def test_two_processes_on_one_gpu_shard_a_plan_bit_identically(name, path, tmp_path):
    """2-rank `dist.sharded_plan` with the HIP planner in both ranks == the 1-rank sharded plan, bit for bit (values, elite
    sets, action, _prev_mean), in-kernel Philox, on the 5M fused model and on the 317M layered model (the case the sharding is
    for); the env-sharded batch gathers the actions one process computes alone.  Two handles of two processes coexist."""
    import torch.multiprocessing as mp

    from tdmpc2_amd.dist import sharded_plan

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    saved = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE")}
    try:
        mp.spawn(_two_proc_worker, args=(2, port, str(tmp_path), name, path), nprocs=2, join=True)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    r0, r1 = torch.load(tmp_path / "shard0.pt"), torch.load(tmp_path / "shard1.pt")
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k  # every rank holds the same plan
    # one rank, same seed, fresh handle (call counter 0, like rank 0's)
    from tdmpc2_amd.native import NativePlanner

    c, model, _ = case_on_gpu(name, path, 2)
    planner = NativePlanner(c["cfg"], c["iterations"], dev(), max_envs=max(c["n_envs"], 2), path=path)
    planner.bind_state_dict(model.sd)
    if path == 2:
        planner.set_ksplit(0)
    inp = plan_inputs(c, model)
    pm = inp["prev_mean"].clone()
    stages = planner.debug_buffers(c["n_envs"])
    assert not (dist.is_initialized() and dist.get_world_size() > 1)
    f0, f1 = (torch.load(tmp_path / f"faults{r}.pt").tolist() for r in (0, 1))
    assert f0[0] == f1[0], "the re-plan verdict is collective"
    print(f"two processes on one GPU: sharded_plan re-planned {f0[0]} time(s); env-sharded re-plans per rank {f0[1]}, {f1[1]}")
    if f0[0]:  # the two ranks finished on the kernels without inter-workgroup waits: compare with the same kernels
        planner.set_fuse_ln(0)
        planner.set_cluster(0)
    a = sharded_plan(planner, inp["z0"], inp["disc_pow"], pm, inp["t0"], eval_mode=c["eval_mode"], tape=None, seed=77, stages=stages,
                     task_emb=inp["task_emb"], act_mask=inp["act_mask"])
    torch.cuda.synchronize()
    assert planner.last_shard_retries == 0, "a single process on the GPU: no wait can give up"
    v1, v2 = stages["value"].cpu(), r0["value"]
    if not torch.equal(v1, v2):  # say where: [env, iteration, row]
        bad = (v1 != v2).nonzero()
        per_it = [(int(i), int(((v1 != v2)[:, i]).sum())) for i in range(v1.shape[1])]
        raise AssertionError(f"2 ranks != 1 rank: first difference at {bad[0].tolist()}, differing rows per iteration {per_it}, "
                             f"max |diff| of the first differing iteration "
                             f"{float((v1 - v2)[:, int(bad[0][1])].abs().max()):.3e}")
    assert torch.equal(stages["elite_idx"].cpu(), r0["elite_idx"])
    assert torch.equal(a.cpu(), r0["action"]) and torch.equal(pm.cpu(), r0["prev_mean"])
    want = planner.plan(inp["z0"], inp["disc_pow"], inp["prev_mean"].clone(), inp["t0"], eval_mode=c["eval_mode"], tape=inp["tape"],
                        task_emb=inp["task_emb"], act_mask=inp["act_mask"])
    e0, e1 = torch.load(tmp_path / "envs0.pt"), torch.load(tmp_path / "envs1.pt")
    assert torch.equal(e0, e1)
    # (a one-env share and the two-env call may take different kernels of the fused family -- cluster path or not: same
    # plan to fp32 round-off, not bit for bit)
    assert torch.allclose(e0, want.cpu(), atol=1e-4)
    planner.close()
