"""Multi-GPU layout: independent environments sharded across ranks, one process
per GPU.

Planning needs no data-path collective: every plan depends only on its own
latent, warm-start mean and task.  RCCL (torch.distributed backend "nccl" on
ROCm) is used only (a) to broadcast the world-model weights once at load and
(b) optionally to gather the per-rank actions when one rank owns the env loop.
The same code runs on gloo for the CPU tests.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_envs: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of the environments owned by `rank`; sizes differ by at most 1."""
    base, rem = divmod(n_envs, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


BUCKET_BYTES = 256 << 20  # weight broadcast bucket: large enough for xGMI's per-link rate, small next to 288 GB of HBM


def broadcast_state_dict(sd: Dict[str, torch.Tensor], src: int = 0, bucket_bytes: int = BUCKET_BYTES) -> Dict[str, torch.Tensor]:
    """In-place broadcast of every tensor of a (structurally identical) state dict from `src`, streamed in flat buckets of
    at most `bucket_bytes` per dtype: the 5M model moves in one collective, the 317M model (1.27 GB) in five, and the
    temporary never exceeds one bucket (a tensor larger than a bucket is broadcast in place, without a copy)."""
    if not (dist.is_available() and dist.is_initialized()):
        return sd
    keys = sorted(k for k, v in sd.items() if torch.is_tensor(v))
    by_dtype: Dict[torch.dtype, list] = {}
    for k in keys:
        by_dtype.setdefault(sd[k].dtype, []).append(k)

    def flush(bucket):
        if not bucket:
            return
        flat = torch.cat([sd[k].reshape(-1) for k in bucket])
        dist.broadcast(flat, src=src)
        off = 0
        for k in bucket:
            n = sd[k].numel()
            sd[k].copy_(flat[off:off + n].view_as(sd[k]))
            off += n

    for dt, ks in by_dtype.items():
        bucket, size = [], 0
        for k in ks:
            nbytes = sd[k].numel() * sd[k].element_size()
            if nbytes >= bucket_bytes and sd[k].is_contiguous():
                dist.broadcast(sd[k], src=src)  # big tensors go as they are
                continue
            if size + nbytes > bucket_bytes:
                flush(bucket)
                bucket, size = [], 0
            bucket.append(k)
            size += nbytes
        flush(bucket)
    return sd


def gather_actions(local_actions: torch.Tensor, n_envs: int) -> torch.Tensor:
    """all_gather of the per-rank action blocks [E_rank, A] into [n_envs, A] (env order).
    Ranks may own different counts (shard_range), so blocks are padded to the largest."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_actions
    world = dist.get_world_size()
    sizes = [shard_range(n_envs, world, r) for r in range(world)]
    mx = max(b - a for a, b in sizes)
    stage = torch.device("cpu") if _host_staged() else local_actions.device
    pad = torch.zeros(mx, local_actions.shape[1], dtype=local_actions.dtype, device=stage)
    pad[: local_actions.shape[0]] = local_actions.to(stage)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: b - a] for o, (a, b) in zip(out, sizes)], dim=0).to(local_actions.device)


def _host_staged(group=None) -> bool:
    """True when the group's collectives must see host tensors (gloo)."""
    return dist.get_backend(group) == "gloo"


def _agree_on_stream(backend, seed: int, device, group=None):
    """Make the ranks of a sharded plan draw identical noise: compare the seeds (raise on mismatch) and adopt rank 0's call
    counter (NativePlanner.call_counter / set_call_counter; stand-ins without a counter have no hidden state to align)."""
    seed = int(seed) & (2**64 - 1)
    has_counter = hasattr(backend, "call_counter") and hasattr(backend, "set_call_counter")
    src = dist.get_global_rank(group, 0) if group is not None else 0
    if _host_staged(group):
        device = torch.device("cpu")
    # int64 cannot hold a u64 seed: ship it as two 32-bit halves
    mine = torch.tensor([seed >> 32, seed & 0xFFFFFFFF, backend.call_counter() if has_counter else 0], dtype=torch.int64, device=device)
    ref = mine.clone()
    dist.broadcast(ref, src=src, group=group)
    bad = torch.tensor([int(not torch.equal(mine[:2], ref[:2]))], dtype=torch.int64, device=device)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
    if int(bad.item()):
        raise ValueError("sharded_plan: the Philox seed differs between ranks (it must not depend on the rank: every rank has to "
                         f"sample the same actions); this rank passed {seed:#x}")
    if has_counter:
        backend.set_call_counter(int(ref[2].item()))


def sharded_plan(backend, z0, disc_pow, prev_mean, t0, eval_mode: bool = False, task_emb=None, act_mask=None, tape=None,
                 seed: int = 0, group=None, stages: Optional[dict] = None) -> torch.Tensor:
    """ONE plan per environment with its sample rows split over the ranks of `group` (last row:
    317M-class models at E = 1).  Every rank calls this with IDENTICAL arguments (same z0, same noise tape or Philox seed):
    the prologue, the action sampling and the elite selection + refit are replicated; a rank evaluates only rows
    [rank * N / G, (rank + 1) * N / G) of every plan and the value slices are all-gathered once per CEM iteration
    (N / G * 4 bytes per plan per rank over RCCL / xGMI).  Returns action [E, A] -- the same on every rank; `prev_mean` is
    updated in place.  If a bounded inter-workgroup wait of the kernels gave up on any rank (NativePlanner.take_fault), all
    ranks re-plan the step on the kernels without such waits (`backend.last_shard_retries`).

    `backend` is a `NativePlanner` (or anything with its shard_begin / shard_values / shard_refit / shard_granularity /
    cfg / iterations: the CPU tests drive this function over gloo with an oracle-backed stand-in)."""
    cfg = backend.cfg
    N, E = cfg.num_samples, int(z0.shape[0])
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    gran = backend.shard_granularity
    if N % (world * gran) != 0:
        raise ValueError(f"num_samples {N} does not split into {world} ranges of a multiple of {gran} rows")
    per = N // world
    r0, r1 = rank * per, (rank + 1) * per
    value = torch.zeros(E, N, dtype=torch.float32, device=z0.device)
    action = torch.empty(E, cfg.action_dim, dtype=torch.float32, device=z0.device)
    can_fault = hasattr(backend, "take_fault")
    prev_in = prev_mean.clone() if can_fault else None
    backend.last_shard_retries = 0
    call0 = None
    if can_fault:
        backend.take_fault()  # a fault left over from an EARLIER call (its caller had its chance) must not cost this plan a re-plan
    for attempt in range(2):
        if world > 1 and tape is None:
            _agree_on_stream(backend, seed, z0.device, group)
        if can_fault and hasattr(backend, "call_counter"):
            call0 = backend.call_counter()  # (after the ranks agreed) a re-plan draws the noise of the attempt it replaces
        backend.shard_begin(z0, prev_mean, t0, task_emb=task_emb, act_mask=act_mask, tape=tape, seed=seed)
        for it in range(backend.iterations):
            backend.shard_values(it, r0, r1, z0, disc_pow, value, act_mask=act_mask, seed=seed)
            if world > 1:
                local = value[:, r0:r1].contiguous()
                # RCCL gathers device tensors in place; gloo (CPU tests, and ranks that SHARE one GPU -- RCCL refuses two ranks
                # on one device) takes the 4 KB slices through host memory
                stage = torch.device("cpu") if _host_staged(group) else value.device
                gathered = torch.empty(world, E, per, dtype=value.dtype, device=stage)
                dist.all_gather_into_tensor(gathered.view(-1), local.to(stage).view(-1), group=group)
                value.copy_(gathered.permute(1, 0, 2).reshape(E, N))
            backend.shard_refit(it, value, prev_mean, action, act_mask=act_mask, eval_mode=eval_mode, seed=seed, stages=stages)
        if not can_fault:
            break
        # A bounded inter-workgroup wait of the planner kernels that gave up on ANY rank (another process or kernel held the
        # compute units) made that rank's value slice garbage, and every rank has refitted on it: the ranks agree on the
        # verdict, switch to the kernels without inter-workgroup waits and plan the step again (once: those cannot fault).
        if z0.is_cuda:
            torch.cuda.synchronize(z0.device)
        bad = torch.tensor([int(backend.take_fault() > 0)], dtype=torch.int64,
                           device=torch.device("cpu") if (world > 1 and _host_staged(group)) else z0.device)
        if world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
        if not int(bad.item()) or attempt == 1:
            break
        # The retry is a property of the retry: TDMPC2_TUNE_SAFE_ONCE covers exactly the next shard_begin .. last shard_refit and
        # touches neither the caller's CLUSTER / FUSE_LN settings (explicit or from the environment) nor the handle's own
        # downgrade / re-arm bookkeeping -- a rank that really faulted stays on the safe paths for `rearm_after` calls, with the
        # library's back-off, instead of running into the same wait on every step.
        backend.last_shard_retries += 1
        if hasattr(backend, "plan_safely_once"):
            backend.plan_safely_once(True)
        prev_mean.copy_(prev_in)
        if call0 is not None:
            backend.set_call_counter(call0)
    return action
