"""Multi-GPU layout: independent environments sharded across ranks, one process
per GPU.

Planning needs no data-path collective: every plan depends only on its own
latent, warm-start mean and task.  RCCL (torch.distributed backend "nccl" on
ROCm) is used only (a) to broadcast the world-model weights once at load and
(b) optionally to gather the per-rank actions when one rank owns the env loop.
The same code runs on gloo for the CPU tests.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.distributed as dist


def shard_range(n_envs: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of the environments owned by `rank`; sizes differ by at most 1."""
    base, rem = divmod(n_envs, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_state_dict(sd: Dict[str, torch.Tensor], src: int = 0) -> Dict[str, torch.Tensor]:
    """In-place broadcast of every tensor of a (structurally identical) state dict from `src`.
    Tensors are flattened into one bucket per dtype so the 5M model moves in one collective."""
    if not (dist.is_available() and dist.is_initialized()):
        return sd
    keys = sorted(k for k, v in sd.items() if torch.is_tensor(v))
    by_dtype: Dict[torch.dtype, list] = {}
    for k in keys:
        by_dtype.setdefault(sd[k].dtype, []).append(k)
    for dt, ks in by_dtype.items():
        flat = torch.cat([sd[k].reshape(-1) for k in ks])
        dist.broadcast(flat, src=src)
        off = 0
        for k in ks:
            n = sd[k].numel()
            sd[k].copy_(flat[off:off + n].view_as(sd[k]))
            off += n
    return sd


def gather_actions(local_actions: torch.Tensor, n_envs: int) -> torch.Tensor:
    """all_gather of the per-rank action blocks [E_rank, A] into [n_envs, A] (env order).
    Ranks may own different counts (shard_range), so blocks are padded to the largest."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_actions
    world = dist.get_world_size()
    sizes = [shard_range(n_envs, world, r) for r in range(world)]
    mx = max(b - a for a, b in sizes)
    pad = torch.zeros(mx, local_actions.shape[1], dtype=local_actions.dtype, device=local_actions.device)
    pad[: local_actions.shape[0]] = local_actions
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: b - a] for o, (a, b) in zip(out, sizes)], dim=0)
