"""Slice sharding for the embarrassingly-parallel sampling path: one process per GPU, no data-path collective.

The only collectives are (1) the one-off broadcast of the packed parameter blob from rank 0 (RCCL over xGMI on GPUs,
gloo in the CPU tests) — the analogue of Disc_diff/guided_diffusion/dist_util.py:54-83 — and (2) the final gather of
the sampled slices (256 KB per 256x256 slice).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def world() -> tuple:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    """Rank r takes items r, r+R, r+2R, ...: sizes differ by at most one, empty shards allowed."""
    return list(range(rank, n_items, world_size))


def broadcast_packed(tensors: Dict[str, torch.Tensor], src: int = 0, device: Optional[torch.device] = None) -> None:
    """Broadcast every tensor of ``tensors`` (same names/shapes on all ranks) as ONE flat buffer; in place."""
    rank, ws = world()
    if ws == 1:
        return
    names = sorted(tensors)
    dev = device if device is not None else tensors[names[0]].device
    flat = torch.cat([tensors[n].detach().reshape(-1).to(dev, torch.float32) for n in names])
    dist.broadcast(flat, src)
    off = 0
    with torch.no_grad():
        for n in names:
            t = tensors[n]
            t.copy_(flat[off:off + t.numel()].view_as(t).to(t.device))
            off += t.numel()


def gather_slices(local: torch.Tensor, n_items: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Inverse of shard_indices: rank dst receives the [n_items, ...] tensor in the original order, others None."""
    rank, ws = world()
    if ws == 1:
        return local
    per = (n_items + ws - 1) // ws
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(ws)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    out = torch.empty((n_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(ws):
        idx = shard_indices(n_items, r, ws)
        out[idx] = bufs[r][: len(idx)]
    return out
