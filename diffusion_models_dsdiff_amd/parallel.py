"""Slice sharding for the embarrassingly-parallel sampling path: one process per GPU, no data-path collective.

The only collectives are (1) the one-off broadcast of the packed parameter blob from rank 0 (RCCL over xGMI on GPUs,
gloo in the CPU tests) — the analogue of Disc_diff/guided_diffusion/dist_util.py:54-83 — and (2) the final gather of
the sampled slices (256 KB per 256x256 slice).
"""
from __future__ import annotations

import contextlib
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

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


@contextlib.contextmanager
def empty_init():
    """Ranks other than the broadcast source: build the module with UNINITIALISED parameter storage (no 981 M-element host
    initialisation whose values the broadcast overwrites anyway — 8 ranks on one node would each spend the host time and RAM
    rank 0 spends)."""
    from .UNet_DS_Diff.model import NativeModule
    old = NativeModule._skip_init
    NativeModule._skip_init = True
    try:
        yield
    finally:
        NativeModule._skip_init = old


def plan_buckets(named_shapes: Sequence[Tuple[str, Tuple[int, ...]]], bucket_elems: int) -> List[List[Tuple[str, Tuple[int, ...], int]]]:
    """Consecutive parameters packed into buckets of at most bucket_elems elements (a larger parameter gets a bucket of its
    own); every entry is (name, shape, offset in the bucket).  Deterministic: every rank derives the same plan."""
    out, cur, fill = [], [], 0
    for name, shape in named_shapes:
        n = 1
        for d in shape:
            n *= int(d)
        if cur and fill + n > bucket_elems:
            out.append(cur)
            cur, fill = [], 0
        cur.append((name, tuple(int(d) for d in shape), fill))
        fill += n
    if cur:
        out.append(cur)
    return out


def broadcast_params_bucketed(named_shapes: Sequence[Tuple[str, Tuple[int, ...]]], get_src: Optional[Callable[[str], torch.Tensor]],
                              put: Callable[[str, torch.Tensor], None], src: int = 0, device: Optional[torch.device] = None,
                              bucket_elems: int = 64 << 20) -> int:
    """The one-off weight distribution (the analogue of Disc_diff/guided_diffusion/dist_util.py:54-83, which ships the
    checkpoint in 1 GiB MPI chunks): rank `src` packs consecutive parameters into flat fp32 buckets (256 MB by default), each
    bucket is ONE broadcast (RCCL over xGMI on GPUs), and every rank hands the views to `put(name, tensor)` — for a native
    module that is the upload into the library's parameter slab straight from the bucket, so no rank ever holds a second full
    copy of the 3.93 GB on its device and non-source ranks never initialise or keep host values.
    get_src(name) -> the source tensor (called on rank src only).  Returns the number of buckets."""
    rank, ws = world()
    buckets = plan_buckets(named_shapes, bucket_elems)
    for b in buckets:
        n = b[-1][2] + _numel(b[-1][1])
        buf = torch.empty(n, dtype=torch.float32, device=device)
        if rank == src or ws == 1:
            for name, shape, off in b:
                buf[off:off + _numel(shape)].copy_(get_src(name).detach().reshape(-1).to(torch.float32))
        if ws > 1:
            dist.broadcast(buf, src)
        for name, shape, off in b:
            put(name, buf[off:off + _numel(shape)].view(shape))
        if buf.is_cuda:
            torch.cuda.current_stream().synchronize()   # the uploads read the bucket: keep it alive until they are done
    return len(buckets)


def _numel(shape) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n
