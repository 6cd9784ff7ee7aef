"""Multi-GPU sharding of independent signals (one process per GPU, torch.distributed).

The fnft_nsev hot path has no cross-signal dependency: a batch of B
signals is cut into contiguous shards, every rank transforms its shard with no communication,
and the result shards meet in ONE gather on the root (RCCL over xGMI when the backend is "nccl",
gloo on CPU for tests).  Nothing here computes a transform; `compute` is injected (the GPU plan in
production, a stand-in in the CPU tests).
"""
from typing import Callable, List, Optional, Tuple

import numpy as np


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) block of rank; the first n_items % world ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_sizes(n_items: int, world: int) -> List[int]:
    return [shard_range(n_items, world, r)[1] - shard_range(n_items, world, r)[0] for r in range(world)]


def gather_shards(local, n_items: int, dst: int = 0, group=None):
    """Gather per-rank result shards (tensor [n_local, L], real dtype) on rank dst.

    Shards may differ by one row; they are padded to the largest shard for the collective (a
    single dist.gather) and trimmed on the root.  Returns the [n_items, L] tensor on dst, None
    elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_items, world)
    mx = max(sizes)
    if local.shape[0] != sizes[rank]:
        raise ValueError("rank %d holds %d rows, expected %d" % (rank, local.shape[0], sizes[rank]))
    pad = local
    if local.shape[0] < mx:
        pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    bufs = None
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.gather(pad.contiguous(), bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: sizes[r]] for r in range(world)], dim=0)


def transform_batch(signals: Optional[np.ndarray], n_signals: int,
                    compute: Callable[[np.ndarray, int], "object"], dst: int = 0, group=None):
    """Root holds `signals` [n_signals, D] complex128 (others pass None).  Scatter the shards,
    run `compute(shard, first_index)` -> real tensor [n_local, L] on every rank, gather on root."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(n_signals, world, rank)
    sizes = shard_sizes(n_signals, world)
    meta = [None]
    if rank == dst:
        meta = [int(signals.shape[1])]
    dist.broadcast_object_list(meta, src=dst, group=group)
    D = meta[0]
    mx = max(sizes)
    recv = torch.zeros((mx, D, 2), dtype=torch.float64)
    chunks = None
    if rank == dst:
        chunks = []
        for r in range(world):
            a, b = shard_range(n_signals, world, r)
            t = torch.zeros((mx, D, 2), dtype=torch.float64)
            if b > a:
                t[: b - a] = torch.from_numpy(np.ascontiguousarray(signals[a:b]).view(np.float64).reshape(b - a, D, 2))
            chunks.append(t)
    dist.scatter(recv, chunks, src=dst, group=group)
    shard = recv[: hi - lo].numpy().reshape(hi - lo, D * 2).view(np.complex128)
    out = compute(shard, lo)
    return gather_shards(out, n_signals, dst=dst, group=group)
