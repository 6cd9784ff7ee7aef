"""Multi-GPU sharding of independent signals and of the xi-grid (one process per GPU, torch.distributed).

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


# ---- one signal, spectral grid cut into contiguous slices --------------
def xi_shard(XI, M: int, world: int, rank: int):
    """Slice of the grid xi_m = XI[0] + m*eps_xi, m < M, owned by `rank`:
    returns ([xi_lo, xi_hi], M_local, first_index).  Every rank needs at least two points (the
    transform takes the grid as end points + count, include/fnft_nsev.h:371-376)."""
    if M < 2 * world:
        raise ValueError("xi-grid sharding needs M >= 2*world")
    lo, hi = shard_range(M, world, rank)
    eps_xi = (XI[1] - XI[0]) / (M - 1)
    return [XI[0] + lo * eps_xi, XI[0] + (hi - 1) * eps_xi], hi - lo, lo


def transform_xi_grid(q: Optional[np.ndarray], T, XI, M: int,
                      compute: Callable[[np.ndarray, list, list, int], np.ndarray], dst: int = 0, group=None):
    """Root holds the signal q [D] complex128 (others pass None).  The signal is broadcast, rank r
    evaluates `compute(q, T, XI_r, M_r)` -> complex array [n_parts, M_r] (n_parts = 1, 2 or 3 blocks
    of the reference's contspec layout) on its slice of the grid, and the slices meet in one gather on
    the root, which returns the [n_parts * M] contspec in the reference's layout (None elsewhere)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    meta = [int(q.shape[0])] if rank == dst else [None]
    dist.broadcast_object_list(meta, src=dst, group=group)
    D = meta[0]
    buf = torch.zeros((D, 2), dtype=torch.float64)
    if rank == dst:
        buf.copy_(torch.from_numpy(np.ascontiguousarray(q).view(np.float64).reshape(D, 2)))
    dist.broadcast(buf, src=dst, group=group)
    qq = buf.numpy().reshape(2 * D).view(np.complex128)
    XI_r, M_r, _ = xi_shard(XI, M, world, rank)
    part = np.asarray(compute(qq, list(T), XI_r, M_r), np.complex128)
    n_parts = part.shape[0]
    rows = torch.from_numpy(np.ascontiguousarray(part.T).view(np.float64).reshape(M_r, 2 * n_parts))
    full = gather_shards(rows, M, dst=dst, group=group)
    if full is None:
        return None
    arr = full.numpy().reshape(M, 2 * n_parts).view(np.complex128)   # [M, n_parts]
    return np.ascontiguousarray(arr.T).reshape(n_parts * M)
