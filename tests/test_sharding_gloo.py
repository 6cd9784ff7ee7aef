"""world_size-2 (and 3) gloo runs of the multi-GPU plumbing on CPU: shard bounds, scatter, the
single gather and the root-side ordering.  The per-shard compute is a stand-in (the product has no
CPU path); what is checked is that row k of the gathered result is the result of signal k."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fnft_amd import sharding


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 8, 9, 512, 513):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = sharding.shard_range(n, world, r)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
            assert sum(sharding.shard_sizes(n, world)) == n
            assert max(sharding.shard_sizes(n, world)) - min(sharding.shard_sizes(n, world)) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stand_in(shard, first):
    # deterministic per-signal "spectrum": [sum, weighted sum, index] -> real view
    n, D = shard.shape
    w = np.arange(1, D + 1)
    out = np.stack([shard.sum(axis=1), (shard * w).sum(axis=1), np.arange(first, first + n) + 0j], axis=1)
    return torch.from_numpy(out.view(np.float64).reshape(n, 6))


def _worker(rank, world, port, n_signals, D, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(7)
        sig = (rng.standard_normal((n_signals, D)) + 1j * rng.standard_normal((n_signals, D))) if rank == 0 else None
        res = sharding.transform_batch(sig, n_signals, _stand_in, dst=0)
        if rank == 0:
            ref = _stand_in(sig, 0)
            q.put(bool(torch.equal(res, ref)) and res.shape == (n_signals, 6))
        else:
            q.put(res is None)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_signals", [(2, 8), (2, 5), (3, 7)])
def test_scatter_compute_gather(world, n_signals):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_signals, 16, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(results)
