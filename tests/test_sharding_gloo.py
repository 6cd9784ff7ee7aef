"""world_size-2 (and 3) gloo runs of the multi-GPU plumbing on CPU: shard bounds, scatter, the
single gather and the root-side ordering.  The per-shard compute is a stand-in (the product has no
CPU path); what is checked is that row k of the gathered result is the result of signal k."""
import os
import socket
import sys

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


# ---- xi-grid sharding of one signal, oracle as the per-rank engine ----------------
def _oracle_slice(q, T, XI_r, M_r):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import load_oracle
    rc, cs = load_oracle().fnft_nsev(q, T, M_r, XI_r, kappa=1, disc="2SPLIT4B", cstype="BOTH")
    assert rc == 0
    return cs.reshape(3, M_r)


def _xi_worker(rank, world, port, D, M, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        T, XI = [-25.0, 25.0], [-1.4, 1.6]
        t = T[0] + np.arange(D) * (T[1] - T[0]) / (D - 1)
        sig = (3.2j / np.cosh(t)).astype(np.complex128) if rank == 0 else None
        res = sharding.transform_xi_grid(sig, T, XI, M, _oracle_slice, dst=0)
        if rank == 0:
            ref = _oracle_slice(sig, T, XI, M).reshape(3 * M)
            err = float(np.sum(np.abs(res - ref)) / np.sum(np.abs(ref)))
            q.put(err < 1e-12 and res.shape == (3 * M,))
        else:
            q.put(res is None)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,M", [(2, 37), (3, 16)])
def test_xi_grid_sharding(world, M):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xi_worker, args=(r, world, port, 256, M, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(results)


def test_xi_shard_ranges():
    XI, M = [-1.4, 1.6], 37
    eps = (XI[1] - XI[0]) / (M - 1)
    pts = []
    for r in range(3):
        xr, mr, lo = sharding.xi_shard(XI, M, 3, r)
        pts += list(xr[0] + np.arange(mr) * (xr[1] - xr[0]) / (mr - 1))
    assert np.allclose(pts, XI[0] + np.arange(M) * eps, atol=1e-15)
    with pytest.raises(ValueError):
        sharding.xi_shard(XI, 5, 3, 0)


# ---- sample-axis sharding of one signal, oracle as the per-rank engine --------------
def _oracle_sample_axis_engine(disc, deg0):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import load_oracle
    orc = load_oracle()

    def subtree(qb, eps_t):
        rc, _, tm, W = orc.nse_fscatter(qb, eps_t, 1, disc)
        assert rc == 0
        return tm, W

    def combine(deg, n, p):
        _, tm, W = orc.poly_fmult2x2(deg, n, p)
        return tm, W

    return orc, sharding.SampleAxisEngine(subtree, combine, orc.poly_chirpz, deg0,
                                          disc in ("2SPLIT2A", "2SPLIT2_MODAL"))


def _sample_axis_worker(rank, world, port, D, M, disc, deg0, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        T, XI = [-25.0, 25.0], [-1.4, 1.6]
        t = T[0] + np.arange(D) * (T[1] - T[0]) / (D - 1)
        sig = (3.2j / np.cosh(t) * np.exp(0.3j * t)).astype(np.complex128) if rank == 0 else None
        orc, eng = _oracle_sample_axis_engine(disc, deg0)
        res = sharding.transform_sample_axis(sig, T, XI, M, eng, dst=0)
        if rank == 0:
            rc, ref = orc.fnft_nsev(sig, T, M, XI, kappa=1, disc=disc, cstype="BOTH")
            err = float(np.sum(np.abs(res - ref)) / np.sum(np.abs(ref)))
            q.put(rc == 0 and err < 1e-12 and res.shape == (3 * M,))
        else:
            q.put(res is None)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,D,disc,deg0", [(2, 256, "2SPLIT2_MODAL", 1), (2, 192, "2SPLIT4B", 2),
                                               (4, 256, "2SPLIT4B", 2)])
def test_sample_axis_sharding(world, D, disc, deg0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sample_axis_worker, args=(r, world, port, D, 33, disc, deg0, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(results)


# ---- per-step gather with persistent buffers (what bench.py --gpus N runs every step) --------------
def _gatherer_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = sharding.ShardGather((6, 2), torch.float64, dst=0, depth=2)
        assert sharding.wire_device() == torch.device("cpu")
        ok = True
        bufs = [torch.zeros(6, 2, dtype=torch.float64) for _ in range(2)]
        for step in range(5):
            g.reserve()                                   # buffer step % 2 may be overwritten now
            bufs[step % 2].fill_(float(100 * step + rank))
            g.start(bufs[step % 2], step)
            if step >= 1 and rank == 0:                   # step - 1 has been gathered once two are in flight
                g.reserve()
                got = g.result(step - 1)
                ok = ok and all(float(got[r][0, 0]) == 100 * (step - 1) + r for r in range(world))
        g.wait()
        if rank == 0:
            got = g.result(4)
            ok = ok and all(float(got[r][3, 1]) == 400 + r for r in range(world))
        q.put(ok)
    finally:
        dist.destroy_process_group()


def test_shard_gather_per_step():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gatherer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(results)
