"""Two ranks (one process each, gloo rendezvous on 127.0.0.1) sharing the one GPU of the test box: the three
partitions of SURVEY 8e with the C-ABI library as the per-rank engine, checked on the root against the oracle.
RCCL itself needs more than one GPU and runs only in the driver's multi-GPU bench; what is covered here is the
whole per-rank path of a multi-process job -- every rank loads libfnft_amd.so and computes on the GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

T, XI = [-25.0, 25.0], [-1.4, 1.6]
DISC, DEG0 = "2SPLIT4B", 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _signals(n, D):
    t = T[0] + np.arange(D) * (T[1] - T[0]) / (D - 1)
    return np.stack([((1.0 + 0.4 * k) * 1j / np.cosh(t - 0.3 * k) * np.exp(0.2j * k * t)) for k in range(n)]
                    ).astype(np.complex128)


def _rel(a, b):
    return float(np.sum(np.abs(a - b)) / np.sum(np.abs(b)))


def _worker(rank, world, port, mode, D, M, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fnft_amd import capi, sharding
        L = capi.load()
        assert L.fnft_amd_device_count() >= 1, capi.last_error()
        # one process per GPU: the rank selects its device once; the library computes on the CURRENT device
        # (include/fnft_amd.h, "Device rule") and never changes it
        local_dev = rank % torch.cuda.device_count()
        torch.cuda.set_device(local_dev)
        assert L.fnft_amd_current_device() == local_dev
        cuda = torch.device("cuda", local_dev)
        ref = None
        if mode == "batch":
            # device-resident path: shards are CUDA tensors, the plan reads and writes HBM, the result shards
            # go into the gather as tensors (under nccl without touching the host; gloo here stages them)
            n = 5
            sig = _signals(n, D) if rank == 0 else None
            compute = sharding.plan_batch_compute(T, XI, M, discretization=DISC, kappa=1, contspec_type="BOTH")
            res = sharding.transform_batch(sig, n, compute, dst=0, data_device=cuda)
            assert all(pl.device == local_dev for pl in compute.plans.values())   # the plan lives on this rank's GPU
            if rank == 0:
                assert res.is_cuda
                res = res.cpu().numpy().reshape(n, 6 * M).view(np.complex128)
        elif mode == "batch_host":
            # the drop-in with host pointers as the per-rank engine (numpy shards)
            n = 5
            sig = _signals(n, D) if rank == 0 else None

            def compute(shard, first):
                rows = []
                for s in shard:
                    rc, cs = capi.fnft_nsev(s, T, M, XI, kappa=1, discretization=DISC, contspec_type="BOTH")
                    assert rc == 0, capi.last_error()
                    rows.append(cs)
                out = np.stack(rows) if rows else np.zeros((0, 3 * M), np.complex128)
                return torch.from_numpy(out.view(np.float64).reshape(len(rows), 6 * M))

            res = sharding.transform_batch(sig, n, compute, dst=0)
            assert L.fnft_amd_current_device() == local_dev   # the host entry points left the device alone
            if rank == 0:
                res = res.numpy().reshape(n, 6 * M).view(np.complex128)
        else:
            sig = _signals(2, D)[1] if rank == 0 else None
            if mode == "xi":
                plans = {}

                def compute(qq, T_, XI_r, M_r):   # qq: CUDA tensor; the slice of the grid on this rank's GPU
                    pl = plans.setdefault(M_r, capi.Plan(D, M_r, batch=1, discretization=DISC, device=local_dev))
                    out = torch.zeros(3 * M_r, dtype=torch.complex128, device=qq.device)
                    st = torch.cuda.current_stream().cuda_stream
                    rc = pl.contspec_device(qq.contiguous().data_ptr(), out.data_ptr(), T_, XI_r, kappa=1,
                                            contspec_type="BOTH", normalization_flag=1, stream=st)
                    assert rc == 0 and pl.finish(st) == 0, capi.last_error()
                    return out.reshape(3, M_r)
                res = sharding.transform_xi_grid(sig, T, XI, M, compute, dst=0, data_device=cuda)
                if rank == 0:
                    res = res.cpu().numpy()
            else:
                eng = sharding.plan_sample_axis_engine(DISC, 1, DEG0)
                res = sharding.transform_sample_axis(sig, T, XI, M, eng, dst=0, data_device=cuda)
                assert all(pl.device == local_dev for pl in eng.plans.values())
        if rank == 0:
            from oracle import load_oracle
            orc = load_oracle()
            sigs = sig if mode.startswith("batch") else sig[None, :]
            res = res if mode.startswith("batch") else res[None, :]
            ok = True
            for k in range(sigs.shape[0]):
                rc, ref = orc.fnft_nsev(sigs[k], T, M, XI, kappa=1, disc=DISC, cstype="BOTH")
                ok = ok and rc == 0 and _rel(res[k], ref) < 1e-11
            q.put(ok)
        else:
            q.put(res is None)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["batch", "batch_host", "xi", "samples"])
def test_two_ranks_one_gpu(mode):
    world, D, M = 2, 1024, 65
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, D, M, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = [q.get(timeout=240) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    assert all(results)
