"""Multi-GPU sharding of independent signals, of the xi-grid and of the sample axis (one process per GPU,
torch.distributed).

The fnft_nsev hot path has no cross-signal dependency: a batch of B signals is
cut into contiguous shards, every rank transforms its shard with no communication, and the result shards
meet in ONE gather on the root.

Where the bytes live.  Every collective here moves torch tensors that live on the "wire device" of the
process group: the rank's current GPU when the backend is nccl (= RCCL over xGMI on ROCm: device buffers go
into the collective as they are, nothing is staged through the host), the CPU when it is gloo (the CPU tests,
and rehearsals on a box with fewer GPUs than ranks).  `data_device` says where the shards handed to `compute`
and the results it returns live: with data_device = cuda and the nccl backend the whole path -- scatter,
plan, gather -- is device-resident; with gloo the tensors cross to the wire device just around the collective.
Nothing here computes a transform; `compute` is injected (`plan_batch_compute` for the GPU plan, stand-ins
in the CPU tests).
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


def wire_device(group=None):
    """Device the collectives of this process group take tensors on: the current GPU under nccl (RCCL),
    the CPU under gloo."""
    import torch
    import torch.distributed as dist

    if "nccl" in str(dist.get_backend(group)).lower():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _wire(t, group=None):
    """`t` as the collective takes it: unchanged when it already lives on the wire device."""
    dev = wire_device(group)
    return t if t.device == dev else t.to(dev)


def gather_shards(local, n_items: int, dst: int = 0, group=None):
    """Gather per-rank result shards (tensor [n_local, L], real dtype, any device) on rank dst.

    Shards may differ by one row; they are padded to the largest shard for the collective (a
    single dist.gather) and trimmed on the root.  Returns the [n_items, L] tensor on dst -- on the
    device `local` lives on --, None elsewhere."""
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
    pad = _wire(pad.contiguous(), group)
    bufs = None
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: sizes[r]] for r in range(world)], dim=0).to(local.device)


class ShardGather:
    """Per-step gather of equally shaped result shards with persistent buffers (bench.py, serving loops):
    `depth` buffer sets rotate so that the gather of step i overlaps the compute of step i + 1.
    local tensors may live on any device; they enter the collective on the wire device."""

    def __init__(self, shard_shape, dtype, dst: int = 0, group=None, depth: int = 2):
        import torch
        import torch.distributed as dist

        self.group, self.dst = group, dst
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.dev = wire_device(group)
        self.bufs = None
        if self.rank == dst:
            self.bufs = [[torch.zeros(tuple(shard_shape), dtype=dtype, device=self.dev) for _ in range(self.world)]
                         for _ in range(depth)]
        self.depth = depth
        self.pending = []

    def reserve(self):
        """Call BEFORE overwriting the result buffer of the step that is about to run: waits until fewer than
        `depth` gathers are in flight, i.e. until the gather that read this buffer `depth` steps ago is done
        (under nccl the wait is a stream dependency, not a host block)."""
        while len(self.pending) >= self.depth:
            self.pending.pop(0).wait()

    def start(self, local, step: int):
        """Enqueue the gather of this step's shard; at most `depth` gathers are in flight."""
        import torch.distributed as dist

        self.reserve()
        src = _wire(local, self.group)
        w = dist.gather(src, self.bufs[step % self.depth] if self.rank == self.dst else None, dst=self.dst,
                        group=self.group, async_op=True)
        self.pending.append(w)
        return w

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []

    def result(self, step: int):
        """The world's shards of `step` on the root (list of tensors on the wire device), None elsewhere."""
        return self.bufs[step % self.depth] if self.rank == self.dst else None


def plan_batch_compute(T, XI, M: int, discretization: str = "2SPLIT2_MODAL", kappa: int = 1,
                       contspec_type: str = "BOTH", device=None):
    """`compute` for transform_batch backed by the device-resident plan of the C ABI (include/fnft_amd.h,
    section 3): shard = complex128 tensor [n_local, D] on the GPU, result = float64 tensor [n_local, 2*cs_len]
    on the GPU; no host staging.  One plan per shard size is kept."""
    import torch
    from . import capi

    plans = {}
    nparts = {"RHO": 1, "REFLECTION_COEFFICIENT": 1, "AB": 2, "BOTH": 3}[contspec_type]

    def compute(shard, first):
        if not shard.is_cuda:
            raise RuntimeError("plan_batch_compute needs data_device = cuda (the product has no CPU path)")
        n, D = int(shard.shape[0]), int(shard.shape[1])
        out = torch.zeros((n, nparts * M), dtype=torch.complex128, device=shard.device)
        if n == 0:
            return torch.view_as_real(out).reshape(0, 2 * nparts * M)
        dev = shard.device.index if shard.device.index is not None else torch.cuda.current_device()
        key = (n, D, dev)
        if key not in plans:
            plans[key] = capi.Plan(D, M, batch=n, discretization=discretization, device=dev)
        pl = plans[key]
        stream = torch.cuda.current_stream(shard.device).cuda_stream
        rc = pl.contspec_device(shard.contiguous().data_ptr(), out.data_ptr(), T, XI, kappa=kappa,
                                contspec_type=contspec_type, normalization_flag=1, stream=stream)
        if rc == 0:
            rc = pl.finish(stream)
        if rc != 0:
            raise RuntimeError("fnft_amd_nsev_contspec_device rc=%d (%s)" % (rc, capi.last_error()))
        return torch.view_as_real(out).reshape(n, 2 * nparts * M)

    compute.plans = plans
    return compute


def transform_batch(signals, n_signals: int, compute: Callable, dst: int = 0, group=None, data_device=None):
    """Root holds `signals` [n_signals, D] complex128 -- a numpy array or a torch tensor on any device --
    (others pass None).  Scatter the shards, run `compute(shard, first_index)` -> real tensor [n_local, L] on
    every rank, gather on root.

    data_device None (default): `compute` gets a numpy shard (host engines, the CPU tests) and may return a
    tensor on any device.  data_device = a torch device: `compute` gets a complex128 tensor on that device
    and the gathered result is returned there (`plan_batch_compute` with data_device = cuda)."""
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
    wdev = wire_device(group)
    recv = torch.zeros((mx, D, 2), dtype=torch.float64, device=wdev)
    chunks = None
    if rank == dst:
        if isinstance(signals, torch.Tensor):
            sv = torch.view_as_real(signals.to(torch.complex128).contiguous()).to(wdev)
        else:
            sv = torch.from_numpy(np.ascontiguousarray(signals, dtype=np.complex128).view(np.float64)
                                  .reshape(n_signals, D, 2)).to(wdev)
        chunks = []
        for r in range(world):
            a, b = shard_range(n_signals, world, r)
            t = torch.zeros((mx, D, 2), dtype=torch.float64, device=wdev)
            if b > a:
                t[: b - a] = sv[a:b]
            chunks.append(t)
    dist.scatter(recv, chunks, src=dst, group=group)
    if data_device is None:
        shard = recv[: hi - lo].cpu().numpy().reshape(hi - lo, D * 2).view(np.complex128)
        out = compute(shard, lo)
    else:
        shard = torch.view_as_complex(recv[: hi - lo].to(torch.device(data_device)).contiguous())
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


def transform_xi_grid(q, T, XI, M: int, compute: Callable, dst: int = 0, group=None, data_device=None):
    """Root holds the signal q [D] complex128 (others pass None).  The signal is broadcast, rank r
    evaluates `compute(q, T, XI_r, M_r)` -> complex array [n_parts, M_r] (n_parts = 1, 2 or 3 blocks
    of the reference's contspec layout) on its slice of the grid, and the slices meet in one gather on
    the root, which returns the [n_parts * M] contspec in the reference's layout (None elsewhere).
    data_device as in transform_batch: None = numpy in / numpy out; a torch device = `compute` gets the signal
    as a complex128 tensor on that device, may return a tensor there, and the root gets a tensor."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    meta = [int(q.shape[0])] if rank == dst else [None]
    dist.broadcast_object_list(meta, src=dst, group=group)
    D = meta[0]
    wdev = wire_device(group)
    buf = torch.zeros((D, 2), dtype=torch.float64, device=wdev)
    if rank == dst:
        if isinstance(q, torch.Tensor):
            buf.copy_(torch.view_as_real(q.to(torch.complex128).contiguous()))
        else:
            buf.copy_(torch.from_numpy(np.ascontiguousarray(q, dtype=np.complex128).view(np.float64).reshape(D, 2)))
    dist.broadcast(buf, src=dst, group=group)
    XI_r, M_r, _ = xi_shard(XI, M, world, rank)
    if data_device is None:
        qq = buf.cpu().numpy().reshape(2 * D).view(np.complex128)
        part = compute(qq, list(T), XI_r, M_r)
    else:
        part = compute(torch.view_as_complex(buf.to(torch.device(data_device))), list(T), XI_r, M_r)
    if isinstance(part, torch.Tensor):       # [n_parts, M_r] complex on any device
        n_parts = int(part.shape[0])
        rows = torch.view_as_real(part.to(torch.complex128).transpose(0, 1).contiguous()).reshape(M_r, 2 * n_parts)
    else:
        part = np.asarray(part, np.complex128)
        n_parts = part.shape[0]
        rows = torch.from_numpy(np.ascontiguousarray(part.T).view(np.float64).reshape(M_r, 2 * n_parts))
    full = gather_shards(rows, M, dst=dst, group=group)
    if full is None:
        return None
    if data_device is not None:
        arr = torch.view_as_complex(full.reshape(M, n_parts, 2).contiguous())    # [M, n_parts]
        return arr.transpose(0, 1).contiguous().reshape(n_parts * M)
    arr = full.cpu().numpy().reshape(M, 2 * n_parts).view(np.complex128)   # [M, n_parts]
    return np.ascontiguousarray(arr.T).reshape(n_parts * M)


# ---- one signal, sample axis cut into contiguous blocks ------------------
# The transfer matrix of the whole signal is the ordered product of the transfer matrices of its
# blocks (sample D-1 is the leftmost factor, fnft__akns_fscatter.c:120).  Rank g builds the sub-tree
# of samples [g*D/G, (g+1)*D/G) -- a 2x2 polynomial matrix of degree D*deg0/G and its exponent W_g --,
# the G matrices meet in ONE gather, and the root runs the last log2(G) levels and the evaluation on
# the xi-grid.  Nothing here computes: the three steps are injected (`SampleAxisEngine`; the C-ABI
# seams fnft__nse_fscatter / fnft__poly_fmult2x2 / fnft__poly_chirpz in production, stand-ins in the
# CPU tests).  For the splitting schemes with upsampling factor 1 (the 2SPLIT* family).
class SampleAxisEngine:
    """subtree(q_block, eps_t) -> (tm[4, d+1], W); combine(deg, n, p[4, n*(deg+1)]) -> (tm[4, n*deg+1], W);
    chirpz(p[N], A, V, M) -> H[M]; deg0 = polynomial degree of one sample's matrix; shifted = the
    discretization is 2SPLIT2A or 2SPLIT2_MODAL (extra eps_t/deg0 in two of the phase factors)."""

    def __init__(self, subtree, combine, chirpz, deg0: int, shifted: bool):
        self.subtree, self.combine, self.chirpz = subtree, combine, chirpz
        self.deg0, self.shifted = int(deg0), bool(shifted)


def capi_sample_axis_engine(discretization: str, kappa: int, deg0: int) -> SampleAxisEngine:
    """The production engine: the private-layer seams of the C ABI (include/fnft_amd.h), GPU only."""
    from . import capi

    def chk(rc, what):
        if rc != 0:
            raise RuntimeError("%s rc=%d (%s)" % (what, rc, capi.last_error()))

    def subtree(qb, eps_t):
        rc, _, tm, W = capi.nse_fscatter(qb, eps_t, kappa, discretization)
        chk(rc, "fnft__nse_fscatter")
        return tm, W

    def combine(deg, n, p):
        rc, _, tm, W = capi.poly_fmult2x2(deg, n, p)
        chk(rc, "fnft__poly_fmult2x2")
        return tm, W

    def chirpz(p, A, V, M):
        rc, H = capi.poly_chirpz(p, A, V, M)
        chk(rc, "fnft__poly_chirpz")
        return H

    return SampleAxisEngine(subtree, combine, chirpz, deg0, discretization in ("2SPLIT2A", "2SPLIT2_MODAL"))


def plan_sample_axis_engine(discretization: str, kappa: int, deg0: int) -> SampleAxisEngine:
    """Production engine with the block matrices kept on the GPU: `subtree` takes the block as a complex128
    CUDA tensor, runs the device-resident plan (coefficients + tree only: a plan with M = 0) and returns the
    transfer matrix as a CUDA tensor [4, d+1] (fnft_amd_plan_get_transfer_matrix_device, device-to-device), so
    that transform_sample_axis(..., data_device=cuda) gathers the G matrices GPU-to-GPU.  The root's product
    and evaluation use the same seams as capi_sample_axis_engine."""
    import torch
    from . import capi

    host = capi_sample_axis_engine(discretization, kappa, deg0)
    plans = {}

    def subtree(qb, eps_t):
        if not isinstance(qb, torch.Tensor):
            return host.subtree(qb, eps_t)
        Db = int(qb.numel())
        dev = qb.device.index if qb.device.index is not None else torch.cuda.current_device()
        if (Db, dev) not in plans:
            plans[(Db, dev)] = capi.Plan(Db, 0, batch=1, discretization=discretization, device=dev)
        pl = plans[(Db, dev)]
        stream = torch.cuda.current_stream(qb.device).cuda_stream
        # the plan derives its step from T and the sample count: give it the block's span at the grid's step
        Tb = [0.0, eps_t * (Db - 1)]
        rc = pl.contspec_device(qb.contiguous().data_ptr(), 0, Tb, [0.0, 1.0], kappa=kappa, contspec_type="BOTH",
                                normalization_flag=1, stream=stream)
        if rc != 0:
            raise RuntimeError("fnft_amd_nsev_contspec_device rc=%d (%s)" % (rc, capi.last_error()))
        tm = torch.zeros((4, Db * deg0 + 1), dtype=torch.complex128, device=qb.device)
        rc, deg, W = pl.transfer_matrix_device(tm.data_ptr(), 0, stream)
        if rc != 0 or deg != Db * deg0:
            raise RuntimeError("transfer matrix rc=%d deg=%d (%s)" % (rc, deg, capi.last_error()))
        return tm, W

    eng = SampleAxisEngine(subtree, host.combine, host.chirpz, deg0, host.shifted)
    eng.plans = plans

    def device_root(rows, d, D, T, XI, M):
        """The root's step with everything on the GPU: rows [G, 8(d+1)+1] float64 CUDA tensor (the gathered block
        matrices and their exponents) -> [rho | a | b] as a complex128 CUDA tensor.  The product of the G matrices is
        fnft_amd_poly_fmult2x2_device (the last block of samples is the leftmost factor), the evaluation
        fnft_amd_nsev_contspec_from_tm_device."""
        G = int(rows.shape[0])
        dev = rows.device
        Ws = [int(w) for w in rows[:, -1].cpu().tolist()]
        tms = torch.view_as_complex(rows[:, :-1].contiguous().view(G, 4, d + 1, 2))
        stream = torch.cuda.current_stream(dev).cuda_stream
        if G == 1:
            tm_all, Wc = tms[0].contiguous(), 0
        else:
            p = torch.stack([torch.cat([tms[G - 1 - j, e] for j in range(G)]) for e in range(4)]).contiguous()
            tm_all = torch.zeros((4, G * d + 1), dtype=torch.complex128, device=dev)
            rc, dd, Wc = capi.poly_fmult2x2_device(d, G, p.data_ptr(), tm_all.data_ptr(), stream)
            if rc != 0 or dd != G * d:
                raise RuntimeError("fnft_amd_poly_fmult2x2_device rc=%d deg=%d (%s)" % (rc, dd, capi.last_error()))
        di = dev.index if dev.index is not None else torch.cuda.current_device()
        key = ("root", D, M, di)
        if key not in plans:
            plans[key] = capi.Plan(D, M, batch=1, discretization=discretization, device=di)
        out = torch.zeros(3 * M, dtype=torch.complex128, device=dev)
        rc = plans[key].contspec_from_tm_device(tm_all.data_ptr(), sum(Ws) + Wc, out.data_ptr(), T, XI, "BOTH", stream)
        if rc == 0:
            rc = plans[key].finish(stream)
        if rc != 0:
            raise RuntimeError("fnft_amd_nsev_contspec_from_tm_device rc=%d (%s)" % (rc, capi.last_error()))
        return out

    eng.device_root = device_root
    return eng


def contspec_from_transfer_matrix(tm: np.ndarray, W: int, D: int, T, XI, M: int, eng: SampleAxisEngine):
    """Host epilogue on the root: [rho | a | b] (contspec_type BOTH) from the transfer matrix of the whole
    signal.  Follows src/fnft_nsev.c:744-891: z-grid parameters (:822-827, lambda -> z of
    fnft__akns_discretization.c:204-219), chirp-z on entries 11 and 21 (:829-833), phase factors of
    fnft__nse_discretization.c:240-379 with boundary coefficient 0.5, epilogue :849-876."""
    eps_t = (T[1] - T[0]) / (D - 1)
    eps_xi = (XI[1] - XI[0]) / (M - 1)
    V = np.exp(2j * eps_xi * eps_t / eng.deg0)
    A = np.exp(-2j * XI[0] * eps_t / eng.deg0)
    H11 = np.asarray(eng.chirpz(tm[0], A, V, M))
    H21 = np.asarray(eng.chirpz(tm[2], A, V, M))
    if np.any(H11 == 0):
        raise ZeroDivisionError("a(xi) = 0 on the grid (FNFT_EC_DIV_BY_ZERO)")
    sh = eps_t / eng.deg0 if eng.shifted else 0.0
    tp, tm_ = T[1] + 0.5 * eps_t, T[0] - 0.5 * eps_t
    pf_rho = -2.0 * tp + sh
    pf_a = -eps_t * D + tp - tm_
    pf_b = -eps_t * D - tp - tm_ + sh
    xi = XI[0] + eps_xi * np.arange(M)
    scale = 2.0 ** W
    return np.concatenate([H21 * np.exp(1j * xi * pf_rho) / H11,
                           H11 * scale * np.exp(1j * xi * pf_a),
                           H21 * scale * np.exp(1j * xi * pf_b)])


def combine_block_matrices(tms, Ws, eng: SampleAxisEngine):
    """Root side of the sample-axis split: block matrices tms[g] [4, d+1] (g = position of the block in
    the signal) with exponents Ws[g] -> (transfer matrix of the whole signal, its exponent)."""
    G = len(tms)
    if G == 1:
        return np.asarray(tms[0]), int(Ws[0])
    d = tms[0].shape[1] - 1
    # fnft__poly_fmult2x2 input layout: entry-major, factor j of the product in block j; the last
    # block of samples is the leftmost factor
    p = np.stack([np.concatenate([tms[G - 1 - j][e] for j in range(G)]) for e in range(4)])
    tm_all, Wc = eng.combine(d, G, p)
    return np.asarray(tm_all), int(sum(Ws)) + int(Wc)


def transform_sample_axis(q: Optional[np.ndarray], T, XI, M: int, eng: SampleAxisEngine, dst: int = 0, group=None,
                          data_device=None):
    """Root holds the signal q [D] complex128 (others pass None); D must be a multiple of the world
    size (the root's product needs factors of equal degree).  Scatter the blocks, rank g reduces its
    block to one matrix, one gather, the root multiplies the G matrices in the reference's order and
    evaluates.  Returns [rho | a | b] (3*M) on the root, None elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    meta = [int(q.shape[0])] if rank == dst else [None]
    dist.broadcast_object_list(meta, src=dst, group=group)
    D = meta[0]
    if D % world or D // world < 1:
        raise ValueError("sample-axis sharding needs D to be a multiple of the world size")
    Db = D // world
    eps_t = (T[1] - T[0]) / (D - 1)          # the step of the WHOLE grid, not of the block
    wdev = wire_device(group)
    recv = torch.zeros((Db, 2), dtype=torch.float64, device=wdev)
    chunks = None
    if rank == dst:
        v = torch.from_numpy(np.ascontiguousarray(q, dtype=np.complex128).view(np.float64).reshape(D, 2)).to(wdev)
        chunks = [v[r * Db:(r + 1) * Db].contiguous() for r in range(world)]
    dist.scatter(recv, chunks, src=dst, group=group)
    d = Db * eng.deg0
    if data_device is None:
        tm, W = eng.subtree(recv.cpu().numpy().reshape(2 * Db).view(np.complex128), eps_t)
        tm = np.asarray(tm, np.complex128)
        if tm.shape != (4, d + 1):
            raise ValueError("sub-tree returned shape %r, expected (4, %d)" % (tm.shape, d + 1))
        # one row per rank: the four coefficient arrays, then the exponent
        row = torch.from_numpy(np.concatenate([tm.reshape(-1).view(np.float64), [float(W)]])).reshape(1, -1)
    else:
        # block and block matrix stay on the device; they enter the gather as device tensors (nccl)
        tm, W = eng.subtree(torch.view_as_complex(recv.to(torch.device(data_device)).contiguous()), eps_t)
        if tuple(tm.shape) != (4, d + 1):
            raise ValueError("sub-tree returned shape %r, expected (4, %d)" % (tuple(tm.shape), d + 1))
        row = torch.cat([torch.view_as_real(tm.contiguous()).reshape(-1),
                         torch.tensor([float(W)], dtype=torch.float64, device=tm.device)]).reshape(1, -1)
    full = gather_shards(row, world, dst=dst, group=group)
    if full is None:
        return None
    if data_device is not None and getattr(eng, "device_root", None) is not None and full.is_cuda:
        # block matrices arrived device-to-device: product and evaluation on the root's GPU, only the result leaves it
        return eng.device_root(full, d, D, T, XI, M).cpu().numpy()
    rows = full.cpu().numpy()
    Ws = [int(rows[g, -1]) for g in range(world)]
    tms = [np.ascontiguousarray(rows[g, :-1]).view(np.complex128).reshape(4, d + 1) for g in range(world)]
    tm_all, W_all = combine_block_matrices(tms, Ws, eng)
    return contspec_from_transfer_matrix(np.asarray(tm_all), W_all, D, T, XI, M, eng)
