#!/usr/bin/env python3
"""bench.py -- headline benchmark of the fnft_nsev continuous-spectrum hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one full pass of the hot path (per-sample AKNS coefficients -> 2x2 polynomial product
tree -> chirp z-transform -> a, b, rho) over one synthetic signal of BASELINE.json's configs[1]:
D = M = 2^20 complex128 samples, q = 3.2i*sech(t), T = [-25, 25], XI = [-7/5, 8/5],
2SPLIT2_MODAL, contspec_type BOTH, normalisation on.  Inputs and outputs are resident in HBM.

`--workload cfg3` runs BASELINE.json configs[2] instead (512 independent signals of D = M = 2^16 over 8
GPUs = 64 signals per GPU in one batched plan, XI = [-4, 4]); the default is the headline configs[1].

With N ranks every rank transforms its own signal (weak scaling, no data-path collective: the signals
are independent).  One RCCL gather per step collects the result shards on rank 0
after EVERY step inside the timed region (`--gather step`, default: K steps = K gathered result sets,
each gather overlapped with the next step's compute); `--gather job` gathers only the last step's shards,
`--gather none` leaves the results sharded.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def bytes_tree(D, deg0):
    """Algorithmic bytes of the product tree, SURVEY.md section 8d:
    sum over levels of 64*[n(d+1) + (n/2)(2d+1)] = 128*D*deg0*log2(D) + 192*(D-1)."""
    return 128 * D * deg0 * int(round(math.log2(D))) + 192 * (D - 1)


def level_bytes(D, deg0, lev):
    """One level of SURVEY 8(d)'s model: 4 n input polynomials of degree d read, 4 n/2 of degree 2d written."""
    n, d = D >> lev, deg0 << lev
    return 64 * (n * (d + 1) + (n // 2) * (2 * d + 1))


def launch_breakdown(plan, run_once, reps, B, D, deg0, sync=None):
    """Per-launch HIP-event times of the tree kernels (event pair around every launch, on the launch
    stream), grouped into the stages of the tree with the algorithmic bytes of the levels each stage
    covers.  A split level is its row kernel (KMid) plus the column kernels up to the next row kernel;
    the leaf stage takes the levels the other launches do not account for."""
    import re
    if sync is None:
        import torch
        sync = torch.cuda.synchronize
    acc = None
    for _ in range(reps):
        plan.set_launch_timing(True)
        run_once()
        sync()
        lt = plan.launch_times()
        plan.set_launch_timing(False)
        if acc is None:
            acc = [[n, 0.0] for n, _ in lt]
        if len(lt) != len(acc):
            return None
        for a, (_, ms) in zip(acc, lt):
            a[1] += ms / reps
    launch_breakdown.last_launches = [[n, round(ms * 1e3, 2)] for n, ms in acc]   # every launch, in order: [kernel, us]
    tree = [(n, ms) for n, ms in acc if not n.startswith(("KChirp", "KResample", "KExportTm"))]
    groups = []   # [kind, [names], us, levels]
    for n, ms in tree:
        base = n.split("<")[0]
        if base in ("KCoeffs", "KCoeffsProg", "KLeaf", "KLeafMulti"):
            kind, lv = "leaf", 0
        elif base == "KMulti":
            kind, lv = "fused levels " + n, int(re.findall(r"\d+", n)[-1])
        elif base in ("KPairSchool", "KPairFft", "KRPairSchool", "KRPair", "KRPair4"):
            kind, lv = "single-launch levels", 1
        elif base in ("KMid", "KMidSym", "KMidGen"):
            kind, lv = "split levels", 1
        elif base in ("KColFwd", "KRColFwd", "KR3ColFwd"):
            kind, lv = "split levels", 0
        elif base == "KRealCheck":
            kind, lv = "leaf", 0
        else:   # KColBridge*, KColInv, KFinalizeScales: belong to the level that is open
            kind, lv = (groups[-1][0] if groups else "leaf"), 0
        if groups and groups[-1][0] == kind:
            g = groups[-1]
        else:
            g = [kind, {}, 0.0, 0]
            groups.append(g)
        g[1][n] = g[1].get(n, 0) + 1
        g[2] += ms * 1e3
        g[3] += lv
    total_levels = int(math.log2(D))
    rest = total_levels - sum(g[3] for g in groups)
    for g in groups:
        if g[0] == "leaf":
            g[3] += rest
    out, lev = [], 0
    for kind, names, us, lv in groups:
        by = B * sum(level_bytes(D, deg0, l) for l in range(lev, lev + lv))
        gbs = by / (us * 1e-6) / 1e9 if us > 0 else 0.0
        out.append({"stage": kind, "launches": names, "levels": ([lev, lev + lv - 1] if lv else None), "us": round(us, 1),
                    "algorithmic_bytes": by, "model_GB/s": round(gbs, 1), "model_frac": round(gbs / 8000.0, 4)})
        lev += lv
    # chirp z-transform + epilogue (src/private/fnft__poly_chirpz.c:52-95, src/fnft_nsev.c:837-884): not part of the tree
    # figure; its byte model is one pass of each of its three kernels over two polynomials -- N = D*deg0 + 1 coefficients
    # in, the two length-L work arrays written and read by the column and row steps (the filter's spectrum read once), M
    # results of nout values out
    ch = [(n, ms) for n, ms in acc if n.startswith("KChirp")]
    if ch:
        names = {}
        for n, _ in ch:
            names[n] = names.get(n, 0) + 1
        us = sum(ms for _, ms in ch) * 1e3
        N, M = D * deg0 + 1, D
        L = 1 << int(math.ceil(math.log2(N + M - 1)))
        nout = 1 if any("true>" in n and "ColInv" in n for n in names) else 3
        by = B * (2 * 16 * N + 2 * 16 * L + (2 * 16 * L + 16 * L + 2 * 16 * L) + 2 * 16 * L + nout * 16 * M)
        gbs = by / (us * 1e-6) / 1e9 if us > 0 else 0.0
        out.append({"stage": "chirp-z + epilogue", "launches": names, "levels": None, "us": round(us, 1),
                    "algorithmic_bytes": by, "model_GB/s": round(gbs, 1), "model_frac": round(gbs / 8000.0, 4)})
    return out


def bench_cfg4(args, rank, world, local_rank):
    """BASELINE.json configs[3]: continuous spectrum AND bound states (default options: 2SPLIT4B, SUBSAMPLE_AND_REFINE, 10
    Newton steps, norming constants) of the sech pulse at D = M = 2^20, through the drop-in fnft_nsev with HOST pointers
    (the discrete spectrum has no device-resident entry).  One step = one call; every rank runs its own signal."""
    import torch
    import torch.distributed as dist
    from fnft_amd import capi
    import signals as S
    # the process group (world > 1) was initialised by main()
    D = M = 1 << args.log2D
    T, XI = [-25.0, 25.0], [-7.0 / 5.0, 8.0 / 5.0]
    q = S.sech_focusing(D, amp=3.2 - 0.01 * rank)
    steps, warm = max(1, min(args.steps, 10)), max(1, min(args.warmup, 2))

    # the caller's arrays, allocated once as in the reference's example (examples/fnft_nsev_example.c)
    capK = 2 * D   # fnft_nsev_max_K for 2SPLIT4B
    bufs = {"bs": np.zeros(capK, np.complex128), "nc": np.zeros(2 * capK, np.complex128), "cs": np.zeros(3 * M, np.complex128)}

    def call():
        out = capi.fnft_nsev_ds(q, T, discretization="2SPLIT4B", M=M, XI=XI, bufs=bufs)
        if out[0] != 0:
            raise RuntimeError("fnft_nsev rc=%d: %s" % (out[0], capi.last_error()))
        return out

    for _ in range(warm):
        out = call()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    stages = {}
    for _ in range(steps):
        out = call()
        for name, ms in capi.discspec_stages():
            stages[name] = stages.get(name, 0.0) + ms / steps
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall_ms = (time.perf_counter() - t0) * 1e3
    tt = torch.tensor([wall_ms], dtype=torch.float64, device="cpu" if (world == 1 or args.rehearse_gloo) else "cuda")
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ms_per_step = float(tt.item()) / steps
    if rank == 0:
        rc, bs, nc, res, cs = out
        exact = np.array([0.7j, 1.7j, 2.7j])
        # tree of the continuous-spectrum part (2SPLIT4B, deg 2) timed on a device plan, as in the headline workload
        plan = capi.Plan(D, M, batch=1, discretization="2SPLIT4B", device=local_rank)
        plan.set_timing(True)
        dq = torch.from_numpy(q).cuda()
        do = torch.zeros(3 * M, dtype=torch.complex128, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        tms = []
        for _ in range(8):
            plan.contspec_device(dq.data_ptr(), do.data_ptr(), T, XI, 1, "BOTH", 1, st)
            torch.cuda.synchronize()
            tms.append(plan.last_ms(0))
        t_tree = float(np.median(tms[2:]))
        bt = bytes_tree(D, 2)
        # counter traffic of the tree launches: the tree of this workload is the 2SPLIT4B tree of one 2^log2D signal --
        # the launches profiles/tree_traffic.json holds under that key (same build only)
        traffic4, bid4 = None, None
        try:
            from fnft_amd import build as fa_build
            bid4 = fa_build.build_id()
            with open(os.path.join(ROOT, "profiles", "tree_traffic.json")) as f:
                ent4 = json.load(f).get("cfg2/D=2^%d/2SPLIT4B/B=1" % args.log2D) or {}
            if ent4.get("build_id") == bid4:
                traffic4 = ent4.get("tree_hbm_bytes_per_step")
        except OSError:
            pass
        # CPU baseline of THIS workload on a bounded sample: the oracle (contspec + bound states with the default
        # options: numpy root finder on the subsampled signal, sequential Newton refinement and norming constants)
        # on one D = M = 2^16 signal, 1 thread -- the full size would take minutes
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            from oracle.oracle import load_oracle
            orc = load_oracle()
            Dc = 1 << min(args.log2D, 16)
            qc = S.sech_focusing(Dc, amp=3.2)
            tc0 = time.perf_counter()
            oc = orc.fnft_nsev_ds(qc, T, "2SPLIT4B")
            rcc, csc = orc.fnft_nsev(qc, T, Dc, XI, kappa=1, disc="2SPLIT4B", cstype="BOTH")
            tc = time.perf_counter() - tc0
            bsc = np.asarray(oc[1])
            cpu = {"value": round(Dc / tc / 1e6, 5), "unit": "Msamples/s", "cores": 1, "kind": "port",
                   "sample": "one fnft_nsev call of the oracle at D = M = 2^%d (contspec + bound states + norming constants, "
                             "default options), %.1f s" % (int(math.log2(Dc)), tc),
                   "bound_states": int(bsc.size),
                   "bound_state_error_vs_exact": (float(np.abs(np.sort_complex(bsc) - np.sort_complex(exact)).max())
                                                  if bsc.size == 3 else None)}
        line = {
            "metric": "Msamples/s fnft_nsev contspec + bound states (D=2^%d fp64)" % args.log2D,
            "value": round(world * D / (ms_per_step * 1e-3) / 1e6, 2), "unit": "Msamples/s", "n_gpus": world, "steps": steps,
            "warmup": warm, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "fnft_nsev D=M=2^%d contspec (a,b + reflection) + bound states, norming constants "
                                   "(default options: 2SPLIT4B, SUBSAMPLE_AND_REFINE), host-pointer drop-in call, 1 signal per GPU"
                                   % args.log2D, "gather": "n/a"},
            "roofline": {"bound": "hbm", "kernel": "poly_fmult2x2 tree of the continuous-spectrum part (2SPLIT4B)",
                         "achieved": round(bt / (t_tree * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(bt / (t_tree * 1e-3) / 1e9 / 8000.0, 4), "traffic": traffic4,
                         "algorithmic_bytes": bt, "tree_ms": round(t_tree, 4),
                         "discspec_stages_ms": {k: round(v, 3) for k, v in stages.items()},
                         "bound_states": int(bs.size),
                         "bound_state_error": (float(max(np.abs(bs[:, None] - exact[None, :]).min(axis=0).max(),
                                                          np.abs(bs[:, None] - exact[None, :]).min(axis=1).max()))
                                               if bs.size else None)},
            "cpu_baseline": cpu, "build_id": bid4,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


PEEL_CYCLES_PER_STEP = 292     # body_peel_leaf, wave 0, steady-state loop (see bench_inverse)
PEEL_CLOCK_HZ = 2.4e9


def bench_inverse(args, rank, world, local_rank):
    """fnft_nsev_inverse through the host-pointer drop-in: b(xi) of a sech pulse below the soliton
    threshold -> q (the reference's b_of_xi test at D = 2^log2D, default 2^16), 2SPLIT2_MODAL, M = D.  One call per step;
    every rank inverts its own spectrum (weak scaling, nothing to gather but D samples)."""
    import torch
    import torch.distributed as dist
    from fnft_amd import capi
    import signals as S

    capi.load()
    log2D = args.log2D if args.log2D != 20 else 16
    D = 1 << log2D
    T = [-25.0, 25.0]
    A, t0 = 0.45 - 0.001 * rank, 1.2
    rc, XI = capi.nsev_inverse_XI(D, T, D, "2SPLIT2_MODAL")
    xi = XI[0] + (XI[1] - XI[0]) / (D - 1) * np.arange(D)
    with np.errstate(over="ignore"):
        cs0 = 1j * np.exp(-2j * xi * t0) * np.sin(np.pi * A) / np.cosh(np.pi * xi)
    exact = 1j * A / np.cosh(S.tgrid(T, D) - t0)
    opts = {"discretization": "2SPLIT2_MODAL", "contspec_type": "B_OF_XI"}

    def call():
        rc, q = capi.fnft_nsev_inverse(D, cs0.copy(), XI, None, None, D, T, 1, opts)
        if rc != 0:
            raise RuntimeError("fnft_nsev_inverse rc=%d: %s" % (rc, capi.last_error()))
        return q
    steps, warm = max(1, min(args.steps, 10)), max(1, min(args.warmup, 2))
    for _ in range(warm):
        q = call()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0w = time.perf_counter()
    for _ in range(steps):
        q = call()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall_ms = (time.perf_counter() - t0w) * 1e3
    tt = torch.tensor([wall_ms], dtype=torch.float64, device="cpu" if (world == 1 or args.rehearse_gloo) else "cuda")
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ms_per_step = float(tt.item()) / steps
    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            from oracle import inverse as INV
            Dc = min(D, 1 << 14)            # bounded sample: the numpy restatement needs 0.6 s at 2^14, 2.5 s at 2^16
            rcx, XIc = capi.nsev_inverse_XI(Dc, T, Dc, "2SPLIT2_MODAL")
            xic = XIc[0] + (XIc[1] - XIc[0]) / (Dc - 1) * np.arange(Dc)
            with np.errstate(over="ignore"):
                csc = 1j * np.exp(-2j * xic * t0) * np.sin(np.pi * A) / np.cosh(np.pi * xic)
            tc0 = time.perf_counter()
            nrep = 8
            for _ in range(nrep):
                rco, qo = INV.fnft_nsev_inverse(Dc, csc.copy(), XIc, None, None, Dc, T, 1, opts)
            tc = time.perf_counter() - tc0
            rcg, qg = capi.fnft_nsev_inverse(Dc, csc.copy(), XIc, None, None, Dc, T, 1, opts)
            cpu = {"value": round(nrep * Dc / tc / 1e6, 5), "unit": "Msamples/s", "cores": 1, "kind": "port",
                   "sample": "%d calls, D=M=2^%d, b(xi) -> q, oracle/inverse.py (numpy, one thread), %.1f s"
                             % (nrep, int(math.log2(Dc)), tc),
                   "gpu_vs_cpu_rel_l1": {"q": float(S.rel_err(qg, qo))}}
        line = {
            "metric": "Msamples/s fnft_nsev_inverse b(xi) -> q (D=2^%d fp64)" % log2D,
            "value": round(world * D / (ms_per_step * 1e-3) / 1e6, 3), "unit": "Msamples/s", "n_gpus": world,
            "steps": steps, "warmup": warm, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "fnft_nsev_inverse D=M=2^%d, contspec_type B_OF_XI (spectral factorization at "
                                   "oversampling 8 + layer peeling), 2SPLIT2_MODAL, host-pointer drop-in call, 1 spectrum per GPU"
                                   % log2D, "gather": "n/a"},
            # Layer peeling is a serial recursion: sample n cannot be formed before sample n+1 has been divided out.  The
            # model is its floor on one wave: D steps of the leaf kernel's first-column chain, each 49 fp64 vector
            # instructions + 16 cross-lane moves (v_readlane / DPP) + 8 register moves = 73 vector instructions of 4
            # clocks (16 lanes per SIMD) = 292 clocks at 2.4 GHz (counted in the ISA of body_peel_leaf's steady-state
            # loop); the pair products above the leaves and the spectral factorization are on top of it, so frac =
            # model / measured counts them as loss.
            "roofline": {"bound": "latency", "kernel": "layer peeling (leaf kernel chain + pair products), DESIGN.md 5",
                         "achieved": round(ms_per_step, 3), "peak": round(D * PEEL_CYCLES_PER_STEP / PEEL_CLOCK_HZ * 1e3, 3),
                         "unit": "ms (peak = modelled floor of the serial chain)",
                         "frac": round(D * PEEL_CYCLES_PER_STEP / PEEL_CLOCK_HZ * 1e3 / ms_per_step, 4), "traffic": None,
                         "latency_model": {"serial_steps": D, "cycles_per_step": PEEL_CYCLES_PER_STEP,
                                           "clock_GHz": PEEL_CLOCK_HZ / 1e9,
                                           "model_ms": round(D * PEEL_CYCLES_PER_STEP / PEEL_CLOCK_HZ * 1e3, 3)},
                         "rel_err_vs_exact_signal": float(S.rel_err(q, exact))},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def measure_cfg3(args, rank, world, local_rank, barrier):
    """BASELINE.json configs[2] on the ranks of this job: every rank transforms its 64 signals of D = M = 2^16 (one
    batched plan) and the results (a, b and the reflection coefficient: 64 x 3 x 2^16 complex128 = 192 MiB per rank)
    meet on rank 0 in one RCCL gather per batch, double-buffered so that the gather of batch i overlaps the transform of
    batch i + 1.  Every rank runs this (the gathers are collectives); rank 0 reports it inside the one JSON line."""
    import torch
    import torch.distributed as dist
    from fnft_amd import capi, sharding
    import signals as S

    D = M = 1 << 16
    B = 64
    T, XI = [-25.0, 25.0], [-4.0, 4.0]
    q = np.stack([S.batch_signal(rank * B + k, D, T) for k in range(B)])
    plan = capi.Plan(D, M, batch=B, discretization="2SPLIT2_MODAL", device=local_rank)
    dq = torch.from_numpy(q).cuda()
    outs = [torch.zeros(B * 3 * M, dtype=torch.complex128, device="cuda") for _ in range(2)]
    g = sharding.ShardGather((B * 3 * M, 2), torch.float64, dst=0, depth=2)
    stream = torch.cuda.current_stream().cuda_stream
    steps = max(2, min(args.steps, 10))
    dev = "cpu" if args.rehearse_gloo else "cuda"

    def batch(i, gather):
        if gather:
            g.reserve()
        rc = plan.contspec_device(dq.data_ptr(), outs[i % 2].data_ptr(), T, XI, kappa=1, contspec_type="BOTH",
                                  normalization_flag=1, stream=stream)
        if rc != 0:
            raise RuntimeError("cfg3 batch rc=%d: %s" % (rc, capi.last_error()))
        if gather:
            if args.rehearse_gloo:
                torch.cuda.synchronize()
            g.start(torch.view_as_real(outs[i % 2]), i)

    def timed(gather):
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            batch(i, gather)
        if gather:
            g.wait()
        barrier()
        t = torch.tensor([(time.perf_counter() - t0) * 1e3], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / steps

    for i in range(2):
        batch(i, True)
    g.wait()
    ms = timed(True)
    comp = timed(False)
    rc = plan.finish(stream)
    plan.close()
    if rc != 0:
        raise RuntimeError("cfg3 device status rc=%d: %s" % (rc, capi.last_error()))
    return {"workload": "configs[2]: %d signals of D=M=2^16 (64 per GPU), 2SPLIT2_MODAL, a, b + reflection; one gather of "
                        "192 MiB per rank to rank 0 per batch, overlapped with the next batch" % (world * B),
            "value": round(world * B * D / (ms * 1e-3) / 1e6, 2), "unit": "Msamples/s", "batches": steps,
            "ms_per_batch": round(ms, 4), "compute_only_ms_per_batch": round(comp, 4),
            "gather_overhead_ms_per_batch": round(max(0.0, ms - comp), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2D", type=int, default=20)
    ap.add_argument("--disc", default="2SPLIT2_MODAL")
    ap.add_argument("--workload", choices=("cfg2", "cfg3", "cfg4", "cfg5", "inverse"), default=None,
                    help="cfg2: one signal D=M=2^20 per GPU (headline); cfg3: BASELINE.json configs[2], "
                         "64 of the 512 signals D=M=2^16 per GPU; cfg4: configs[3], contspec + bound states at "
                         "D=M=2^20 through the drop-in fnft_nsev (host pointers, default options); cfg5: configs[4], "
                         "fnft_kdvv D=M=2^18 2SPLIT8B")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ramp-steps", type=int, default=48,
                    help="untimed steps before the timed region in total (warmup included): the clock ramp after an idle GPU")
    ap.add_argument("--stage-events", type=int, default=1,
                    help="record the plan's tree / chirp HIP events on every N-th step of the timed region (0: none there; "
                         "the roofline pass after it always records them)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the two-transforms-in-flight rate")
    ap.add_argument("--no-host-call", action="store_true",
                    help="skip the host-pointer drop-in timing (PMC passes: its single-signal launches share kernel names with the batch)")
    ap.add_argument("--gather", choices=("job", "step", "none"), default="step",
                    help="N>1: 'step' (default) = the result shards of EVERY step are gathered on rank 0 inside the "
                         "timed region (RCCL, overlapped with the next step's compute); 'job' = only the last "
                         "step's shards are gathered; 'none' = results stay sharded")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--gather-parts", choices=("rho", "ab", "both"), default="both",
                    help="N>1: which part of every signal's result the per-step gather moves to rank 0: the reflection "
                         "coefficient (M values), a and b (2M), or all three (3M, default).  At N = 8 and D = 2^20 'both' asks "
                         "rank 0 to take in 7 x 48 MiB per 0.73 ms step -- about the one-directional rate of its xGMI links; "
                         "'rho' (the library's default contspec_type) is a third of that")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N>1 control-flow rehearsal on a box with fewer GPUs than ranks: gloo backend, "
                         "host-staged gather, ranks share the visible GPUs (not a measurement)")
    args = ap.parse_args()
    workload_given = args.workload is not None
    if args.workload is None:
        args.workload = "cfg2"

    import torch
    import torch.distributed as dist
    from fnft_amd import capi
    import signals as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if args.rehearse_gloo:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if args.workload == "cfg4":
        return bench_cfg4(args, rank, world, local_rank)
    if args.workload == "inverse":
        return bench_inverse(args, rank, world, local_rank)
    if args.no_gather:
        args.gather = "none"
    cfg3 = args.workload == "cfg3"
    cfg5 = args.workload == "cfg5"
    if cfg3:
        args.log2D = 16
    if cfg5:
        args.log2D = 18
        args.disc = "2SPLIT8B"
    D = M = 1 << args.log2D
    B = 64 if cfg3 else 1   # signals per GPU
    deg0 = {"2SPLIT2_MODAL": 1, "2SPLIT4B": 2, "2SPLIT8B": 12}.get(args.disc, 1)
    T = [-25.0, 25.0]
    XI = [-4.0, 4.0] if cfg3 else [-7.0 / 5.0, 8.0 / 5.0]
    if cfg5:
        # SURVEY 8d cfg 5: u = 3.2 sech^2(t), T = [-16, 15], XI = [-71/20, 79/20]
        T, XI = [-16.0, 15.0], [-71.0 / 20.0, 79.0 / 20.0]
        q_host = S.kdvv_sech(D, T) * (1.0 - 0.001 * rank)
    elif cfg3:
        # SURVEY 8d cfg 3: signal k = A*sech(t - tau)*exp(i*w*t), (A, tau, w) from splitmix64(0x5EED0000+k)
        q_host = np.stack([S.batch_signal(rank * B + k, D, T) for k in range(B)])
    else:
        # each rank gets its own (slightly different) signal so that no rank can reuse another's work
        q_host = S.sech_focusing(D, amp=3.2 - 0.01 * rank)

    if cfg5:
        plan = capi.KdvvPlan(D, M, batch=B, discretization=args.disc, device=local_rank)
    else:
        plan = capi.Plan(D, M, batch=B, discretization=args.disc, device=local_rank)
    plan.set_timing(True)
    dq = torch.from_numpy(q_host).cuda()
    nout = 1 if cfg5 else 3   # fnft_kdvv returns the reflection coefficient only
    outs = [torch.zeros(B * nout * M, dtype=torch.complex128, device="cuda") for _ in range(2)]
    # N > 1: the result shards meet on rank 0 through the product's sharding module (fnft_amd/sharding.py):
    # device tensors into RCCL under nccl; under the gloo rehearsal the same calls stage through the host
    gatherer = None
    part = (0, nout) if nout == 1 else {"rho": (0, 1), "ab": (1, 3), "both": (0, 3)}[args.gather_parts]
    npart = part[1] - part[0]
    if world > 1 and args.gather != "none":
        from fnft_amd import sharding
        gatherer = sharding.ShardGather((B * npart * M, 2), torch.float64, dst=0, depth=2)
    stream = torch.cuda.current_stream().cuda_stream

    def shard_of(buf):
        """the part of a result buffer ([signal][rho | a | b][M]) the gather moves, as a real [n, 2] tensor"""
        if npart == nout:
            return torch.view_as_real(buf)
        sel = buf.view(B, nout, M)[:, part[0]:part[1], :]
        return torch.view_as_real(sel.contiguous().view(-1))

    def transform(out_ptr):
        if cfg5:
            return plan.contspec_device(dq.data_ptr(), out_ptr, T, XI, stream=stream)
        return plan.contspec_device(dq.data_ptr(), out_ptr, T, XI, kappa=1, contspec_type="BOTH",
                                    normalization_flag=1, stream=stream)

    def one_step(i, pending, last=False):
        buf = outs[i % 2]
        plan.set_timing(args.stage_events > 0 and (last or i % args.stage_events == 0))
        if gatherer is not None:
            gatherer.reserve()   # the gather that read this buffer two steps ago has finished
        rc = transform(buf.data_ptr())
        if rc != 0:
            raise RuntimeError("fnft_amd_nsev_contspec_device rc=%d: %s" % (rc, capi.last_error()))
        if gatherer is not None and (args.gather == "step" or (args.gather == "job" and last)):
            if args.rehearse_gloo:
                torch.cuda.synchronize()   # gloo copies through the host: the kernels must have finished
            # nccl: the collective is ordered behind this step's kernels (same current stream)
            gatherer.start(shard_of(buf), i)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pending = []
    for i in range(args.warmup):
        one_step(i, pending, last=(i == args.warmup - 1))
    # The GPU's clocks ramp over the first ~15 back-to-back transforms after an idle period (tests/gpu_debug/step_trend.py:
    # 0.74 -> 0.70 ms per step), longer than the W warmup steps the command line asks for.  The timed region measures the
    # transform, not the ramp: untimed steps follow the warmup until about 30 ms of back-to-back work have been queued
    # (no result of theirs is used; `config.untimed_ramp_steps` says how many).
    ramp_steps = max(0, args.ramp_steps - args.warmup)
    for i in range(ramp_steps):
        one_step(args.warmup + i, pending, last=(i == ramp_steps - 1))
    if gatherer is not None:
        gatherer.wait()
    pending = []
    rc = plan.finish(stream)
    if rc != 0:
        raise RuntimeError("device status rc=%d: %s" % (rc, capi.last_error()))

    # ---- timed region: exactly K steps between barrier + synchronize pairs -----------------
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        one_step(i, pending, last=(i == args.steps - 1))
    if gatherer is not None:
        gatherer.wait()
    ev1.record()
    barrier()
    t1 = time.perf_counter()
    wall_ms = (t1 - t0) * 1e3
    ev_ms = ev0.elapsed_time(ev1)
    # HIP-event timers of the last step, recorded on the launch stream inside the plan
    tree_ms = plan.last_ms(0)
    chirp_ms = plan.last_ms(1)
    rc = plan.finish(stream)
    if rc != 0:
        raise RuntimeError("device status rc=%d: %s" % (rc, capi.last_error()))

    t_all = torch.tensor([wall_ms], dtype=torch.float64, device="cpu" if args.rehearse_gloo else "cuda")
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
    wall_ms_max = float(t_all.item())
    ms_per_step = wall_ms_max / args.steps
    value = world * B * D / (ms_per_step * 1e-3) / 1e6  # Msamples/s, whole job

    # ---- N > 1: what the gather costs, and BASELINE.json configs[2] in the same run ------------------------
    multi = {}
    if world > 1 and gatherer is not None and args.gather == "step":
        # the same K steps with the results left on their GPUs: the difference is the collective's share of a step
        barrier()
        tc0 = time.perf_counter()
        for i in range(args.steps):
            if transform(outs[i % 2].data_ptr()) != 0:
                raise RuntimeError("compute-only step: %s" % capi.last_error())
        barrier()
        t_c = torch.tensor([(time.perf_counter() - tc0) * 1e3], dtype=torch.float64, device="cpu" if args.rehearse_gloo else "cuda")
        dist.all_reduce(t_c, op=dist.ReduceOp.MAX)
        comp = float(t_c.item()) / args.steps
        multi = {"gather_parts": args.gather_parts if nout == 3 else "rho",
                 "gather_bytes_per_rank_per_step": B * npart * M * 16,
                 "compute_only_ms_per_step": round(comp, 4),
                 "gather_overhead_ms_per_step": round(max(0.0, ms_per_step - comp), 4)}
        if plan.finish(stream) != 0:
            raise RuntimeError("device status: %s" % capi.last_error())
    if world > 1 and not workload_given and args.gather != "none":
        # the driver's scaling command passes no --workload: next to the headline (one 2^20 signal per GPU) it also gets
        # configs[2] -- 64 signals of 2^16 per GPU, one gather of every GPU's results per batch -- as "cfg3"
        multi["cfg3"] = measure_cfg3(args, rank, world, local_rank, barrier)

    # ---- tree-only timing with HIP events over many passes (roofline) -----------------------
    roof = None
    cpu = None
    if rank == 0:
        reps = max(5, args.steps)
        tms = []
        plan.set_timing(True)
        for i in range(reps):
            transform(outs[0].data_ptr())
            torch.cuda.synchronize()
            tms.append(plan.last_ms(0))
        t_tree = float(np.median(tms))
        # KdV has no NSE symmetry and r = -1: the general 4-entry tree, same byte model
        bt = B * bytes_tree(D, deg0)
        achieved = bt / (t_tree * 1e-3) / 1e9
        # HBM bytes of the tree launches of one step from the PMC passes of this build (rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE in separate runs of this command, FETCH doubled as the microarch guide
        # prescribes; profiles/traffic_from_pmc.py writes the file).  Counters cannot be read from inside
        # the process, so the number is the committed one for this workload, or null.
        traffic = None
        traffic_note = None
        tpath = os.path.join(ROOT, "profiles", "tree_traffic.json")
        wkey = "%s/D=2^%d/%s/B=%d" % (args.workload, args.log2D, args.disc, B)
        from fnft_amd import build as fa_build
        bid = fa_build.build_id()
        if os.path.exists(tpath):
            with open(tpath) as f:
                ent = json.load(f).get(wkey) or {}
            # a counter figure is only reported next to a timing of the build it was measured on
            if ent.get("build_id") == bid:
                traffic = ent.get("tree_hbm_bytes_per_step")
            elif ent:
                traffic_note = "profiles/tree_traffic.json holds counters of build %s, this is %s" % (ent.get("build_id"), bid)
        stages = launch_breakdown(plan, lambda: transform(outs[0].data_ptr()), reps, B, D, deg0)
        roof = {"bound": "hbm", "kernel": "poly_fmult2x2 tree (coefficients + all level launches of one transform)",
                "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                "algorithmic_bytes": bt, "tree_ms": round(t_tree, 4),
                "chirpz_epilogue_ms": round(float(chirp_ms), 4), "stages": stages,
                "launches_us": getattr(launch_breakdown, "last_launches", None)}
        if traffic is not None:   # what the counters say the tree moved, as a rate (achieved / frac are the byte MODEL)
            roof["counter_GBps"] = round(traffic / (t_tree * 1e-3) / 1e9, 1)
            roof["counter_frac"] = round(traffic / (t_tree * 1e-3) / 1e9 / 8000.0, 4)
        if traffic_note:
            roof["traffic_note"] = traffic_note
        if world == 1 and not cfg5:
            # (i) cold spectral grid: `value` reuses the spectrum of the chirp filter the plan keeps between
            # calls on the same (T, XI) grids (plan data, like a twiddle table); a call on a NEW grid forms it
            # again as fnft__poly_chirpz.c:76-84 does on every call.  Every step here gets its own XI.
            torch.cuda.synchronize()
            tcg0 = time.perf_counter()
            for i in range(args.steps):
                xi_i = [XI[0], XI[1] * (1.0 + 1e-9 * (i + 1))]
                rc = plan.contspec_device(dq.data_ptr(), outs[0].data_ptr(), T, xi_i, kappa=1, contspec_type="BOTH",
                                          normalization_flag=1, stream=stream)
                if rc != 0:
                    raise RuntimeError("cold-grid step rc=%d: %s" % (rc, capi.last_error()))
            torch.cuda.synchronize()
            roof["cold_grid_ms_per_step"] = round((time.perf_counter() - tcg0) * 1e3 / args.steps, 4)
            # (ii) the drop-in itself: fnft_nsev() with HOST pointers (PCIe both ways, plan from the cache)
            # The caller owns q and contspec and reuses them between calls, as the reference's callers do
            # (examples/fnft_nsev_example.c allocates once): host_call_ms is such a call; host_call_fresh_ms is a call
            # whose result array was just allocated (its pages are first touched by the device-to-host copy).
            hq = q_host[0] if cfg3 else q_host
            hout = np.zeros(3 * M, np.complex128)
            th, thf = [], []
            for i in range(0 if args.no_host_call else 6):
                th0 = time.perf_counter()
                rch, _ = capi.fnft_nsev(hq, T, M, XI, kappa=1, discretization=args.disc, contspec_type="BOTH", out=hout)
                th.append((time.perf_counter() - th0) * 1e3)
                if rch != 0:
                    raise RuntimeError("fnft_nsev (host pointers) rc=%d: %s" % (rch, capi.last_error()))
            for i in range(0 if args.no_host_call else 3):
                th0 = time.perf_counter()
                rch, _ = capi.fnft_nsev(hq, T, M, XI, kappa=1, discretization=args.disc, contspec_type="BOTH")
                thf.append((time.perf_counter() - th0) * 1e3)
            if th:
                roof["host_call_fresh_ms"] = round(float(np.median(thf)), 4)
                roof["host_call_ms"] = round(float(np.median(th[1:])), 4)
                roof["host_call_Msamples_per_s"] = round(D / (roof["host_call_ms"] * 1e-3) / 1e6, 1)
        if world == 1 and not args.no_pipelined:
            # (iii) two transforms in flight: a second plan on a second stream, steps alternating between them
            # (successive signals of a receiver; independent work, every step still one whole transform).
            # Informational -- `value` is the one-transform-at-a-time rate above.
            plan2 = (capi.KdvvPlan if cfg5 else capi.Plan)(D, M, batch=B, discretization=args.disc, device=local_rank)
            s2 = torch.cuda.Stream()
            lanes = [(plan, stream, outs[0]), (plan2, s2.cuda_stream, outs[1])]

            def piped(i):
                pl, st, ob = lanes[i % 2]
                if cfg5:
                    return pl.contspec_device(dq.data_ptr(), ob.data_ptr(), T, XI, stream=st)
                return pl.contspec_device(dq.data_ptr(), ob.data_ptr(), T, XI, kappa=1, contspec_type="BOTH",
                                          normalization_flag=1, stream=st)
            for i in range(4):
                piped(i)
            torch.cuda.synchronize()
            tp0 = time.perf_counter()
            for i in range(args.steps):
                if piped(i) != 0:
                    raise RuntimeError("pipelined step: %s" % capi.last_error())
            torch.cuda.synchronize()
            tp = (time.perf_counter() - tp0) * 1e3 / args.steps
            if plan2.finish(s2.cuda_stream) != 0:
                raise RuntimeError("pipelined steps, device status: %s" % capi.last_error())
            roof["two_in_flight"] = {"ms_per_step": round(tp, 4), "Msamples_per_s": round(B * D / (tp * 1e-3) / 1e6, 1)}
            plan2.close()
        if not args.no_cpu_baseline and world == 1:   # the CPU checker is timed at N = 1 only
            from oracle import load_oracle
            orc = load_oracle()
            Dc = min(D, 1 << 20)
            nc = 8 if cfg3 else 1   # bounded sample: the first nc signals of this rank
            qc = q_host[:nc] if cfg3 else S.sech_focusing(Dc)[None, :]
            res = outs[(args.steps - 1) % 2].cpu().numpy().reshape(B, nout, M) if Dc == D else None
            tc = 0.0
            worst = [0.0] * nout
            if cfg5:
                qc = q_host[None, :]
            for k in range(nc):
                tc0 = time.perf_counter()
                if cfg5:
                    rcc, ref = orc.fnft_kdvv(qc[k], T, Dc, XI, args.disc)
                else:
                    rcc, ref = orc.fnft_nsev(qc[k], T, Dc, XI, kappa=1, disc=args.disc, cstype="BOTH")
                tc += time.perf_counter() - tc0
                if res is not None and rcc == 0:
                    for j in range(nout):
                        worst[j] = max(worst[j], float(S.rel_err(res[k, j], ref[j * M:(j + 1) * M])))
            err = dict(zip(("rho", "a", "b")[:nout], worst)) if res is not None else None
            # oracle time / reference time for the same call, measured in the build container where the
            # reference could be run (the reference cannot travel to this box)
            ref_ratio = {"D=2^12 2SPLIT2_MODAL": 1.19, "D=2^12 2SPLIT4B": 1.31, "D=2^16 2SPLIT2_MODAL": 1.20,
                         "D=2^16 2SPLIT4B": 1.04, "D=2^18 2SPLIT2_MODAL": 0.94, "D=2^18 2SPLIT4B": 0.94,
                         "D=2^20 2SPLIT2_MODAL": 0.90}.get("D=2^%d %s" % (int(math.log2(Dc)), args.disc))
            cpu = {"value": round(nc * Dc / tc / 1e6, 5), "unit": "Msamples/s", "cores": 1, "kind": "port",
                   "reference_ratio": ref_ratio,
                   "sample": "%d signal(s), D=M=2^%d, %s, oracle/fnft_oracle.c single thread, %.1f s"
                             % (nc, int(math.log2(Dc)), args.disc, tc),
                   "host_cpus": os.cpu_count(), "gpu_vs_cpu_rel_l1": err}

    if rank == 0:
        line = {
            "metric": "Msamples/s %s contspec (D=2^%d fp64)%s" % ("fnft_kdvv" if cfg5 else "fnft_nsev", args.log2D,
                                                                  " batch" if cfg3 else ""),
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s D=M=2^%d contspec (%s), %s, %d signal%s per GPU%s"
                                   % ("fnft_kdvv" if cfg5 else "fnft_nsev", args.log2D,
                                      "reflection" if cfg5 else "a,b + reflection", args.disc, B, "" if B == 1 else "s",
                                      " (configs[2]: 512 signals over 8 GPUs)" if cfg3 else ""),
                       "gather": (args.gather if world > 1 else "n/a (1 GPU)"),
                       "event_ms_per_step": round(ev_ms / args.steps, 4), "untimed_ramp_steps": ramp_steps},
            "roofline": roof, "cpu_baseline": cpu, "build_id": bid,
        }
        if multi:
            line["multi_gpu"] = multi
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
