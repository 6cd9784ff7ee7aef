"""Probe: does running two independent half-size transforms on two HIP streams, offset in time,
overlap the load / compute / store phases of their kernels?  (one level of the tree at D = 2^20 is
exactly one round of resident workgroups, so within one stream the phases of all workgroups
coincide)"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import signals as S
from fnft_amd import capi

T, XI = [-25.0, 25.0], [-1.4, 1.6]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

D = 1 << 19
M = 1024  # small spectral grid: the tree dominates
q = torch.from_numpy(S.sech_focusing(D)).cuda()
plans = [capi.Plan(D, M, 1, "2SPLIT2_MODAL") for _ in range(2)]
outs = [torch.zeros(3 * M, dtype=torch.complex128, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
def one():
    plans[0].contspec_device(q.data_ptr(), outs[0].data_ptr(), T, XI, 1, "BOTH", 1, streams[0].cuda_stream)
t1 = timeit(one)
print("one D=2^19: %.3f ms" % t1, flush=True)
for cyc in (0, 500, 1000, 1500, 2000, 3000, 5000, 10000, 20000, 50000):
    def sl():
        with torch.cuda.stream(streams[1]):
            torch.cuda._sleep(cyc)
    ts = timeit(sl) if cyc else 0.0
    def two_conc():
        if cyc:
            with torch.cuda.stream(streams[1]):
                torch.cuda._sleep(cyc)
        for i in range(2):
            plans[i].contspec_device(q.data_ptr(), outs[i].data_ptr(), T, XI, 1, "BOTH", 1, streams[i].cuda_stream)
    print("offset cycles %6d (sleep alone %.4f ms): two streams %.3f ms" % (cyc, ts, timeit(two_conc)), flush=True)
