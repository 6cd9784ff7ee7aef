"""Per-step GPU time of the headline transform across one run of K back-to-back steps after an idle barrier (one HIP event
between steps): does the first part of a run differ from the steady state?   python tests/gpu_debug/step_trend.py [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import signals as S
from fnft_amd import capi
capi.load()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
D = M = 1 << 20
T, XI = [-25.0, 25.0], [-1.4, 1.6]
plan = capi.Plan(D, M, batch=1, discretization="2SPLIT2_MODAL", device=0)
dq = torch.from_numpy(S.sech_focusing(D, amp=3.2)).cuda()
out = torch.zeros(3 * M, dtype=torch.complex128, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def step():
    rc = plan.contspec_device(dq.data_ptr(), out.data_ptr(), T, XI, kappa=1, contspec_type="BOTH", normalization_flag=1, stream=st)
    assert rc == 0
for _ in range(5): step()
torch.cuda.synchronize()
for trial in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(K):
        step(); ev[i + 1].record()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    d = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(K)])
    print("trial %d: enqueue %.2f ms, wall %.2f ms for %d steps; step ms: first 5 %s | 5-20 mean %.4f | 20-50 mean %.4f | last 20 mean %.4f"
          % (trial, t_enq * 1e3, wall * 1e3, K, np.round(d[:5], 3), d[5:20].mean(), d[20:50].mean() if K >= 50 else -1, d[-20:].mean()), flush=True)
