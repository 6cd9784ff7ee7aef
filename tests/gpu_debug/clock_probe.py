"""Shader clock and power while a workload runs (rocm-smi sampled from a side thread):
    python tests/gpu_debug/clock_probe.py cfg4|cfg2|idle [seconds]"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
what = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
samples, stop = [], False


def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showuse"], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, text=True, timeout=10).stdout
            keep = [l.strip() for l in o.splitlines() if any(k in l for k in ("sclk", "Power", "GPU use", "fclk", "mclk"))]
            samples.append((time.time(), keep))
        except Exception as e:   # noqa: BLE001
            samples.append((time.time(), [repr(e)]))
        time.sleep(0.5)


th = threading.Thread(target=sampler, daemon=True)
th.start()
if what != "idle":
    import numpy as np, torch
    import signals as S
    from fnft_amd import capi
    capi.load()
    t_end = time.time() + secs
    if what == "cfg4":
        q = S.sech_focusing(1 << 20)
        while time.time() < t_end:
            capi.fnft_nsev_ds(q, [-25.0, 25.0], discretization="2SPLIT4B")
    else:
        D = M = 1 << 20
        plan = capi.Plan(D, M, batch=1, discretization="2SPLIT2_MODAL", device=0)
        dq = torch.from_numpy(S.sech_focusing(D, amp=3.2)).cuda()
        out = torch.zeros(3 * M, dtype=torch.complex128, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        while time.time() < t_end:
            for _ in range(200):
                plan.contspec_device(dq.data_ptr(), out.data_ptr(), [-25.0, 25.0], [-1.4, 1.6], kappa=1, contspec_type="BOTH",
                                     normalization_flag=1, stream=st)
            torch.cuda.synchronize()
else:
    time.sleep(secs)
stop = True
th.join(timeout=15)
t0 = samples[0][0] if samples else 0
for t, keep in samples:
    print("%.1f s: %s" % (t - t0, " | ".join(keep)))
