"""Sweep a plan tuning parameter (diagnostic build with -DFNFT_AMD_TUNING) and print tree time and the
row-kernel launch times.  usage: python tests/gpu_debug/tune.py LIB which v0 v1 v2 ... [--log2D 20] [--disc X]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fnft_amd import capi
import signals as S
args = sys.argv[1:]
log2D, disc = 20, "2SPLIT2_MODAL"
if "--log2D" in args:
    i = args.index("--log2D"); log2D = int(args[i + 1]); del args[i:i + 2]
if "--disc" in args:
    i = args.index("--disc"); disc = args[i + 1]; del args[i:i + 2]
capi.LIB_PATH = os.path.abspath(args[0])
which = int(args[1]); vals = [int(x) for x in args[2:]]
L = capi.load()
D = M = 1 << log2D
plan = capi.Plan(D, M, 1, disc)
plan.set_timing(True)
q = torch.from_numpy(S.sech_focusing(D)).cuda()
out = torch.zeros(3 * M, dtype=torch.complex128, device="cuda")
T, XI = [-25.0, 25.0], [-1.4, 1.6]
run = lambda: plan.contspec_device(q.data_ptr(), out.data_ptr(), T, XI, 1, "BOTH", 1, torch.cuda.current_stream().cuda_stream)
ref = None
for v in vals:
    L.fnft_amd_debug_tune(plan.h, which, v)
    for _ in range(3): run()
    torch.cuda.synchronize()
    tms = []
    for _ in range(15):
        run(); torch.cuda.synchronize(); tms.append(plan.last_ms(0))
    plan.set_launch_timing(True); run(); torch.cuda.synchronize(); lt = plan.launch_times(); plan.set_launch_timing(False)
    mids = [ms * 1e3 for n, ms in lt if n.startswith("KMid")]
    res = out.cpu().numpy()
    if ref is None: ref = res
    print("param %d = %3d: tree %.4f ms (min %.4f)  row kernels us: %s   max|diff| vs first %.2e" % (
        which, v, float(np.median(tms)), min(tms), " ".join("%.1f" % m for m in mids), float(np.max(np.abs(res - ref)))), flush=True)
