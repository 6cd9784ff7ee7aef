"""Timeline of the symmetric row kernel from in-kernel s_memrealtime stamps (100 MHz, chip-wide; diagnostic
build with -DFNFT_AMD_STAMPS [-DFNFT_AMD_TUNING]).  usage: python tests/gpu_debug/stamps.py LIB [level] [stagger]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fnft_amd import capi
import signals as S
capi.LIB_PATH = os.path.abspath(sys.argv[1])
level = int(sys.argv[2]) if len(sys.argv) > 2 else 3
stagger = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L = capi.load()
D = M = 1 << 20
plan = capi.Plan(D, M, 1, "2SPLIT2_MODAL")
q = torch.from_numpy(S.sech_focusing(D)).cuda()
out = torch.zeros(3 * M, dtype=torch.complex128, device="cuda")
T, XI = [-25.0, 25.0], [-1.4, 1.6]
L.fnft_amd_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
if stagger:
    L.fnft_amd_debug_tune(plan.h, 0, stagger)
run = lambda: plan.contspec_device(q.data_ptr(), out.data_ptr(), T, XI, 1, "BOTH", 1, torch.cuda.current_stream().cuda_stream)
for _ in range(3): run()
torch.cuda.synchronize()
L.fnft_amd_debug_stamps(plan.h, level, None, 0)
for _ in range(3): run()
torch.cuda.synchronize()
buf = np.zeros(4096 * 16, np.uint64)
L.fnft_amd_debug_stamps(plan.h, level, buf.ctypes.data_as(C.c_void_p), buf.size)
st = buf.reshape(4096, 16)[:, :8].astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
us = (st - t0) * 0.01
names = ["start", "loads issued", "A twiddled", "fft A", "fft B", "product", "inverse fft", "stores issued"]
print("level %d stagger %d: waves %d  kernel span %.2f us  start spread %.2f us" % (level, stagger, len(st), us[:, 7].max(), us[:, 0].max()))
for i in range(8):
    d = us[:, i] - us[:, i - 1] if i else us[:, 0]
    print("  %-16s at %6.2f us (min %6.2f max %6.2f)   phase %5.2f us (min %5.2f max %5.2f)" % (
        names[i], us[:, i].mean(), us[:, i].min(), us[:, i].max(), d.mean(), d.min(), d.max()))
# how many waves are in which phase, per microsecond
print("  t(us): waves in [not started, issuing loads, waiting A, fft A, fft B, product, inverse, storing, done]")
for t in np.arange(0.0, us[:, 7].max() + 1.0, 1.0):
    cnt = [(us[:, 0] > t).sum()] + [((us[:, i] <= t) & (us[:, i + 1] > t)).sum() for i in range(7)] + [(us[:, 7] <= t).sum()]
    print("  %5.1f: %s" % (t, " ".join("%5d" % c for c in cnt)))
