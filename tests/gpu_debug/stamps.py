"""Timeline of the symmetric row kernel from in-kernel s_memtime stamps (diagnostic build with
-DFNFT_AMD_STAMPS).  usage: python tests/gpu_debug/stamps.py LIB [level] [log2D]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fnft_amd import capi
import signals as S
capi.LIB_PATH = os.path.abspath(sys.argv[1])
level = int(sys.argv[2]) if len(sys.argv) > 2 else 3
log2D = int(sys.argv[3]) if len(sys.argv) > 3 else 20
L = capi.load()
D = M = 1 << log2D
plan = capi.Plan(D, M, 1, "2SPLIT2_MODAL")
q = torch.from_numpy(S.sech_focusing(D)).cuda()
out = torch.zeros(3 * M, dtype=torch.complex128, device="cuda")
T, XI = [-25.0, 25.0], [-1.4, 1.6]
L.fnft_amd_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
run = lambda: plan.contspec_device(q.data_ptr(), out.data_ptr(), T, XI, 1, "BOTH", 1, torch.cuda.current_stream().cuda_stream)
for _ in range(3): run()
torch.cuda.synchronize()
L.fnft_amd_debug_stamps(plan.h, level, None, 0)
for _ in range(3): run()
torch.cuda.synchronize()
nw = 2048 * (8 // 8)
buf = np.zeros(4096 * 16, np.uint64)
L.fnft_amd_debug_stamps(plan.h, level, buf.ctypes.data_as(C.c_void_p), buf.size)
st = buf.reshape(4096, 16)[:, :8].astype(np.int64)
st = st[st[:, 0] > 0]
print("waves stamped:", len(st))
t0 = st[:, 0].min()
names = ["start", "loads issued", "loads arrived+twiddled", "fft A", "fft B", "product", "inverse fft", "stores issued"]
# s_memtime ticks at 100 MHz (constant), 10 ns per tick
print("kernel span (first start -> last end): %.2f us" % ((st[:, 7].max() - t0) * 0.01))
print("start spread: %.2f us" % ((st[:, 0].max() - t0) * 0.01))
for i in range(8):
    rel = (st[:, i] - t0) * 0.01
    d = (st[:, i] - st[:, i - 1]) * 0.01 if i else rel
    print("%-26s at mean %6.2f us (min %6.2f max %6.2f)   phase mean %5.2f us  (min %5.2f max %5.2f)" % (
        names[i], rel.mean(), rel.min(), rel.max(), d.mean(), d.min(), d.max()))
