"""Host <-> device copy rates on the box (pageable / pinned / registered) and the cost of registering a caller's buffer.
usage: python tests/gpu_debug/host_copy_probe.py"""
import ctypes as C, time, numpy as np, torch
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
H2D, D2H = 1, 2
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
for mib in (16, 48):
    nb = mib << 20
    dev = torch.empty(nb, dtype=torch.uint8, device="cuda")
    pag = np.ones(nb, np.uint8)
    pin = torch.ones(nb, dtype=torch.uint8).pin_memory()
    for name, host in (("pageable", pag.ctypes.data), ("pinned", pin.data_ptr())):
        th = t(lambda: hip.hipMemcpy(dev.data_ptr(), host, nb, H2D))
        td = t(lambda: hip.hipMemcpy(host, dev.data_ptr(), nb, D2H))
        print("%2d MiB %-9s H2D %.3f ms (%.1f GB/s)  D2H %.3f ms (%.1f GB/s)" % (mib, name, th * 1e3, nb / th / 1e9, td * 1e3, nb / td / 1e9))
    buf = np.ones(nb, np.uint8)
    t0 = time.perf_counter(); rc = hip.hipHostRegister(buf.ctypes.data, nb, 0); t1 = time.perf_counter()
    th = t(lambda: hip.hipMemcpy(dev.data_ptr(), buf.ctypes.data, nb, H2D))
    td = t(lambda: hip.hipMemcpy(buf.ctypes.data, dev.data_ptr(), nb, D2H))
    t2 = time.perf_counter(); hip.hipHostUnregister(buf.ctypes.data); t3 = time.perf_counter()
    print("%2d MiB registered rc=%d register %.3f ms unregister %.3f ms  H2D %.3f ms (%.1f GB/s) D2H %.3f ms (%.1f GB/s)" % (
        mib, rc, (t1 - t0) * 1e3, (t3 - t2) * 1e3, th * 1e3, nb / th / 1e9, td * 1e3, nb / td / 1e9))
    a = np.ones(nb, np.uint8); b = np.empty(nb, np.uint8)
    tm = min((lambda s: (np.copyto(b, a), time.perf_counter() - s)[1])(time.perf_counter()) for _ in range(5))
    print("%2d MiB single-thread memcpy %.3f ms (%.1f GB/s)" % (mib, tm * 1e3, nb / tm / 1e9))
import os
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
