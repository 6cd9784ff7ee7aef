"""One fnft_nsev_inverse call (b(xi) of a sech pulse, D = 2^LOG2D, default 18) for the profiler.
    python tests/gpu_debug/inverse_one.py [LOG2D [LIB]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import signals as S
from fnft_amd import capi
if len(sys.argv) > 2:
    capi.LIB_PATH = os.path.abspath(sys.argv[2])   # a diagnostic variant (build_variant.py)
capi.load(); capi.silence_errors()
D = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 18)
T = [-25.0, 25.0]
A, t0 = 0.45, 1.2
XI = capi.nsev_inverse_XI(D, T, D)[1]
xi = XI[0] + (XI[1] - XI[0]) / (D - 1) * np.arange(D)
with np.errstate(over="ignore"):
    cs0 = 1j * np.exp(-2j * xi * t0) * np.sin(np.pi * A) / np.cosh(np.pi * xi)
for _ in range(2):
    t = time.perf_counter()
    rc, q = capi.fnft_nsev_inverse(D, cs0.copy(), XI, None, None, D, T, 1, {"discretization": "2SPLIT2_MODAL", "contspec_type": "B_OF_XI"})
    print(rc, (time.perf_counter() - t) * 1e3, "ms", S.rel_err(q, 1j * A / np.cosh(S.tgrid(T, D) - t0)))
