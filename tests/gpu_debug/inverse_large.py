import sys, time, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import signals as S
from fnft_amd import capi
capi.load(); capi.silence_errors()
for log2D in (19, 20):
    D = 1 << log2D
    T = [-32.0, 32.0]
    t = S.tgrid(T, D)
    q0 = 0.4 / np.cosh(t) * np.exp(-1j * t)
    rc, XI = capi.nsev_inverse_XI(D, T, 2 * D, "2SPLIT2_MODAL")
    rc, cs = capi.fnft_nsev(q0, T, 2 * D, XI, kappa=1, discretization="2SPLIT2_MODAL", contspec_type="REFLECTION_COEFFICIENT")
    print("fwd", rc)
    t0 = time.perf_counter()
    rc, q = capi.fnft_nsev_inverse(2 * D, cs[:2 * D].copy(), XI, None, None, D, T, 1, {"discretization": "2SPLIT2_MODAL"})
    print(log2D, "rc", rc, capi.last_error() if rc else "", "ms", (time.perf_counter() - t0) * 1e3, "err", S.rel_err(q, q0) if rc == 0 else None)
    if log2D >= 19:
        A, t00 = 0.45, 1.2
        XI2 = capi.nsev_inverse_XI(D, [-25.0, 25.0], D)[1]
        xi = XI2[0] + (XI2[1] - XI2[0]) / (D - 1) * np.arange(D)
        with np.errstate(over="ignore"):
            c2 = 1j * np.exp(-2j * xi * t00) * np.sin(np.pi * A) / np.cosh(np.pi * xi)
        t0 = time.perf_counter()
        rc, q = capi.fnft_nsev_inverse(D, c2, XI2, None, None, D, [-25.0, 25.0], 1, {"discretization": "2SPLIT2_MODAL", "contspec_type": "B_OF_XI"})
        print("b_of_xi 2^%d rc" % log2D, rc, "ms", (time.perf_counter() - t0) * 1e3, "err", S.rel_err(q, 1j * A / np.cosh(S.tgrid([-25.0, 25.0], D) - t00)) if rc == 0 else capi.last_error())
