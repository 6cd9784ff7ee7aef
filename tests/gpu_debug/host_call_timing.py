"""PCIe-inclusive timing of the drop-in host-pointer fnft_nsev() (not the bench `value`)."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import signals as S
from fnft_amd import capi

for log2D, disc in ((12, "2SPLIT2_MODAL"), (16, "2SPLIT2_MODAL"), (20, "2SPLIT2_MODAL"), (20, "2SPLIT4B")):
    D = M = 1 << log2D
    q = S.sech_focusing(D)
    T, XI = [-25.0, 25.0], [-1.4, 1.6]
    capi.fnft_nsev(q, T, M, XI, discretization=disc, contspec_type="BOTH")  # plan creation + warm-up
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        rc, cs = capi.fnft_nsev(q, T, M, XI, discretization=disc, contspec_type="BOTH")
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    print("host fnft_nsev D=M=2^%d %-14s rc=%d  %.3f ms  %.1f Msamples/s (H2D 16B/sample + D2H 48B/sample included)"
          % (log2D, disc, rc, t * 1e3, D / t / 1e6))
