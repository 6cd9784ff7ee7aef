"""cfg 4 (D = 2^20, default options, bound states on) stage times; with a library built with -DFNFT_AMD_TUNING
(tests/gpu_debug/build_variant.py) the Aberth sweeps log their largest correction on stderr.
    python tests/gpu_debug/cfg4_sweeps.py [LIB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, signals as S
from fnft_amd import capi
if len(sys.argv) > 1:
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
D = 1 << 20
out = capi.fnft_nsev_ds(S.sech_focusing(D), [-25.0, 25.0], discretization="2SPLIT4B")
out = capi.fnft_nsev_ds(S.sech_focusing(D), [-25.0, 25.0], discretization="2SPLIT4B")
print(out[0], out[1])
for what, ms in capi.discspec_stages():
    print("%-30s %8.3f ms" % (what, ms))
