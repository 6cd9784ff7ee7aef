import os, sys
os.environ["FNFT_AMD_DS_TIMING"] = "1"
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
