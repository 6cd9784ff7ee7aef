"""BASELINE.json configs[3]: fnft_nsev D = M = 2^20, contspec + bound states, default options
(2SPLIT4B, SUBSAMPLE_AND_REFINE).  Host-pointer drop-in call, so H2D/D2H are inside the time; the
CPU oracle is timed on the discrete part for comparison (bounded: Newton refinement + norming
constants on the full signal from the converged eigenvalues)."""
import sys, time, json
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import signals as S
from fnft_amd import capi

D = M = 1 << 20
T, XI = [-25.0, 25.0], [-1.4, 1.6]
q = S.sech_focusing(D)
capi.fnft_nsev_ds(q, T, discretization="2SPLIT4B", M=M, XI=XI)   # warm-up: plans, code objects
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    rc, bs, nc, res, cs = capi.fnft_nsev_ds(q, T, discretization="2SPLIT4B", M=M, XI=XI)
    ts.append(time.perf_counter() - t0)
t_all = min(ts)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    capi.fnft_nsev(q, T, M, XI, discretization="2SPLIT4B", contspec_type="BOTH")
    ts.append(time.perf_counter() - t0)
t_cs = min(ts)
out = {"D": D, "rc": rc, "bound_states": [[float(z.real), float(z.imag)] for z in bs],
       "normconsts": [[float(z.real), float(z.imag)] for z in nc],
       "t_contspec_and_bound_states_s": t_all, "t_contspec_only_s": t_cs,
       "t_discrete_part_s": t_all - t_cs, "Msamples_per_s_all": D / t_all / 1e6}
if "--cpu" in sys.argv:
    from oracle import load_oracle
    o = load_oracle()
    t0 = time.perf_counter()
    rc2, a, ap, b = o.scatter_bound_states(q, T, bs, 1, skip_b=False)
    out["cpu_oracle_one_scatter_pass_s"] = time.perf_counter() - t0
print(json.dumps(out))
