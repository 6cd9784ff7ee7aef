"""Wall-clock of fnft_nsev_inverse through the host-pointer C ABI (GPU) and of the numpy oracle on the same inputs.
    python tests/gpu_debug/inverse_timing.py > gpurun_out/inverse_timing.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import inverse_cases as IC, signals as S
from fnft_amd import capi
from oracle import inverse as INV

capi.load(); capi.silence_errors()


def xi_of(D, T, M):
    return capi.nsev_inverse_XI(D, T, M)[1]


def gpu(case, reps=3):
    ts = []
    for _ in range(reps):
        cs = None if case.get("contspec") is None else np.array(case["contspec"], np.complex128)
        t0 = time.perf_counter()
        rc, q = capi.fnft_nsev_inverse(case["M"], cs, case.get("XI"), case.get("bound_states"), case.get("normconsts"),
                                       case["D"], case["T"], case["kappa"], case["opts"], q_seed=case.get("q_seed"))
        ts.append(time.perf_counter() - t0)
        assert rc == 0, capi.last_error()
    return min(ts), q


def cpu(case):
    cs = None if case.get("contspec") is None else np.array(case["contspec"], np.complex128)
    t0 = time.perf_counter()
    rc, q = INV.fnft_nsev_inverse(case["M"], cs, case.get("XI"), case.get("bound_states"), case.get("normconsts"),
                                  case["D"], case["T"], case["kappa"], case["opts"], q_seed=case.get("q_seed"))
    assert rc == 0
    return time.perf_counter() - t0, q


out = []
for log2D in (14, 16, 18):
    D = 1 << log2D
    T = [-25.0, 25.0]
    t = S.tgrid(T, D)
    # b(xi) of a sech pulse below the soliton threshold (the reference's B_of_tau / b_of_xi test, any D)
    for kind in ("b_of_xi", "B_of_tau"):
        A, t0 = 0.45, 1.2
        case = dict(M=D, D=D, T=T, kappa=1, q_exact=1j * A / np.cosh(t - t0),
                    opts=dict(discretization="2SPLIT2_MODAL", contspec_type="B_OF_XI" if kind == "b_of_xi" else "B_OF_TAU"))
        if kind == "B_of_tau":
            case["XI"] = [-1.0, 1.0]
            case["contspec"] = 1j / (2 * np.pi) * np.sin(np.pi * A) / np.cosh((2 * t - 2 * t0) / 2)
        else:
            XI = xi_of(D, T, D)
            xi = XI[0] + (XI[1] - XI[0]) / (D - 1) * np.arange(D)
            case["XI"] = XI
            case["contspec"] = 1j * np.exp(-2j * xi * t0) * np.sin(np.pi * A) / np.cosh(np.pi * xi)
        tg, qg = gpu(case)
        ent = {"case": kind, "D": D, "gpu_ms": round(tg * 1e3, 2), "rel_err_vs_exact": S.rel_err(qg, case["q_exact"])}
        if log2D <= 16:
            tc, qc = cpu(case)
            ent.update(oracle_ms=round(tc * 1e3, 1), gpu_vs_oracle=S.rel_err(qg, qc))
        out.append(ent)
    # five solitons (the reference's multisoliton test at this D)
    case = IC.multisoliton_cdt("NORMING_CONSTANTS", D)
    tg, qg = gpu(case)
    ent = {"case": "5 solitons", "D": D, "gpu_ms": round(tg * 1e3, 2), "rel_err_vs_exact": S.rel_err(qg, case["q_exact"])}
    tc, qc = cpu(case)
    ent.update(oracle_ms=round(tc * 1e3, 1), gpu_vs_oracle=S.rel_err(qg, qc))
    out.append(ent)
print(json.dumps(out, indent=1))
