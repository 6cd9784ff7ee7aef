"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the CPU oracle and the
reference's known-answer vectors.  Run with `pytest -m gpu` on an MI355X.

Tolerances (floating point, fp64): the reference's own unit tests use rel-L1 <= 100*eps at toy
sizes; for full transforms SURVEY.md section 8c gives GPU-vs-CPU rel-L1 bounds on a/b/rho of
1e-12 (D <= 2^12), 2e-11 (D <= 2^16), 5e-10 (D = 2^20) -- about 8x the reference's own distance
from the exact answer at those sizes.
"""
import numpy as np
import pytest

import signals as S

pytestmark = pytest.mark.gpu

EPS = np.finfo(np.float64).eps
AKNS_SCHEMES = ["2SPLIT2_MODAL", "2SPLIT1A", "2SPLIT1B", "2SPLIT2A", "2SPLIT2B", "2SPLIT2S",
                "2SPLIT3A", "2SPLIT3B", "2SPLIT3S", "2SPLIT4A", "2SPLIT4B", "2SPLIT5A", "2SPLIT5B",
                "2SPLIT6A", "2SPLIT6B", "2SPLIT7A", "2SPLIT7B", "2SPLIT8A", "2SPLIT8B"]
ALL_FAST = AKNS_SCHEMES + ["4SPLIT4A", "4SPLIT4B"]


def tol_for(D):
    if D <= 4096:
        return 1e-12
    if D <= 65536:
        return 2e-11
    return 5e-10


@pytest.fixture(scope="module")
def capi(lib):
    from fnft_amd import capi as c
    assert lib.fnft_amd_device_count() >= 1, c.last_error()
    return c


# ---- reference known-answer vectors ---------------------------------------------------------
@pytest.mark.parametrize("key", ["fmult2x2_pow2", "fmult2x2_nopow2"])
@pytest.mark.parametrize("normalize", [False, True])
def test_poly_fmult2x2_golden(capi, fixtures, key, normalize):
    fx = fixtures[key]
    exact = S.l2c(fx["result_exact"])
    p = S.fmult_test_input(fx["deg"], fx["n"])
    rc, d, res, W = capi.poly_fmult2x2(fx["deg"], fx["n"], p, normalize=normalize)
    assert rc == 0, capi.last_error()
    assert d == exact.size // 4 - 1
    if normalize:
        assert W != 0
        res = res * 2.0 ** W
    assert S.rel_err(res.ravel(), exact) <= fx["tol_rel_l1"]


def test_device_fft_known_answer(capi, fixtures):
    """test/fnft__fft_wrapper/fnft__fft_wrapper_test.c:31-32: the length-4 known answer of the reference's FFT
    back end, on the device transforms.  There is no FFT entry point in either library (the reference's wrapper
    is header-only); the chirp z-transform with A = 1, W = exp(-2 pi i/4), M = 4 IS the DFT of the reversed
    coefficient vector, and on the GPU it runs three workgroup FFT passes per direction (columns and rows of an
    8192-point transform, forward and inverse), so a wrong butterfly, twiddle or exchange shows up here at 100 eps."""
    x = S.l2c(fixtures["fft_wrapper"]["in_exact"])
    X = S.l2c(fixtures["fft_wrapper"]["out_exact"])
    # result[m] = sum_n p[deg-n] (A W^-m)^-n... with A = 1: sum_n p[deg - n] W^{m n}: p = reversed x
    rc, out = capi.poly_chirpz(x[::-1].copy(), 1.0 + 0j, np.exp(-2j * np.pi / 4), 4)
    assert rc == 0, capi.last_error()
    assert S.rel_err(out, X) <= 100 * 2.220446049250313e-16
    # inverse direction (sign +1, no 1/N), :53-60 of the same file: W = exp(+2 pi i/4) maps out_exact back to 4*in
    rc, back = capi.poly_chirpz(X[::-1].copy(), 1.0 + 0j, np.exp(2j * np.pi / 4), 4)
    assert rc == 0
    assert S.rel_err(back, 4 * x) <= 100 * 2.220446049250313e-16


def test_poly_chirpz_golden(capi, fixtures):
    fx = fixtures["chirpz"]
    p = S.l2c(fx["p"])
    for M, key in ((3, "result_M3"), (6, "result_M6")):
        rc, out = capi.poly_chirpz(p, complex(*fx["A"]), np.exp(1j * fx["W_arg"]), M)
        assert rc == 0, capi.last_error()
        assert S.rel_err(out, S.l2c(fx[key])) <= fx["tol_rel_l1"]


@pytest.mark.parametrize("scheme", AKNS_SCHEMES)
@pytest.mark.parametrize("normalize", [False, True])
def test_akns_fscatter_golden(capi, oracle, fixtures, scheme, normalize):
    fx = fixtures["akns_fscatter"]["schemes"][scheme]
    q, r, z = S.akns_test_signal(fx["D"])
    rc, deg, tm, W = capi.akns_fscatter(q, r, fx["eps_t"], scheme, normalize=normalize)
    assert rc == 0, capi.last_error()
    if normalize:
        assert W != 0
        tm = tm * 2.0 ** W
    vals = np.concatenate([oracle.poly_eval(tm[e], z) for e in range(4)])
    assert S.rel_err(vals, S.l2c(fx["result_exact"])) <= fx["tol_rel_l1"]  # err_bnd of the reference test of this scheme


# ---- product tree vs oracle on seeded inputs ----------------------------------------------------
@pytest.mark.parametrize("deg,n", [(1, 1), (1, 2), (1, 3), (1, 7), (2, 8), (2, 13), (3, 6), (4, 5),
                                   (5, 9), (7, 33), (1, 64), (2, 100), (3, 300)])
def test_poly_fmult2x2_vs_oracle(capi, oracle, deg, n):
    rng = np.random.default_rng(1000 * deg + n)
    p = 0.3 * (rng.standard_normal((4, n * (deg + 1))) + 1j * rng.standard_normal((4, n * (deg + 1))))
    rc, d, res, W = capi.poly_fmult2x2(deg, n, p)
    assert rc == 0, capi.last_error()
    d2, ref, W2 = oracle.poly_fmult2x2(deg, n, p)
    assert d == d2 == deg * n
    a = res.astype(np.clongdouble) * np.ldexp(np.longdouble(1), W)
    b = ref.astype(np.clongdouble) * np.ldexp(np.longdouble(1), W2)
    if n < 100:
        assert float(np.sum(np.abs(a - b)) / np.sum(np.abs(b))) < 1e-12
    else:
        # Products of many random factors have a coefficient range of hundreds of decades and
        # any FFT product (the reference's and the oracle's included) keeps absolute accuracy
        # only, so two FFT implementations differ by far more than either differs from the
        # exact product.  Compare with the direct (convolution) product in extended precision.
        exact = S.tree_direct(p.astype(np.clongdouble), deg, n)
        assert float(np.sum(np.abs(a - exact)) / np.sum(np.abs(exact))) < 1e-10


def _tm_err(capi_tm, W, ref_tm, W2):
    a = capi_tm.astype(np.clongdouble) * np.ldexp(np.longdouble(1), W)
    b = ref_tm.astype(np.clongdouble) * np.ldexp(np.longdouble(1), W2)
    return float(np.sum(np.abs(a - b)) / np.sum(np.abs(b)))


@pytest.mark.parametrize("D,disc", [(2, "2SPLIT2_MODAL"), (3, "2SPLIT4B"), (255, "2SPLIT2_MODAL"),
                                    (4096, "2SPLIT2_MODAL"), (4097, "2SPLIT4B"), (8192, "2SPLIT4B"),
                                    (16384, "2SPLIT2_MODAL"), (3000, "2SPLIT3A"), (5000, "2SPLIT4A"),
                                    (65536, "2SPLIT2_MODAL"), (65536, "2SPLIT4B"),
                                    (1024, "2SPLIT5A"), (1000, "2SPLIT5B"), (1024, "2SPLIT6A"),
                                    (4096, "2SPLIT6B"), (256, "2SPLIT7A"), (300, "2SPLIT7B"),
                                    (2048, "2SPLIT8A"), (4096, "2SPLIT8B"), (16384, "2SPLIT6B")])
@pytest.mark.parametrize("kappa", [1, -1])
def test_nse_fscatter_vs_oracle(capi, oracle, D, disc, kappa):
    """Transfer matrices (coeffs * 2^W) of sech pulses; covers fused levels and split transforms."""
    T = (-25.0, 25.0)
    q = S.sech_focusing(D, T, amp=1.4 if kappa == -1 else 3.2) * np.exp(0.3j * S.tgrid(T, D))
    eps_t = (T[1] - T[0]) / (D - 1)
    rc, deg, tm, W = capi.nse_fscatter(q, eps_t, kappa, disc)
    assert rc == 0, capi.last_error()
    rc2, deg2, ref, W2 = oracle.nse_fscatter(q, eps_t, kappa, disc)
    assert rc2 == 0 and deg == deg2
    # coeffs * 2^W in the reference's own metric (misc_rel_err, sum|d| / sum|exact|)
    assert _tm_err(tm, W, ref, W2) < tol_for(D)


# ---- full transform vs oracle ---------------------------------------------------------------------
@pytest.mark.parametrize("D,M,disc", [
    (2, 2, "2SPLIT2_MODAL"), (256, 16, "2SPLIT2_MODAL"), (256, 64, "2SPLIT4B"), (1000, 37, "2SPLIT2A"),
    (4096, 16, "2SPLIT2_MODAL"), (4096, 4096, "2SPLIT2_MODAL"), (4097, 100, "2SPLIT4B"),
    (4095, 4095, "2SPLIT4B"), (300, 50, "2SPLIT3A"), (512, 33, "2SPLIT4A"), (777, 40, "2SPLIT1A"),
    (640, 64, "2SPLIT1B"), (2048, 128, "2SPLIT2B"), (2048, 128, "2SPLIT2S"), (1500, 64, "2SPLIT3B"),
    (1500, 64, "2SPLIT3S"), (16384, 16384, "2SPLIT2_MODAL"), (65536, 65536, "2SPLIT2_MODAL"),
    (65536, 1000, "2SPLIT4B"),
    (1024, 1024, "2SPLIT5A"), (1025, 200, "2SPLIT5B"), (1024, 1024, "2SPLIT6A"), (4096, 4096, "2SPLIT6B"),
    (256, 256, "2SPLIT7A"), (512, 100, "2SPLIT7B"), (1024, 512, "2SPLIT8A"), (2048, 2048, "2SPLIT8B"),
    (512, 512, "4SPLIT4A"), (513, 100, "4SPLIT4B"), (1000, 64, "4SPLIT4A"), (4096, 4096, "4SPLIT4B"),
    (3, 8, "4SPLIT4B"), (16384, 1000, "4SPLIT4A"),
])
def test_fnft_nsev_vs_oracle(capi, oracle, D, M, disc):
    T, XI = [-25.0, 25.0], [-1.4, 1.6]
    q = S.sech_focusing(D)
    rc, cs = capi.fnft_nsev(q, T, M, XI, kappa=1, discretization=disc, contspec_type="BOTH")
    assert rc == 0, capi.last_error()
    rc2, ref = oracle.fnft_nsev(q, T, M, XI, kappa=1, disc=disc, cstype="BOTH")
    assert rc2 == 0
    tol = tol_for(D)
    if disc[6] in "5678":   # order 5..8 schemes: conditioning-aware bound, see signals.contspec_tol
        tol = S.contspec_tol(oracle, q, T, 1, disc, tol)
    assert S.rel_err(cs[:M], ref[:M]) < tol
    assert S.rel_err(cs[M:2 * M], ref[M:2 * M]) < tol
    assert S.rel_err(cs[2 * M:], ref[2 * M:]) < tol


@pytest.mark.parametrize("cstype", ["RHO", "AB", "BOTH"])
@pytest.mark.parametrize("norm", [0, 1])
def test_cstype_and_normalization(capi, oracle, cstype, norm):
    D, M = 1024, 200
    T, XI = [-20.0, 22.0], [-3.0, 2.5]
    q = S.batch_signal(3, D, T)
    rc, cs = capi.fnft_nsev(q, T, M, XI, kappa=1, discretization="2SPLIT4B", contspec_type=cstype,
                            normalization_flag=norm)
    assert rc == 0, capi.last_error()
    rc2, ref = oracle.fnft_nsev(q, T, M, XI, kappa=1, disc="2SPLIT4B", cstype=cstype, normalize=bool(norm))
    assert rc2 == 0 and cs.size == ref.size
    assert S.rel_err(cs, ref) < 1e-12


def test_defocusing(capi, oracle, fixtures):
    fx = fixtures["nsev_sech_defocusing"]
    D, M = 4096, fx["M"]
    q = S.sech_defocusing(D)
    for disc, bound in (("2SPLIT2_MODAL", 1.2e-4), ("2SPLIT4B", 1.3e-4)):
        rc, cs = capi.fnft_nsev(q, fx["T"], M, fx["XI"], kappa=-1, discretization=disc,
                                contspec_type="REFLECTION_COEFFICIENT")
        assert rc == 0, capi.last_error()
        assert S.rel_err(cs, S.l2c(fx["contspec"])) <= bound  # test/fnft_nsev/..._defocusing_*.c
        rc2, ref = oracle.fnft_nsev(q, fx["T"], M, fx["XI"], kappa=-1, disc=disc, cstype="RHO")
        assert S.rel_err(cs, ref) < 1e-12


# ---- the reference's analytic integration tests, run on the GPU path ---------------------------
# slow (non-polynomial) discretizations: outside the fast path this repo covers
SLOW_SCHEMES = ("BO", "CF4_2", "CF4_3", "CF5_3", "CF6_4", "ES4", "TES4")


def _analytic_cases():
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "reference_fixtures.json")) as f:
        fx = json.load(f)
    out = []
    for b in fx["nsev_error_bounds"]:
        # every file of test/fnft_nsev/ is either replayed or named here with the reason it is not
        if b["discretization"] not in ALL_FAST:
            assert b["discretization"] in SLOW_SCHEMES, "fixture file %s would be skipped silently" % b["file"]
            continue
        if b["testcase"] == "SECH_FOCUSING2":
            continue   # second focusing test case (fnft__nsev_testcases.c:289-461): not extracted
        assert b["stages"], b["file"]
        out.append(pytest.param(b, id=b["file"].replace("fnft_nsev_test_", "").replace(".c", "")))
    return out


def _exact(fixtures, testcase):
    if testcase == "SECH_FOCUSING":
        fx = fixtures["nsev_sech_focusing"]
        return fx, S.sech_focusing, S.l2c(fx["contspec"]), S.l2c(fx["ab"])
    if testcase == "SECH_DEFOCUSING":
        fx = fixtures["nsev_sech_defocusing"]
        return fx, S.sech_defocusing, S.l2c(fx["contspec"]), None
    fx = fixtures["nsev_truncated_soliton"]
    return fx, S.truncated_soliton, S.truncated_soliton_contspec(fx["XI"], fx["M"]), None


@pytest.mark.parametrize("b", _analytic_cases())
def test_fnft_nsev_analytic_bounds(capi, fixtures, b):
    """src/private/fnft__nsev_testcases.c:711-822 driven as test/fnft_nsev/<file>.c drives it: every
    harness call of the file -- D, D+1, D-1, the finer grid with rescaled bounds, the Richardson
    stages -- is one entry of b["stages"] (tests/golden/extract_reference_fixtures.py)."""
    fx, sig, exact_rho, exact_ab = _exact(fixtures, b["testcase"])
    M = fx["M"]
    assert b["stages"]
    for st in b["stages"]:
        rc, cs = capi.fnft_nsev(sig(st["D"]), fx["T"], M, fx["XI"], kappa=fx["kappa"],
                                discretization=b["discretization"], contspec_type="BOTH",
                                richardson=bool(st["richardson"]))
        assert rc == 0, capi.last_error()
        errs = [S.rel_err(cs[:M], exact_rho)]
        if exact_ab is not None:
            errs += [S.rel_err(cs[M:2 * M], exact_ab[:M]), S.rel_err(cs[2 * M:], exact_ab[M:])]
        assert all(np.isfinite(e) for e in errs), (st, errs, b["file"])   # files with infinite bounds still have to run
        for e, bound in zip(errs, st["bounds"]):
            if np.isfinite(bound):
                assert e <= bound, (st, errs, b["file"])


@pytest.mark.parametrize("disc,kappa,D", [("2SPLIT4A", 1, 4096), ("2SPLIT4A", -1, 4097), ("2SPLIT2A", 1, 3001),
                                          ("4SPLIT4A", 1, 512), ("4SPLIT4B", -1, 800), ("4SPLIT4A", 1, 511)])
def test_fnft_nsev_richardson_vs_oracle(capi, oracle, disc, kappa, D):
    """richardson_extrapolation_flag = 1 (src/fnft_nsev.c:316-406) against the oracle's restatement
    of the same combination (order 2 for 2SPLIT, 4 for 4SPLIT; coarse 4SPLIT pass resamples the
    full signal)."""
    T, XI, M = [-25.0, 25.0], [-1.4, 1.6], 64
    q = S.sech_focusing(D, amp=3.2 if kappa == 1 else 1.1)
    for cst, oc in (("BOTH", "BOTH"), ("REFLECTION_COEFFICIENT", "RHO")):
        rc, cs = capi.fnft_nsev(q, T, M, XI, kappa=kappa, discretization=disc, contspec_type=cst,
                                richardson=True)
        assert rc == 0, capi.last_error()
        rc2, ref = oracle.fnft_nsev(q, T, M, XI, kappa=kappa, disc=disc, cstype=oc, richardson=True)
        assert rc2 == 0
        for j in range(len(ref) // M):
            assert S.rel_err(cs[j * M:(j + 1) * M], ref[j * M:(j + 1) * M]) < 1e-11, (cst, j)


# ---- BASELINE.json full size: properties that do not need a CPU run of that size ---------------
@pytest.mark.parametrize("disc,order_bound,floor", [("2SPLIT2_MODAL", 5.0e-3, 2e-9), ("2SPLIT4B", 3.9e-6, 2e-8)])
def test_full_size_2p20_analytic(capi, disc, order_bound, floor):
    """cfg 2: D = M = 2^20.  Every grid point is compared with the closed-form Satsuma-Yajima
    spectrum; the discretization error of a second-order scheme falls as (4096/D)^2 from the
    reference's own bound at D = 4096 (test/fnft_nsev/fnft_nsev_test_sech_focusing_*.c)."""
    D = M = 1 << 20
    T, XI = [-25.0, 25.0], [-7.0 / 5.0, 8.0 / 5.0]
    q = S.sech_focusing(D)
    rc, cs = capi.fnft_nsev(q, T, M, XI, kappa=1, discretization=disc, contspec_type="BOTH")
    assert rc == 0, capi.last_error()
    xi = XI[0] + np.arange(M) * (XI[1] - XI[0]) / (M - 1)
    a, b = S.sech_focusing_analytic(xi)
    # floor: the 2SPLIT4B coefficient formulas (fnft__akns_fscatter.c:414-425) are differences
    # of O(1) terms that cancel to O((eps_t*|q|)^2); at eps_t = 4.8e-5 that costs ~8 digits in the
    # reference's arithmetic as much as here, so the scheme cannot get below ~5e-9 at this D
    bound = max(order_bound * (4096.0 / D) ** 2 * 4.0, floor)
    assert S.rel_err(cs[M:2 * M], a) < bound
    assert S.rel_err(cs[2 * M:], b) < bound
    assert S.rel_err(cs[:M], b / a) < bound
    # |a|^2 + |b|^2 = 1 on the real axis for the focusing NSE (unimodularity of the transfer
    # matrix): a size-independent invariant of the whole pipeline
    inv = np.abs(cs[M:2 * M]) ** 2 + np.abs(cs[2 * M:]) ** 2
    assert np.max(np.abs(inv - 1.0)) < 1e-7


@pytest.mark.parametrize("disc", ["2SPLIT2_MODAL", "2SPLIT4B"])
def test_full_size_2p20_vs_oracle(capi, oracle, disc):
    """cfg 2 against the CPU oracle at the full size (5 s of CPU for MODAL, 10 s for the library's default 2SPLIT4B):
    a, b and rho."""
    D = M = 1 << 20
    T, XI = [-25.0, 25.0], [-7.0 / 5.0, 8.0 / 5.0]
    q = S.sech_focusing(D)
    rc, cs = capi.fnft_nsev(q, T, M, XI, kappa=1, discretization=disc, contspec_type="BOTH")
    assert rc == 0, capi.last_error()
    rc2, ref = oracle.fnft_nsev(q, T, M, XI, kappa=1, disc=disc, cstype="BOTH")
    assert rc2 == 0
    tol = tol_for(D)
    for k in range(3):
        assert S.rel_err(cs[k * M:(k + 1) * M], ref[k * M:(k + 1) * M]) < tol


# ---- device-resident batch API -------------------------------------------------------------------
def test_batch_plan_device(capi, oracle):
    import torch
    D, M, B = 4096, 512, 6
    T, XI = [-25.0, 25.0], [-4.0, 4.0]
    qs = np.stack([S.batch_signal(k, D, T) for k in range(B)])
    plan = capi.Plan(D, M, batch=B, discretization="2SPLIT2_MODAL")
    dq = torch.from_numpy(qs).cuda()
    out = torch.zeros(B * 3 * M, dtype=torch.complex128, device="cuda")
    torch.cuda.synchronize()
    rc = plan.contspec_device(dq.data_ptr(), out.data_ptr(), T, XI, kappa=1, contspec_type="BOTH")
    assert rc == 0, capi.last_error()
    assert plan.finish() == 0
    res = out.cpu().numpy().reshape(B, 3 * M)
    for k in range(B):
        rc2, ref = oracle.fnft_nsev(qs[k], T, M, XI, kappa=1, disc="2SPLIT2_MODAL", cstype="BOTH")
        assert rc2 == 0
        assert S.rel_err(res[k], ref) < 1e-12, k
    rc, deg, tm, W = plan.transfer_matrix(2)
    eps_t = (T[1] - T[0]) / (D - 1)
    rc2, deg2, ref_tm, W2 = oracle.nse_fscatter(qs[2], eps_t, 1, "2SPLIT2_MODAL")
    assert rc == 0 and deg == deg2
    assert _tm_err(tm, W, ref_tm, W2) < 1e-10
    plan.close()


def test_batch_cfg3_shard(capi, oracle):
    """BASELINE.json configs[2] as one GPU sees it: 64 of the 512 signals, D = M = 2^16, MODAL,
    XI = [-4, 4].  Three signals against the oracle, all 64 through rho = b/a and |a|^2 + |b|^2 = 1.
    The MODAL step is unitary on the unit circle (fnft__akns_fscatter.c:118-148), but the chirp
    z-transform of the reference evaluates at z_m = V^m / A with the double-rounded V
    (fnft_nsev.c:822-833, fnft__poly_chirpz.c:52-95): |V| = 1 +- 1.1e-16, so z_m sits m*1.1e-16 off
    the circle and a degree-D polynomial magnifies that to at most D*M*2.2e-16 (= 9.5e-7 here; the
    oracle shows the same 3e-8 .. 7e-8 in the band of the pulse).  Parity with the reference means
    reproducing that, so the bound of the invariant is D*M*2.2e-16, not round-off."""
    import torch
    D = M = 1 << 16
    B = 64
    first = 128  # the shard of rank 2
    T, XI = [-25.0, 25.0], [-4.0, 4.0]
    qs = np.stack([S.batch_signal(first + k, D, T) for k in range(B)])
    plan = capi.Plan(D, M, batch=B, discretization="2SPLIT2_MODAL")
    dq = torch.from_numpy(qs).cuda()
    out = torch.zeros(B * 3 * M, dtype=torch.complex128, device="cuda")
    rc = plan.contspec_device(dq.data_ptr(), out.data_ptr(), T, XI, kappa=1, contspec_type="BOTH")
    assert rc == 0, capi.last_error()
    assert plan.finish() == 0
    res = out.cpu().numpy().reshape(B, 3, M)
    for k in (0, 31, 63):
        rc2, ref = oracle.fnft_nsev(qs[k], T, M, XI, kappa=1, disc="2SPLIT2_MODAL", cstype="BOTH")
        assert rc2 == 0
        for j in range(3):
            assert S.rel_err(res[k, j], ref[j * M:(j + 1) * M]) < 2e-11, (k, j)
    a, b, rho = res[:, 1], res[:, 2], res[:, 0]
    assert np.max(np.abs(np.abs(a) ** 2 + np.abs(b) ** 2 - 1.0)) < D * M * 2.2e-16
    assert np.max(np.abs(rho * a - b)) < 1e-12 * max(1.0, float(np.max(np.abs(b))))
    plan.close()


# ---- error behaviour that needs the device ----------------------------------------------------------
def test_modal_step_size_error(capi):
    """fnft__akns_fscatter.c:122-126 -> fnft_nsev returns -5 (E_OTHER wrapped by CHECK_RETCODE)."""
    capi.silence_errors()
    q = np.full(16, 40.0 + 0j)
    rc, _ = capi.fnft_nsev(q, [0.0, 1.0], 8, [-1.0, 1.0], kappa=-1, discretization="2SPLIT2_MODAL")
    assert rc == -5


# ---- fnft_kdvv -----------------------------------------------------------------------------------------
def _kdvv_cases():
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "reference_fixtures.json")) as f:
        fx = json.load(f)
    return [pytest.param(b, id=b["file"].replace("fnft_kdvv_test_", "").replace(".c", ""))
            for b in fx["kdvv_error_bounds"]]


@pytest.mark.parametrize("b", _kdvv_cases())
def test_fnft_kdvv_analytic_bounds(capi, oracle, fixtures, b):
    """src/private/fnft__kdvv_testcases.c:294-367 driven as test/fnft_kdvv/<file>.c drives it (all 37
    files, every harness call), and the same calls against the oracle."""
    assert b["stages"]
    for st in b["stages"]:
        u, T, XI, M, exact = S.kdvv_case(fixtures, b["testcase"], st["D"])
        rc, cs = capi.fnft_kdvv(u, T, M, XI, discretization=b["discretization"])
        assert rc == 0, capi.last_error()
        assert S.rel_err(cs, exact) <= st["bounds"][0], (st, b["file"])
        rc2, ref = oracle.fnft_kdvv(u, T, M, XI, b["discretization"])
        assert rc2 == 0
        assert S.rel_err(cs, ref) < (1e-9 if b["discretization"][6] in "5678" else 1e-11), (st, b["file"])


@pytest.mark.parametrize("D,M,disc", [(4096, 4096, "2SPLIT4B"), (4097, 1000, "2SPLIT2A"), (65536, 65536, "2SPLIT4B"),
                                      (16384, 16384, "2SPLIT8B"), (3000, 100, "2SPLIT3S")])
def test_fnft_kdvv_vs_oracle(capi, oracle, D, M, disc):
    T, XI = [-16.0, 15.0], [-71.0 / 20.0, 79.0 / 20.0]
    u = S.kdvv_sech(D, T)
    rc, cs = capi.fnft_kdvv(u, T, M, XI, discretization=disc)
    assert rc == 0, capi.last_error()
    rc2, ref = oracle.fnft_kdvv(u, T, M, XI, disc)
    assert rc2 == 0
    assert S.rel_err(cs, ref) < (2e-11 if D <= 4097 else 2e-10)


def test_kdv_fscatter_vs_oracle(capi, oracle):
    T = (-16.0, 15.0)
    for D, disc in ((512, "2SPLIT4B"), (300, "2SPLIT2A"), (64, "2SPLIT8B")):
        u = S.kdvv_sech(D, T)
        eps_t = (T[1] - T[0]) / (D - 1)
        rc, deg, tm, W = capi.kdv_fscatter(u, eps_t, disc)
        assert rc == 0, capi.last_error()
        rc2, deg2, ref, W2 = oracle.kdv_fscatter(u, eps_t, disc, normalize=True)
        assert rc2 == 0 and deg == deg2
        assert _tm_err(tm, W, ref, W2) < 1e-12


def test_kdvv_device_batch(capi, oracle):
    import torch
    D, M, B = 2048, 512, 5
    T, XI = [-16.0, 15.0], [-3.0, 3.5]
    # amplitudes away from N(N+1) = 2, 6 (reflectionless potentials: rho ~ 0 has no relative accuracy)
    us = np.stack([(1.1 + 0.37 * k) * S.kdvv_sech(D, T) / 3.2 for k in range(B)])
    plan = capi.KdvvPlan(D, M, batch=B, discretization="2SPLIT4B")
    du = torch.from_numpy(us).cuda()
    out = torch.zeros(B * M, dtype=torch.complex128, device="cuda")
    rc = plan.contspec_device(du.data_ptr(), out.data_ptr(), T, XI)
    assert rc == 0, capi.last_error()
    assert plan.finish() == 0
    res = out.cpu().numpy().reshape(B, M)
    for k in range(B):
        rc2, ref = oracle.fnft_kdvv(us[k], T, M, XI, "2SPLIT4B")
        assert rc2 == 0 and S.rel_err(res[k], ref) < 1e-11, k
    plan.close()


@pytest.mark.parametrize("D,M,disc", [(4096, 4096, "2SPLIT8B"), (2048, 512, "2SPLIT2A"), (3000, 700, "2SPLIT4A"),
                                      (8192, 8192, "2SPLIT4B"), (1 << 15, 4096, "2SPLIT8B"), (5000, 333, "2SPLIT5B"),
                                      (1 << 14, 1000, "2SPLIT1A")])
def test_kdvv_real_and_complex_tree(capi, oracle, D, M, disc):
    """The real-coefficient tree (nft_real.h: folded negacyclic transforms; what a real potential gets) and the complex
    general-form tree on the same real potential: both against the oracle, and against each other.  Sizes cover direct
    products, single-workgroup pair products (exact and loose transform lengths) and split levels with bridges."""
    import torch
    T, XI = [-16.0, 15.0], [-71.0 / 20.0, 79.0 / 20.0]
    u = 0.8 * S.kdvv_sech(D, T)
    du = torch.from_numpy(u).cuda()
    res = {}
    for mode in (1, 0, -1):
        plan = capi.KdvvPlan(D, M, batch=1, discretization=disc)
        assert plan.set_real_mode(mode) == 0
        out = torch.zeros(M, dtype=torch.complex128, device="cuda")
        plan.set_launch_timing(True)
        rc = plan.contspec_device(du.data_ptr(), out.data_ptr(), T, XI)
        assert rc == 0, capi.last_error()
        assert plan.finish() == 0
        torch.cuda.synchronize()
        names = [n for n, _ in plan.launch_times()]
        used_real = any(n.startswith(("KRPair", "KRCol", "KRBridge")) for n in names)
        assert used_real == (mode != 0), (mode, names)   # the default mode finds the potential real
        res[mode] = out.cpu().numpy()
        plan.close()
    rc2, ref = oracle.fnft_kdvv(u, T, M, XI, disc)
    assert rc2 == 0
    tol = 1e-9 if disc[6] in "5678" else 2e-11
    assert S.rel_err(res[1], ref) < tol and S.rel_err(res[0], ref) < tol
    assert S.rel_err(res[1], res[0]) < tol
    assert np.array_equal(res[1], res[-1])


def test_kdvv_complex_potential_takes_complex_tree(capi, oracle):
    """A potential with a non-zero imaginary part: the default mode falls back to the complex tree (same result as the
    oracle, which computes in complex arithmetic like the reference); mode 1 ('the caller guarantees a real u')
    reports the violation from fnft_amd_plan_finish instead of returning a wrong spectrum."""
    import torch
    capi.silence_errors()
    D, M = 2048, 256
    T, XI = [-16.0, 15.0], [-3.0, 3.5]
    u = 0.8 * S.kdvv_sech(D, T) * (1.0 + 0.05j)
    rc, cs = capi.fnft_kdvv(u, T, M, XI, discretization="2SPLIT4B")
    assert rc == 0
    rc2, ref = oracle.fnft_kdvv(u, T, M, XI, "2SPLIT4B")
    assert rc2 == 0 and S.rel_err(cs, ref) < 2e-11
    du = torch.from_numpy(u).cuda()
    out = torch.zeros(M, dtype=torch.complex128, device="cuda")
    plan = capi.KdvvPlan(D, M, batch=1, discretization="2SPLIT4B")
    assert plan.contspec_device(du.data_ptr(), out.data_ptr(), T, XI) == 0 and plan.finish() == 0
    assert S.rel_err(out.cpu().numpy(), ref) < 2e-11
    plan.set_real_mode(1)
    assert plan.contspec_device(du.data_ptr(), out.data_ptr(), T, XI) == 0
    assert plan.finish() == capi.FNFT_EC_INVALID_ARGUMENT
    plan.close()


def test_kdv_fscatter_real_tree_exports_reference_layout(capi, oracle):
    """fnft__kdv_fscatter on a real potential rides on the real-coefficient tree; the exported transfer matrix is the
    reference's complex layout (imaginary parts exactly zero)."""
    T = (-16.0, 15.0)
    for D, disc in ((4096, "2SPLIT8B"), (1000, "2SPLIT4A")):
        u = S.kdvv_sech(D, T)
        eps_t = (T[1] - T[0]) / (D - 1)
        rc, deg, tm, W = capi.kdv_fscatter(u, eps_t, disc)
        assert rc == 0, capi.last_error()
        assert np.all(tm.imag == 0.0)
        rc2, deg2, ref, W2 = oracle.kdv_fscatter(u, eps_t, disc, normalize=True)
        assert rc2 == 0 and deg == deg2
        assert _tm_err(tm, W, ref, W2) < 1e-11


def test_kdvv_4split4_is_2split4_on_raw_samples(capi, oracle):
    """fnft_kdvv / fnft__kdv_fscatter with 4SPLIT4A/B: the reference maps them to the AKNS schemes of the same name
    (src/private/fnft__kdv_discretization.c:139-143) and kdv_fscatter hands the samples over as they are, where those
    schemes share formulas and degree with 2SPLIT4A/B (fnft__akns_fscatter.c:362-363,402-403): identical results."""
    D, M = 1024, 256
    T, XI = [-16.0, 15.0], [-3.0, 3.5]
    u = 0.8 * S.kdvv_sech(D, T)
    for four, two in (("4SPLIT4A", "2SPLIT4A"), ("4SPLIT4B", "2SPLIT4B")):
        rc, a = capi.fnft_kdvv(u, T, M, XI, discretization=four)
        rc2, b = capi.fnft_kdvv(u, T, M, XI, discretization=two)
        assert rc == 0 and rc2 == 0 and np.array_equal(a, b)
        rc3, ref = oracle.fnft_kdvv(u, T, M, XI, two)
        assert rc3 == 0 and S.rel_err(a, ref) < 2e-11
        eps_t = (T[1] - T[0]) / (D - 1)
        r1 = capi.kdv_fscatter(u, eps_t, four)
        r2 = capi.kdv_fscatter(u, eps_t, two)
        assert r1[0] == 0 and r1[1] == r2[1] and np.array_equal(r1[2], r2[2]) and r1[3] == r2[3]


def test_kdvv_argument_errors(capi):
    capi.silence_errors()
    u = np.ones(8, np.complex128)
    assert capi.fnft_kdvv(u[:1], [0, 1], 4, [-1, 1])[0] == 2
    assert capi.fnft_kdvv(u, [1, 0], 4, [-1, 1])[0] == 2
    assert capi.fnft_kdvv(u, [0, 1], 4, [-1, 1], want_contspec=False)[0] == 2
    assert capi.fnft_kdvv(u, [0, 1], 4, [1, -1])[0] == 2
    assert capi.fnft_kdvv(u, [0, 1], 4, [-1, 1], K=3)[0] == capi.FNFT_EC_NOT_YET_IMPLEMENTED
    assert capi.fnft_kdvv(u, [0, 1], 4, [-1, 1], discretization="BO")[0] == capi.FNFT_EC_NOT_YET_IMPLEMENTED


def test_kdvv_cfg5_full_size(capi, oracle, fixtures):
    """BASELINE.json configs[4]: fnft_kdvv, D = 2^18, M ~ 2^18, 2SPLIT8B (degree 12: the top product of the
    tree is a 2^22-point transform, column size 2048).  The 16 analytic points of
    fnft__kdvv_testcases.c:113-131 are the grid points k*(M-1)/15; the discretization error is far
    below the round-off floor at this D, so the bound is the floor (conditioning of a degree-3.1e6
    polynomial in double).  Plus a subsampled comparison with the oracle on the same grid."""
    import torch
    D = 1 << 18
    M = 15 * 17475 + 1   # 262126: the largest grid below 2^18 that contains the 16 analytic points
    u, T, XI, M16, exact = S.kdvv_case(fixtures, "SECH", D)
    plan = capi.KdvvPlan(D, M, batch=1, discretization="2SPLIT8B")
    du = torch.from_numpy(u).cuda()
    out = torch.zeros(M, dtype=torch.complex128, device="cuda")
    rc = plan.contspec_device(du.data_ptr(), out.data_ptr(), T, XI)
    assert rc == 0, capi.last_error()
    assert plan.finish() == 0
    cs = out.cpu().numpy()
    idx = np.arange(16) * ((M - 1) // 15)
    assert (M - 1) % 15 == 0
    assert S.rel_err(cs[idx], exact) < 1e-7
    plan.close()
    # the oracle on the same grid (about 30 s of CPU: the degree-3.1e6 tree).  On a coarser sub-grid the two agree
    # only to 4e-8: the chirp phases W^(n^2/2), n up to 3.1e6, lose accuracy with the larger grid step in either
    # implementation -- the same floor the analytic bound above allows for.
    rc2, ref = oracle.fnft_kdvv(u, T, M, XI, "2SPLIT8B")
    assert rc2 == 0
    assert S.rel_err(cs, ref) < 5e-10   # measured 7.5e-12 (round 1 bench leg)


# ---- discrete spectrum (bound states, norming constants, residues) -------------------------------
def _ds_cases():
    return [c for c in _analytic_cases() if c.values[0]["testcase"] == "SECH_FOCUSING"]


@pytest.mark.parametrize("b", _ds_cases())
def test_fnft_nsev_discrete_spectrum_bounds(capi, fixtures, b):
    """Entries [3..5] of the reference's bounds (Hausdorff distance of the bound states, norming
    constants, residues; fnft__nsev_testcases.c:650-700) for every harness call of the file, with the
    harness's options: SUBSAMPLE_AND_REFINE, FULL filtering, 10 Newton steps, dstype BOTH, K = deg*D."""
    fx = fixtures["nsev_sech_focusing"]
    ex = [S.l2c(fx[k]) for k in ("bound_states", "normconsts", "residues")]
    for st in b["stages"]:
        # per-call options of the file: opts.Dsub / opts.niter (fnft_nsev_test_adaptable_subsampling_factor.c:43-54)
        out = capi.fnft_nsev_ds(S.sech_focusing(st["D"]), fx["T"], discretization=b["discretization"],
                                richardson=bool(st["richardson"]), M=fx["M"], XI=fx["XI"],
                                Dsub=st["Dsub"] or 0, niter=10 if st["niter"] is None else st["niter"])
        rc, bs, nc, res = out[:4]
        assert rc == 0, capi.last_error()
        if not any(np.isfinite(x) for x in st["bounds_ds"]):
            continue   # fnft_nsev_test_nonregression_1.c (D = 126): the call has to succeed, nothing else is checked
        assert bs.size == 3, (st, bs)
        errs = S.ds_errors(bs, nc, res, *ex)
        for e, bound in zip(errs, st["bounds_ds"]):
            if np.isfinite(bound):
                assert e <= bound, (st, errs, b["file"])


@pytest.mark.parametrize("D,disc,bsloc", [(1024, "2SPLIT4B", "SUBSAMPLE_AND_REFINE"), (1000, "2SPLIT2_MODAL", "FAST_EIGENVALUE"),
                                          (512, "4SPLIT4B", "SUBSAMPLE_AND_REFINE"), (2048, "2SPLIT2A", "NEWTON"),
                                          (1024, "2SPLIT3A", "SUBSAMPLE_AND_REFINE")])
def test_discrete_spectrum_vs_oracle(capi, oracle, D, disc, bsloc):
    T = [-25.0, 25.0]
    q = S.sech_focusing(D) * np.exp(0.4j * S.tgrid(T, D))   # eigenvalues shifted off the imaginary axis
    guesses = np.array([-0.2 + 0.6j, -0.2 + 1.8j, -0.2 + 2.6j]) if bsloc == "NEWTON" else None
    rc, bs, nc, res = capi.fnft_nsev_ds(q, T, discretization=disc, bsloc=bsloc, guesses=guesses)
    assert rc == 0, capi.last_error()
    rc2, bs_o, nc_o, res_o = oracle.fnft_nsev_ds(q, T, disc, bsloc=bsloc, guesses=guesses)
    assert rc2 == 0 and bs.size == bs_o.size == 3, (bs, bs_o)
    tol = 1e-6 if bsloc == "FAST_EIGENVALUE" else 1e-10
    eps_t = (T[1] - T[0]) / (D - 1)
    rcp, qp, _, _ = oracle.preprocess(q, eps_t, D, disc)
    rcs, a_o, _, _ = oracle.scatter_bound_states(qp, T, bs_o, 2 if disc.startswith("4SPLIT") else 1, skip_b=True)
    for j in range(3):
        i = int(np.argmin(np.abs(bs - bs_o[j])))
        assert abs(bs[i] - bs_o[j]) < tol
        # b = phi/psi does not depend on the grid point only at a zero of a: compare where Newton converged
        if bsloc != "FAST_EIGENVALUE" and abs(a_o[j]) < 1e-9:
            assert abs(nc[i] - nc_o[j]) < 1e-8 * abs(nc_o[j])
            assert abs(res[i] - res_o[j]) < 1e-8 * abs(res_o[j])


def test_discrete_spectrum_cfg4_full_size(capi, fixtures):
    """BASELINE.json configs[3]: D = 2^20, default options (2SPLIT4B, SUBSAMPLE_AND_REFINE: roots of a
    degree-40960 polynomial, Newton on all 2^20 samples), eigenvalues 0.7i, 1.7i, 2.7i."""
    fx = fixtures["nsev_sech_focusing"]
    D = 1 << 20
    rc, bs, nc, res = capi.fnft_nsev_ds(S.sech_focusing(D), fx["T"], discretization="2SPLIT4B")
    assert rc == 0, capi.last_error()
    ex = [S.l2c(fx[k]) for k in ("bound_states", "normconsts", "residues")]
    assert bs.size == 3, bs
    errs = S.ds_errors(bs, nc, res, *ex)
    assert errs[0] < 1e-9 and errs[1] < 1e-6 and errs[2] < 1e-6, errs


def test_reference_example_scenario(capi, oracle):
    """The call of examples/fnft_nsev_example.c:32-76 -- rectangular pulse q = 2 on T = [-1, 1], D = 256,
    M = 8, XI = [-2, 2], K = D, default options (2SPLIT4B, reflection coefficient, SUBSAMPLE_AND_REFINE,
    norming constants) -- through the drop-in entry point: one bound state at 1.57423i."""
    D, M = 256, 8
    q = np.full(D, 2.0 + 0j)
    T, XI = [-1.0, 1.0], [-2.0, 2.0]
    opts = capi.default_opts()
    K = D
    bs = np.zeros(K, np.complex128)
    nc = np.zeros(K, np.complex128)
    rc, cs = capi.fnft_nsev(q, T, M, XI, kappa=1, opts=opts, bound_states=bs, K=K, normconsts=nc)
    assert rc == 0, capi.last_error()
    k = capi.fnft_nsev.last_K
    assert k == 1
    assert abs(bs[0] - 1.57423j) < 1e-5
    rc2, ref = oracle.fnft_nsev(q, T, M, XI, kappa=1, disc="2SPLIT4B", cstype="RHO")
    assert rc2 == 0 and S.rel_err(cs, ref) < 1e-12
    rc3, bs_o, nc_o, _ = oracle.fnft_nsev_ds(q, T, "2SPLIT4B")
    assert rc3 == 0 and bs_o.size == 1
    assert abs(bs[0] - bs_o[0]) < 1e-11 and abs(nc[0] - nc_o[0]) < 1e-9 * abs(nc_o[0])


def test_xi_grid_slices_match_full_grid(capi):
    """SURVEY 8e-iii on one GPU: the slices a 4-rank job would evaluate, computed one after the other,
    against the transform on the full grid.  The chirp parameters A, V differ per slice, and the
    reference's evaluation points z_m = V^m / A drift off the unit circle by m*1.1e-16 (see
    test_batch_cfg3_shard), a drift the slices restart at their first point: the two agree within
    deg * M * 2.2e-16, not to the last bit."""
    from fnft_amd import sharding
    D, M = 4096, 1000
    T, XI = [-25.0, 25.0], [-1.4, 1.6]
    q = S.sech_focusing(D)
    rc, full = capi.fnft_nsev(q, T, M, XI, discretization="2SPLIT4B", contspec_type="BOTH")
    assert rc == 0
    full = full.reshape(3, M)
    for r in range(4):
        XI_r, M_r, lo = sharding.xi_shard(XI, M, 4, r)
        rc, part = capi.fnft_nsev(q, T, M_r, XI_r, discretization="2SPLIT4B", contspec_type="BOTH")
        assert rc == 0
        part = part.reshape(3, M_r)
        for j in range(3):
            assert S.rel_err(part[j], full[j, lo:lo + M_r]) < 2 * D * M * 2.2e-16


def test_sample_axis_blocks_match_whole_signal(capi, oracle):
    """SURVEY 8e-ii on one GPU: the block matrices an 8-rank (4-rank) job would build, one after the
    other through the fnft__nse_fscatter seam, multiplied on the root through fnft__poly_fmult2x2 and
    evaluated through fnft__poly_chirpz + the host epilogue, against fnft_nsev on the whole signal and
    against the oracle."""
    from fnft_amd import sharding
    T, XI = [-25.0, 25.0], [-1.4, 1.6]
    for D, M, G, disc, deg0 in ((1 << 14, 999, 8, "2SPLIT2_MODAL", 1), (6000, 512, 4, "2SPLIT4B", 2)):
        q = S.sech_focusing(D)
        eng = sharding.capi_sample_axis_engine(disc, 1, deg0)
        eps_t = (T[1] - T[0]) / (D - 1)
        Db = D // G
        blocks = [eng.subtree(q[g * Db:(g + 1) * Db], eps_t) for g in range(G)]
        tm, W = sharding.combine_block_matrices([b[0] for b in blocks], [b[1] for b in blocks], eng)
        assert tm.shape == (4, D * deg0 + 1)
        cs = sharding.contspec_from_transfer_matrix(tm, W, D, T, XI, M, eng).reshape(3, M)
        rc, full = capi.fnft_nsev(q, T, M, XI, discretization=disc, contspec_type="BOTH")
        assert rc == 0
        rc2, ref = oracle.fnft_nsev(q, T, M, XI, kappa=1, disc=disc, cstype="BOTH")
        assert rc2 == 0
        for j in range(3):
            assert S.rel_err(cs[j], full.reshape(3, M)[j]) < 2e-11
            assert S.rel_err(cs[j], ref.reshape(3, M)[j]) < 2e-11


def test_concurrent_host_threads(capi, oracle):
    """SURVEY 8b threading contract: the drop-in entry points may be called from several host threads
    at once (own opts per thread).  ctypes releases the GIL, so these calls really overlap; every
    result is checked against the oracle."""
    import threading
    T, XI = [-25.0, 25.0], [-1.4, 1.6]
    jobs = [(1024, 64, "2SPLIT4B"), (1000, 50, "2SPLIT2_MODAL"), (2048, 128, "2SPLIT3A"), (512, 33, "4SPLIT4B"),
            (1024, 64, "2SPLIT4B"), (777, 20, "2SPLIT2A")]
    out = [None] * len(jobs)

    def work(i):
        D, M, disc = jobs[i]
        q = S.sech_focusing(D, amp=2.0 + 0.2 * i)
        res = []
        for _ in range(3):
            rc, cs = capi.fnft_nsev(q, T, M, XI, discretization=disc, contspec_type="BOTH")
            res.append((rc, cs))
        rc_d, bs, nc, resd = capi.fnft_nsev_ds(q, T, discretization=disc) if i % 2 == 0 else (0, None, None, None)
        out[i] = (q, res, rc_d, bs)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    for i, (D, M, disc) in enumerate(jobs):
        q, res, rc_d, bs = out[i]
        rc2, ref = oracle.fnft_nsev(q, T, M, XI, kappa=1, disc=disc, cstype="BOTH")
        assert rc2 == 0
        for rc, cs in res:
            assert rc == 0, capi.last_error()
            assert S.rel_err(cs, ref) < 1e-11, (i, disc)
        assert rc_d == 0
        if bs is not None:
            rc3, bs_o, _, _ = oracle.fnft_nsev_ds(q, T, disc)
            assert rc3 == 0 and bs.size == bs_o.size
            for v in bs_o:
                assert np.min(np.abs(bs - v)) < 1e-9


def test_two_plans_two_streams_in_flight(capi):
    """Two device plans on two streams, transforms of different signals interleaved without synchronisation in between
    (what bench.py reports as two_in_flight): every result equals the one the same plan gives when it runs alone."""
    import torch
    D = M = 1 << 16
    T, XI = [-25.0, 25.0], [-1.4, 1.6]
    sig = [torch.from_numpy(S.sech_focusing(D, amp=2.0 + 0.3 * i)).cuda() for i in range(4)]
    plans = [capi.Plan(D, M, batch=1, discretization="2SPLIT4B") for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    alone = []
    for i in range(4):                                   # reference: one at a time
        out = torch.zeros(3 * M, dtype=torch.complex128, device="cuda")
        st = streams[i % 2].cuda_stream
        assert plans[i % 2].contspec_device(sig[i].data_ptr(), out.data_ptr(), T, XI, stream=st) == 0
        assert plans[i % 2].finish(st) == 0
        torch.cuda.synchronize()
        alone.append(out.cpu().numpy())
    outs = [torch.zeros(3 * M, dtype=torch.complex128, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    for rep in range(3):
        for i in range(4):                               # both streams busy at once, no host synchronisation
            st = streams[i % 2].cuda_stream
            assert plans[i % 2].contspec_device(sig[i].data_ptr(), outs[i].data_ptr(), T, XI, stream=st) == 0
    torch.cuda.synchronize()
    for i in range(2):
        assert plans[i].finish(streams[i].cuda_stream) == 0
    for i in range(4):
        assert np.array_equal(outs[i].cpu().numpy(), alone[i])   # same kernels, same order per plan: bit for bit
    for p in plans:
        p.close()


def test_discrete_spectrum_options(capi, oracle, fixtures):
    """discspec_type (norming constants / residues / both, fnft_nsev.c:946-964), filtering (NONE keeps the
    spurious roots, BASIC the upper half plane, FULL the bounding box, :615-650) and a bound_states buffer
    that is too small (:728-734: as many as fit, plus a warning through the printf hook)."""
    fx = fixtures["nsev_sech_focusing"]
    D = 1024
    q = S.sech_focusing(D)
    T = fx["T"]
    rc, bs, nc, res = capi.fnft_nsev_ds(q, T, discretization="2SPLIT4B", dstype="BOTH")
    assert rc == 0 and bs.size == 3
    rc, bs1, nc1, res1 = capi.fnft_nsev_ds(q, T, discretization="2SPLIT4B", dstype="NORMING_CONSTANTS")
    assert rc == 0 and res1 is None and np.allclose(bs1, bs, atol=1e-13) and np.allclose(nc1, nc, rtol=1e-12)
    rc, bs2, nc2, res2 = capi.fnft_nsev_ds(q, T, discretization="2SPLIT4B", dstype="RESIDUES")
    assert rc == 0 and nc2 is None and np.allclose(res2, res, rtol=1e-12)
    # filtering: raw roots of the a-polynomial of the (not subsampled) signal
    rc, bsn, _, _ = capi.fnft_nsev_ds(q, T, discretization="2SPLIT4B", bsloc="FAST_EIGENVALUE", bsfilt="NONE",
                                      dstype="NORMING_CONSTANTS")
    assert rc == 0 and bsn.size == 2 * D          # every root: deg * D
    rc, bsb, _, _ = capi.fnft_nsev_ds(q, T, discretization="2SPLIT4B", bsloc="FAST_EIGENVALUE", bsfilt="BASIC",
                                      dstype="NORMING_CONSTANTS")
    assert rc == 0 and 3 <= bsb.size < 2 * D and np.all(bsb.imag >= 0)
    rc, bsf, _, _ = capi.fnft_nsev_ds(q, T, discretization="2SPLIT4B", bsloc="FAST_EIGENVALUE", bsfilt="FULL",
                                      dstype="NORMING_CONSTANTS")
    assert rc == 0 and bsf.size == 3
    for v in (0.7j, 1.7j, 2.7j):
        assert np.min(np.abs(bsf - v)) < 1e-3 and np.min(np.abs(bsb - v)) < 1e-3 and np.min(np.abs(bsn - v)) < 1e-3
    # capacity 2 for 3 bound states
    import ctypes as C
    msgs = []
    CB = C.CFUNCTYPE(C.c_int32, C.c_char_p)

    @CB
    def hook(fmt):
        msgs.append(fmt)
        return 0
    L = capi.load()
    old = L.fnft_errwarn_getprintf()
    L.fnft_errwarn_setprintf(C.cast(hook, C.c_void_p))
    try:
        rc, bsk, nck, resk = capi.fnft_nsev_ds(q, T, discretization="2SPLIT4B", K=2)
    finally:
        L.fnft_errwarn_setprintf(old)
    assert rc == 0 and bsk.size == 2
    assert any(b"Warning" in m for m in msgs)


# ---- seams under the discrete spectrum / the 4SPLIT4 front end, with the reference's own vectors ---------
def _hausdorff(x, y):
    return max(max(min(abs(a - b) for b in y) for a in x), max(min(abs(a - b) for b in x) for a in y))


def test_misc_resample_golden(capi, oracle, fixtures):
    """fnft__misc_resample on the GPU (DFT of length 1256 by Bluestein, phase ramp, inverse DFT) with
    test/fnft__misc/fnft__misc_resample_test.c:28-66 (rel-L1 <= 3e-7 against the analytically shifted signal)
    and against the oracle."""
    f = fixtures["misc_resample"]
    D = f["D"]
    eps = f["span"] / (D - 1)
    t = f["T0"] + np.arange(D) * eps
    sig = lambda tt: f["amp"] / np.cosh(tt) * np.exp(1j * f["freq"] * tt)   # noqa: E731
    for delta in f["deltas"]:
        rc, qn = capi.misc_resample(sig(t), eps, delta)
        assert rc == 0, capi.last_error()
        assert S.rel_err(qn, sig(t + delta)) <= f["tol_rel_l1"]
        rc2, qo = oracle.misc_resample(sig(t), eps, delta)
        assert rc2 == 0 and S.rel_err(qn, qo) < 1e-12
    assert capi.misc_resample(sig(t)[:2], eps, 0.1)[0] == 2   # D <= 2: fnft__misc.c:331-332


def test_poly_roots_fasteigen_golden(capi, fixtures):
    """fnft__poly_roots_fasteigen (Ehrlich-Aberth kernels in place of eiscor) with
    test/fnft__poly/fnft__poly_roots_fasteigen_test.c:27-44 as a SET comparison (Hausdorff distance <= 100 eps),
    plus polynomials built from known roots: a cluster, a double root (coincident estimates must not freeze)
    and 600 roots on two circles."""
    f = fixtures["poly_roots_fasteigen"]
    rc, r = capi.poly_roots_fasteigen(S.l2c(f["p"]))
    assert rc == 0, capi.last_error()
    assert _hausdorff(r, S.l2c(f["roots_exact"])) <= f["tol_hausdorff"]
    rng = np.random.default_rng(5)
    known = np.concatenate([0.9 * np.exp(2j * np.pi * rng.random(50)), 1.3 * np.exp(2j * np.pi * rng.random(50))])
    rc, r = capi.poly_roots_fasteigen(np.poly(known))
    # conditioning of np.poly's coefficients: LAPACK's backward-stable roots of the same coefficients are 2.6e-8 away
    assert rc == 0 and _hausdorff(r, known) < 1e-7
    # 600 well-conditioned roots, known exactly: (z^300 - a)(z^300 - b), two circles of radius 0.9 and 1.3
    n = 300
    known = np.concatenate([0.9 * np.exp(2j * np.pi * (np.arange(n) + 0.3) / n),
                            1.3 * np.exp(2j * np.pi * (np.arange(n) + 0.1) / n)])
    a, b = (0.9 * np.exp(2j * np.pi * 0.3 / n)) ** n, (1.3 * np.exp(2j * np.pi * 0.1 / n)) ** n
    c = np.zeros(2 * n + 1, np.complex128)
    c[0], c[n], c[2 * n] = 1.0, -(a + b), a * b
    rc, r = capi.poly_roots_fasteigen(c)
    assert rc == 0 and _hausdorff(r, known) < 1e-12
    clustered = np.array([0.5 + 0.5j, 0.5 + 0.5j + 1e-4, 0.5 + 0.5j - 1e-4j, -0.3j, 2.0])
    rc, r = capi.poly_roots_fasteigen(np.poly(clustered))
    # three roots 1e-4 apart: sensitivity eps/sep^2 ~ 2e-8 (numpy.roots, backward stable: 2.6e-8).  The sweeps' worst-case
    # stopping bound leaves 1.7e-6; the two Newton steps that follow them (AberthParams::polish) reach the conditioning
    assert rc == 0 and _hausdorff(r, clustered) < 1e-7
    double = np.array([0.25 + 0.1j, 0.25 + 0.1j, -1.0, 0.7j])
    rc, r = capi.poly_roots_fasteigen(np.poly(double))
    assert rc == 0                                     # a double root: both estimates end inside its sqrt(eps) ball
    assert _hausdorff(r, double) < 1e-7


def test_scatter_bound_states_bo_golden(capi, oracle, fixtures):
    """fnft__nse_scatter_bound_states (BO) on the GPU with
    test/fnft__nse_scatter/fnft__nse_scatter_bound_states_test_bo.c:30-131 and against the oracle; for the
    tolerance on b see tests/test_oracle_golden.py::test_scatter_bound_states_bo_golden."""
    f = fixtures["nse_scatter_bound_states_bo"]
    D, T = f["D"], f["T"]
    eps = (T[1] - T[0]) / (D - 1)
    q = 3.0 / np.cosh(T[0] + np.arange(D) * eps) + 0j
    lam = S.l2c(f["bound_states"])
    rc, a, ap, b = capi.nse_scatter_bound_states(q, T, lam)
    assert rc == 0, capi.last_error()
    assert np.max(np.abs(a - S.l2c(f["a_vals"]))) < 1e-13
    assert S.rel_err(ap, S.l2c(f["aprime_vals"])) < 1e-12
    assert S.rel_err(b, S.l2c(f["b_vals"])) < 5e-3
    rc2, ao, apo, bo = oracle.scatter_bound_states(q, T, lam, 1)
    assert rc2 == 0
    # b = phi/psi is matching-point dependent at O(|a|) ~ 2e-5 where a != 0: chunked and sequential search may
    # settle on neighbouring grid points (measured 4.7e-6)
    assert np.max(np.abs(a - ao)) < 1e-14 and S.rel_err(ap, apo) < 1e-13 and S.rel_err(b, bo) < 5e-5
    assert capi.nse_scatter_bound_states(q, T, lam, discretization="CF4_3")[0] == 6   # not covered: says so


def test_scatter_bound_states_cf4_2_vs_oracle(capi, oracle):
    """fnft__nse_scatter_bound_states with CF4_2 through the exported symbol (the scatterer fnft_nsev refines with for
    4SPLIT4A/B, src/private/fnft__nse_scatter_bound_states.c:188-197): the preprocessed signal (two samples per grid
    point) of a sech pulse, a / a' / b against the oracle's sequential restatement."""
    D, T = 512, [-12.0, 12.0]
    q = S.sech_focusing(D, amp=3.2) if False else (3.2 / np.cosh(S.tgrid(T, D)) + 0j)
    eps_t = (T[1] - T[0]) / (D - 1)
    rc0, q_pre, Dsub, _ = oracle.preprocess(q, eps_t, D, "4SPLIT4B")
    assert rc0 == 0 and q_pre.size == 2 * D
    # at (near) eigenvalues: b = phi/psi does not depend on the matching point only where a = 0
    lam = np.array([0.7j, 1.7j, 2.7j])
    rc, a, ap, b = capi.nse_scatter_bound_states(q_pre, T, lam, discretization="CF4_2")
    assert rc == 0, capi.last_error()
    rc2, ao, apo, bo = oracle.scatter_bound_states(q_pre, T, lam, 2)
    assert rc2 == 0
    assert np.max(np.abs(a - ao)) < 1e-13 * max(1.0, np.max(np.abs(ao))) and S.rel_err(ap, apo) < 1e-12
    assert S.rel_err(b, bo) < 5e-5
    assert capi.nse_scatter_bound_states(q_pre[:-1], T, lam, discretization="CF4_2")[0] == 8   # odd D, :188-191


@pytest.mark.parametrize("deg,n", [(1, 4), (3, 5), (64, 8), (1000, 3), (4096, 2), (2048, 16)])
def test_poly_fmult2x2_device_vs_host_seam(capi, deg, n):
    """fnft_amd_poly_fmult2x2_device (factors and product in device memory: the root's step of a sample-axis split)
    against fnft__poly_fmult2x2 on the same matrices through host buffers: the same tree, so the same numbers."""
    import torch
    rng = np.random.default_rng(deg + n)
    p = (rng.standard_normal((4, n * (deg + 1))) + 1j * rng.standard_normal((4, n * (deg + 1)))) / np.sqrt(deg + 1)
    rc, d0, ref, W0 = capi.poly_fmult2x2(deg, n, p.copy())
    assert rc == 0 and d0 == n * deg
    dp = torch.from_numpy(p).cuda()
    out = torch.zeros((4, n * deg + 1), dtype=torch.complex128, device="cuda")
    rc, d1, W1 = capi.poly_fmult2x2_device(deg, n, dp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, capi.last_error()
    assert (d1, W1) == (d0, W0)
    assert np.array_equal(out.cpu().numpy(), ref)
    assert capi.poly_fmult2x2_device(0, n, dp.data_ptr(), out.data_ptr())[0] == 2


def test_release_cached_keeps_results(capi):
    """fnft_amd_release_cached: cached plans, the resident layer-peeling state and the cache of released device blocks go
    back to the driver; the next call rebuilds what it needs and returns the same numbers."""
    D, M = 4096, 512
    T, XI = [-25.0, 25.0], [-1.4, 1.6]
    q = S.sech_focusing(D)
    rc, a = capi.fnft_nsev(q, T, M, XI, discretization="2SPLIT4B", contspec_type="BOTH")
    assert rc == 0
    capi.release_cached(-1)
    rc, b = capi.fnft_nsev(q, T, M, XI, discretization="2SPLIT4B", contspec_type="BOTH")
    assert rc == 0 and np.array_equal(a, b)
    capi.release_cached(0)
    rc, c = capi.fnft_nsev(q, T, M, XI, discretization="2SPLIT4B", contspec_type="BOTH")
    assert rc == 0 and np.array_equal(a, c)


def test_contspec_from_transfer_matrix_and_div_by_zero(capi, oracle):
    """nsev_compute_contspec on a caller-supplied transfer matrix (fnft_amd_nsev_contspec_from_tm_device):
    (i) fed with the plan's own transfer matrix it reproduces the spectrum of the full call; (ii) a transfer
    matrix whose a-polynomial vanishes gives H11 == 0 at every grid point, the branch src/fnft_nsev.c:850-853
    -- FNFT_EC_DIV_BY_ZERO, wrapped to -3 like every error of a callee; contspec_type AB has no division and
    succeeds (:861-876)."""
    import torch
    D, M = 1024, 64
    T, XI = [-25.0, 25.0], [-1.4, 1.6]
    q = torch.from_numpy(S.sech_focusing(D)).cuda()
    for disc, deg0 in (("2SPLIT2_MODAL", 1), ("2SPLIT4B", 2)):
        plan = capi.Plan(D, M, batch=1, discretization=disc)
        out = torch.zeros(3 * M, dtype=torch.complex128, device="cuda")
        assert plan.contspec_device(q.data_ptr(), out.data_ptr(), T, XI) == 0 and plan.finish() == 0
        tm = torch.zeros(4 * (D * deg0 + 1), dtype=torch.complex128, device="cuda")
        rc, deg, W = plan.transfer_matrix_device(tm.data_ptr())
        assert rc == 0 and deg == D * deg0
        out2 = torch.zeros(3 * M, dtype=torch.complex128, device="cuda")
        assert plan.contspec_from_tm_device(tm.data_ptr(), W, out2.data_ptr(), T, XI) == 0 and plan.finish() == 0
        assert S.rel_err(out2.cpu().numpy(), out.cpu().numpy()) < 1e-13
        rc_o, ref = oracle.fnft_nsev(S.sech_focusing(D), T, M, XI, kappa=1, disc=disc, cstype="BOTH")
        assert rc_o == 0 and S.rel_err(out2.cpu().numpy(), ref) < 1e-12
        # a(z) == 0: entry 11 zeroed
        tm0 = tm.clone()
        tm0[: D * deg0 + 1] = 0
        assert plan.contspec_from_tm_device(tm0.data_ptr(), W, out2.data_ptr(), T, XI, "BOTH") == 0
        assert plan.finish() == -3          # -FNFT_EC_DIV_BY_ZERO
        assert plan.contspec_from_tm_device(tm0.data_ptr(), W, out2.data_ptr(), T, XI, "REFLECTION_COEFFICIENT") == 0
        assert plan.finish() == -3
        assert plan.contspec_from_tm_device(tm0.data_ptr(), W, out2.data_ptr(), T, XI, "AB") == 0
        assert plan.finish() == 0           # no division in a, b
        plan.close()


def test_root_finalize_on_the_chirp_kernel_or_on_its_own(capi):
    """The root of a tree that ends in split levels is finalized (64 maxima -> pending scale, exponent) either by the chirp
    transform's first kernel (a transform with a spectrum) or by a launch of its own (a consumer that comes first: here the
    export of the transfer matrix after a tree-only pass): both orders give the same matrices, exponents and spectra,
    for every signal of a batch (src/private/fnft__poly_fmult.c:330-374,493: a and W of the last level)."""
    import torch
    D, M, B = 1 << 14, 256, 3
    T, XI = [-25.0, 25.0], [-1.4, 1.6]
    qh = np.stack([S.sech_focusing(D, amp=3.2 - 0.3 * b) for b in range(B)])
    q = torch.from_numpy(qh).cuda()
    for disc, deg0 in (("2SPLIT2_MODAL", 1), ("2SPLIT4B", 2)):
        p1 = capi.Plan(D, M, batch=B, discretization=disc)
        out = torch.zeros(B * 3 * M, dtype=torch.complex128, device="cuda")
        assert p1.contspec_device(q.data_ptr(), out.data_ptr(), T, XI) == 0 and p1.finish() == 0
        p2 = capi.Plan(D, M, batch=B, discretization=disc)
        assert p2.contspec_device(q.data_ptr(), 0, T, XI) == 0 and p2.finish() == 0      # tree only
        for b in range(B):
            rc1, d1, tm1, W1 = p1.transfer_matrix(b)     # after the chirp kernel finalized the root
            rc2, d2, tm2, W2 = p2.transfer_matrix(b)     # KFinalizeScales in front of the export
            assert rc1 == 0 and rc2 == 0 and d1 == d2 == D * deg0 and W1 == W2 and W1 != 0
            assert np.array_equal(tm1, tm2)
            tmd = torch.from_numpy(tm2.reshape(-1)).cuda()
            o2 = torch.zeros(3 * M, dtype=torch.complex128, device="cuda")
            p3 = capi.Plan(D, M, batch=1, discretization=disc)
            assert p3.contspec_from_tm_device(tmd.data_ptr(), W2, o2.data_ptr(), T, XI) == 0 and p3.finish() == 0
            assert S.rel_err(o2.cpu().numpy(), out.cpu().numpy()[b * 3 * M:(b + 1) * 3 * M]) < 1e-13
            p3.close()
        p1.close()
        p2.close()


def test_discrete_spectrum_cfg4_normconsts_vs_oracle(capi, oracle, fixtures):
    """BASELINE.json configs[3] at full size: the norming constants and residues the GPU returns at D = 2^20 against
    the ORACLE's slow scatterer (BO, all 2^20 samples, sequential) evaluated at the GPU's own bound states:
    b = phi/psi, residue = b/a'.  (The oracle's root finder is not run at this size; the bound states themselves are
    pinned by the exact eigenvalues in test_discrete_spectrum_cfg4_full_size.)"""
    fx = fixtures["nsev_sech_focusing"]
    D = 1 << 20
    q = S.sech_focusing(D)
    rc, bs, nc, res = capi.fnft_nsev_ds(q, fx["T"], discretization="2SPLIT4B")
    assert rc == 0 and bs.size == 3, (rc, bs)
    rc2, a, ap, b = oracle.scatter_bound_states(q, fx["T"], bs, 1)
    assert rc2 == 0
    assert np.max(np.abs(a)) < 1e-8                       # Newton converged onto zeros of the discrete a
    assert S.rel_err(nc, b) < 1e-9
    assert S.rel_err(res, b / ap) < 1e-9


# ---- scalar products and single pair products (the calls fnft__nse_finvscatter.c:128,155 make) -------------
@pytest.mark.parametrize("key", ["fmult_pow2", "fmult_nopow2"])
@pytest.mark.parametrize("normalize", [False, True])
def test_poly_fmult_scalar_golden(capi, fixtures, key, normalize):
    """test/fnft__poly/fnft__poly_fmult_test_n_is_power_of_2.c:26-77, ..._no_power_of_2.c:26-88 (100 eps; W != 0 when
    normalising)."""
    fx = fixtures[key]
    deg, n = fx["deg"], fx["n"]
    i = np.arange((deg + 1) * n, dtype=np.float64)
    p = np.sqrt(i + 1.0) * (np.cos(i) + 1j * np.sin(-2.0 * i))
    rc, d, res, W = capi.poly_fmult(deg, n, p, normalize=normalize)
    assert rc == 0, capi.last_error()
    exact = S.l2c(fx["result_exact"])
    assert d == exact.size - 1
    if normalize:
        assert W != 0
        res = res * 2.0 ** W
    assert S.rel_err(res, exact) <= fx["tol_rel_l1"]


def test_poly_fmult_two_polys_modes(capi):
    """fnft__poly_fmult_two_polys in the call patterns of fnft__poly_fmult_two_polys2x2 (src/private/
    fnft__poly_fmult.c:288-324): mode 0 (plain product), mode 2 then mode 3 with a NULL factor (partial sum kept in
    `result`, first factor reused), mode 1 (accumulate) -- against numpy's convolution; and the 2x2 pair product
    against the four entry-wise sums of convolutions."""
    rng = np.random.default_rng(11)
    deg = 37
    mk = lambda: rng.standard_normal(deg + 1) + 1j * rng.standard_normal(deg + 1)   # noqa: E731
    a, b, c = mk(), mk(), mk()
    wr = 2 * deg + 1
    rc, res, bufs = capi.poly_fmult_two_polys(a, b, mode=0)
    assert rc == 0, capi.last_error()
    assert S.rel_err(res[:wr], np.convolve(a, b)) < 1e-14
    rc, res, bufs = capi.poly_fmult_two_polys(a, b, mode=2)                        # partial: a*b
    rc2, res, bufs = capi.poly_fmult_two_polys(None, c, result=res, mode=3, bufs=bufs)   # + a*c (a from the last call)
    assert rc == 0 and rc2 == 0
    assert S.rel_err(res[:wr], np.convolve(a, b) + np.convolve(a, c)) < 1e-14
    rc, res, bufs = capi.poly_fmult_two_polys(b, c, result=res, mode=1, bufs=bufs)
    assert rc == 0
    assert S.rel_err(res[:wr], np.convolve(a, b) + np.convolve(a, c) + np.convolve(b, c)) < 1e-14
    P1 = np.stack([mk() for _ in range(4)])
    P2 = np.stack([mk() for _ in range(4)])
    rc, R = capi.poly_fmult_two_polys2x2(P1, P2)
    assert rc == 0, capi.last_error()
    cv = np.convolve
    ref = np.stack([cv(P1[0], P2[0]) + cv(P1[1], P2[2]), cv(P1[0], P2[1]) + cv(P1[1], P2[3]),
                    cv(P1[2], P2[0]) + cv(P1[3], P2[2]), cv(P1[2], P2[1]) + cv(P1[3], P2[3])])
    assert S.rel_err(R.ravel(), ref.ravel()) < 1e-14
    assert capi.load().fnft__poly_fmult_two_polys_len(4) == 9 and capi.load().fnft__poly_fmult_two_polys_len(16) == 36


# ---- inverse scattering by layer peeling: the other caller of the pair-product kernels -------
def _finv_cases():
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "reference_fixtures.json")) as f:
        return [pytest.param(c, id=c["file"].replace("fnft__nse_finvscatter_test_", "").replace(".c", ""))
                for c in json.load(f)["nse_finvscatter"]["cases"]]


@pytest.mark.parametrize("c", _finv_cases())
def test_nse_finvscatter_round_trip(capi, oracle, fixtures, c):
    """test/fnft__nse_finvscatter/fnft__nse_finvscatter_test.inc:28-75 on the GPU library: fnft__nse_fscatter (W_ptr =
    NULL) then fnft__nse_finvscatter returns the D = 16384 samples within the file's bound; and the samples
    recovered from the ORACLE's transfer matrix agree with the oracle's own inverse."""
    from oracle.oracle import nse_finvscatter
    f = fixtures["nse_finvscatter"]
    D, eps_t = f["D"], f["eps_t"]
    i = np.arange(D)
    q_exact = ((i + 1) / (D + 1) / D) * np.exp(1j * i / D)
    rc, deg, tm, _ = capi.nse_fscatter(q_exact, eps_t, c["kappa"], c["discretization"], normalize=False)
    assert rc == 0 and deg == D, capi.last_error()
    rc2, q = capi.nse_finvscatter(tm, eps_t, c["kappa"], c["discretization"])
    assert rc2 == 0, capi.last_error()
    assert S.rel_err(q, q_exact) < c["bound_eps"] * 2.220446049250313e-16
    rc3, _, tm_o, _ = oracle.nse_fscatter(q_exact, eps_t, c["kappa"], c["discretization"], normalize=False)
    rc4, q_o = nse_finvscatter(tm_o, eps_t, c["kappa"], c["discretization"])
    rc5, q_g = capi.nse_finvscatter(tm_o, eps_t, c["kappa"], c["discretization"])
    assert rc3 == 0 and rc4 == 0 and rc5 == 0
    assert S.rel_err(q_g, q_o) < 2 * c["bound_eps"] * 2.220446049250313e-16


def test_nse_finvscatter_arguments(capi):
    tm = np.ones((4, 9), np.complex128)
    assert capi.nse_finvscatter(tm, 0.1, 2, "2SPLIT2_MODAL")[0] == 2       # kappa
    assert capi.nse_finvscatter(tm, -0.1, 1, "2SPLIT2_MODAL")[0] == 2      # eps_t
    assert capi.nse_finvscatter(np.ones((4, 7), np.complex128), 0.1, 1, "2SPLIT2_MODAL")[0] == 5   # D = 6: no power of two
    assert capi.nse_finvscatter(tm, 0.1, 1, "2SPLIT4B")[0] in (2, 5)       # no base case for this scheme


# ---- sizes beyond round 1's D*deg <= 2^22: column transforms of 4096 and 8192, chirp length 2^24 ---------------
@pytest.mark.parametrize("log2D,disc,bound", [(22, "2SPLIT2_MODAL", 2e-7), (21, "2SPLIT4B", 5e-8), (20, "2SPLIT8B", 2e-6),
                                              (20, "2SPLIT6B", 2e-6)])
def test_beyond_2p22_analytic(capi, log2D, disc, bound):
    """The reference has no size limit (fnft__poly_fmult.c:448-455 just mallocs): D = 2^22 MODAL (top product 2^22
    points), D = 2^21 with the default 2SPLIT4B, and D = 2^20 with the degree-12 and degree-6 schemes (top product of
    2^24 points: column transforms of 8192, chirp length 2^24), against the closed-form Satsuma-Yajima spectrum on a
    grid of 2^16 points, plus |a|^2 + |b|^2 = 1.  Bounds: MODAL at 2^22 has a discretization error of
    5e-3 * (4096/D)^2 = 5e-9 in a (measured 3.8e-9) and sits at the round-off floor of 4 million factors in b and rho
    (measured 3.5e-8); the higher-order schemes at the conditioning floor of their coefficient form (cancellation
    in the per-sample formulas at eps_t ~ 5e-5)."""
    D, M = 1 << log2D, 1 << 16
    T, XI = [-25.0, 25.0], [-7.0 / 5.0, 8.0 / 5.0]
    q = S.sech_focusing(D)
    rc, cs = capi.fnft_nsev(q, T, M, XI, kappa=1, discretization=disc, contspec_type="BOTH")
    assert rc == 0, capi.last_error()
    xi = XI[0] + np.arange(M) * (XI[1] - XI[0]) / (M - 1)
    a, b = S.sech_focusing_analytic(xi)
    errs = (S.rel_err(cs[M:2 * M], a), S.rel_err(cs[2 * M:], b), S.rel_err(cs[:M], b / a))
    assert max(errs) < bound, errs
    inv = np.abs(cs[M:2 * M]) ** 2 + np.abs(cs[2 * M:]) ** 2
    assert np.max(np.abs(inv - 1.0)) < 100 * bound


def test_not_bandlimited_warning(capi):
    """src/private/fnft__misc.c:371-381: the resampler of the 4SPLIT4A/B front end warns (a message, no return code) when
    the 5 % bands next to the Nyquist bin hold more than sqrt(eps) of the spectrum's l2 norm -- a rectangular pulse does,
    the smooth sech pulse does not; the stand-alone fnft__misc_resample seam says the same."""
    import ctypes as C
    L = capi.load()
    seen = []
    CB = C.CFUNCTYPE(C.c_int32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p)

    def hook(fmt, msg, func, line, a, b, c, suffix):
        seen.append((fmt, msg, func))
        return 0

    cb = CB(hook)
    L.fnft_errwarn_setprintf(C.cast(cb, C.c_void_p))
    try:
        D, M = 1024, 32
        T, XI = [-25.0, 25.0], [-1.4, 1.6]
        rect = np.where(np.abs(S.tgrid(T, D)) < 3.0, 1.5 + 0j, 0j)
        rc, _ = capi.fnft_nsev(S.sech_focusing(D), T, M, XI, kappa=1, discretization="4SPLIT4B", contspec_type="BOTH")
        assert rc == 0 and not seen                       # smooth pulse: band-limited to working accuracy
        rc, _ = capi.fnft_nsev(rect, T, M, XI, kappa=1, discretization="4SPLIT4B", contspec_type="BOTH")
        assert rc == 0                                    # a warning is not an error
        assert len(seen) == 1 and seen[0][0].startswith(b"FNFT Warning: %s")
        assert b"does not appear to be bandlimited" in seen[0][1] and seen[0][2] == b"fnft__misc_resample"
        rc, _ = capi.fnft_nsev(rect, T, M, XI, kappa=1, discretization="2SPLIT4B", contspec_type="BOTH")
        assert rc == 0 and len(seen) == 1                 # no resampling, no warning
        rc, _ = capi.misc_resample(rect, (T[1] - T[0]) / (D - 1), 0.01)
        assert rc == 0 and len(seen) == 2
    finally:
        L.fnft_errwarn_setprintf(None)
