"""Pin the CPU oracle against the reference's own known-answer vectors (no GPU needed).

Each test mirrors one of the reference's tests and uses the same tolerance:
  test/fnft__poly/fnft__poly_fmult2x2_test_n_is_power_of_2.c / ..._no_power_of_2.c   (100 eps)
  test/fnft__poly/fnft__poly_chirpz_test.c                                            (100 eps)
  test/fnft__fft_wrapper/fnft__fft_wrapper_test.c
  test/fnft__akns_fscatter/*.c                                                        (100 eps)
  test/fnft_nsev/*.c via src/private/fnft__nsev_testcases.c:711-822 (per-scheme error bounds)
"""
import numpy as np
import pytest

import signals as S
from oracle.oracle import AKNS_DISC, NSE_DISC

EPS = np.finfo(np.float64).eps


def test_next_fast_size(oracle):
    # kiss_fft.c:396-408; schedule quoted in SURVEY App. B
    assert [oracle.next_fast_size(n) for n in (3, 5, 9, 17, 33, 7, 11, 1)] == [3, 5, 9, 18, 36, 8, 12, 1]
    assert oracle.next_fast_size(2 ** 20 + 1) == 1049760
    assert oracle.next_fast_size(2 ** 21) == 2097152


def test_fft_known_answer(oracle, fixtures):
    fx = fixtures["fft_wrapper"]
    x, y = S.l2c(fx["in_exact"]), S.l2c(fx["out_exact"])
    assert S.rel_err(oracle.fft(x, -1), y) < 100 * EPS
    assert S.rel_err(oracle.fft(y, +1) / x.size, x) < 100 * EPS


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16, 18, 20, 25, 27, 30, 36, 45,
                               60, 64, 81, 100, 125, 128, 360, 1000, 1024, 1080])
def test_fft_matches_numpy(oracle, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    assert np.allclose(oracle.fft(x, -1), np.fft.fft(x), rtol=0, atol=1e-12 * np.sqrt(n) * 4)
    assert np.allclose(oracle.fft(x, +1), np.fft.ifft(x) * n, rtol=0, atol=1e-12 * np.sqrt(n) * 4)


@pytest.mark.parametrize("key", ["fmult2x2_pow2", "fmult2x2_nopow2"])
@pytest.mark.parametrize("normalize", [False, True])
def test_poly_fmult2x2_golden(oracle, fixtures, key, normalize):
    fx = fixtures[key]
    deg, n = fx["deg"], fx["n"]
    exact = S.l2c(fx["result_exact"])
    p = S.fmult_test_input(deg, n)
    d, res, W = oracle.poly_fmult2x2(deg, n, p, normalize=normalize)
    assert d == exact.size // 4 - 1
    if normalize:
        assert W != 0
        res = res * 2.0 ** W
    assert S.rel_err(res.ravel(), exact) <= fx["tol_rel_l1"]


@pytest.mark.parametrize("deg,n", [(1, 2), (1, 7), (2, 8), (2, 13), (3, 6), (4, 5), (1, 64), (2, 100)])
def test_poly_fmult2x2_vs_direct_convolution(oracle, deg, n):
    rng = np.random.default_rng(100 * deg + n)
    p = rng.standard_normal((4, n * (deg + 1))) + 1j * rng.standard_normal((4, n * (deg + 1)))
    d, res, W = oracle.poly_fmult2x2(deg, n, p, normalize=True)
    ref = S.tree_direct(p, deg, n)
    assert d == deg * n
    assert S.rel_err((res * 2.0 ** W).ravel(), ref.ravel()) < 1e-13


def test_poly_chirpz_golden(oracle, fixtures):
    fx = fixtures["chirpz"]
    p = S.l2c(fx["p"])
    A = complex(*fx["A"])
    W = np.exp(1j * fx["W_arg"])
    for M, key in ((3, "result_M3"), (6, "result_M6")):
        out = oracle.poly_chirpz(p, A, W, M)
        assert S.rel_err(out, S.l2c(fx[key])) <= fx["tol_rel_l1"]


@pytest.mark.parametrize("scheme", [s for s in sorted(AKNS_DISC) if s.startswith("2SPLIT")])
@pytest.mark.parametrize("normalize", [False, True])
def test_akns_fscatter_golden(oracle, fixtures, scheme, normalize):
    fx = fixtures["akns_fscatter"]["schemes"][scheme]
    q, r, z = S.akns_test_signal(fx["D"])
    rc, deg, tm, W = oracle.akns_fscatter(q, r, fx["eps_t"], scheme, normalize=normalize)
    assert rc == 0
    if normalize:
        assert W != 0
        tm = tm * 2.0 ** W
    vals = np.concatenate([oracle.poly_eval(tm[e], z) for e in range(4)])
    assert S.rel_err(vals, S.l2c(fx["result_exact"])) <= fx["tol_rel_l1"]  # err_bnd of the reference test of this scheme


def _nsev_errors(oracle, fixtures, testcase, disc, D, richardson=False):
    if testcase == "SECH_FOCUSING":
        fx = fixtures["nsev_sech_focusing"]
        q = S.sech_focusing(D)
        exact_rho = S.l2c(fx["contspec"])
        exact_ab = S.l2c(fx["ab"])
    elif testcase == "SECH_DEFOCUSING":
        fx = fixtures["nsev_sech_defocusing"]
        q = S.sech_defocusing(D)
        exact_rho = S.l2c(fx["contspec"])
        exact_ab = None
    else:
        fx = fixtures["nsev_truncated_soliton"]
        q = S.truncated_soliton(D)
        exact_rho = S.truncated_soliton_contspec(fx["XI"], fx["M"])
        exact_ab = None
    M = fx["M"]
    rc, cs = oracle.fnft_nsev(q, fx["T"], M, fx["XI"], kappa=fx["kappa"], disc=disc, cstype="BOTH",
                              richardson=richardson)
    assert rc == 0
    errs = [S.rel_err(cs[:M], exact_rho)]
    if exact_ab is not None:
        errs += [S.rel_err(cs[M:2 * M], exact_ab[:M]), S.rel_err(cs[2 * M:], exact_ab[M:])]
    return errs


# slow (non-polynomial) discretizations: outside the fast path this repo covers
SLOW_SCHEMES = ("BO", "CF4_2", "CF4_3", "CF5_3", "CF6_4", "ES4", "TES4")


def _bound_cases(fixtures_path="tests/golden/reference_fixtures.json"):
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, fixtures_path)) as f:
        fx = json.load(f)
    cases = []
    for b in fx["nsev_error_bounds"]:
        # every file of test/fnft_nsev/ is either replayed or named here with the reason it is not
        if b["discretization"] not in AKNS_DISC:
            assert b["discretization"] in SLOW_SCHEMES, "fixture file %s would be skipped silently" % b["file"]
            continue
        if b["testcase"] == "SECH_FOCUSING2":
            continue   # second focusing test case (fnft__nsev_testcases.c:289-461): not extracted
        assert b["stages"], b["file"]
        cases.append(pytest.param(b, id=b["file"].replace("fnft_nsev_test_", "").replace(".c", "")))
    return cases


@pytest.mark.parametrize("b", _bound_cases())
def test_fnft_nsev_analytic_bounds(oracle, fixtures, b):
    """fnft__nsev_testcases.c:711-822 driven exactly as test/fnft_nsev/<file>.c drives it: every
    harness call of the file (D, D+1, D-1, 2D with rescaled bounds, and the Richardson stages
    where the file has them) is one entry of b["stages"]."""
    assert b["stages"]
    for st in b["stages"]:
        errs = _nsev_errors(oracle, fixtures, b["testcase"], b["discretization"], st["D"],
                            richardson=bool(st["richardson"]))
        assert all(np.isfinite(e) for e in errs), (st, errs, b["file"])   # files with infinite bounds still have to run
        for e, bound in zip(errs, st["bounds"]):
            if np.isfinite(bound):
                assert e <= bound, (st, errs, b["file"])


def test_richardson_improves(oracle, fixtures):
    plain = _nsev_errors(oracle, fixtures, "SECH_FOCUSING", "2SPLIT4A", 4096)
    rich = _nsev_errors(oracle, fixtures, "SECH_FOCUSING", "2SPLIT4A", 4096, richardson=True)
    assert rich[0] < 0.1 * plain[0]


def test_modal_defocusing_step_check(oracle):
    """fnft__akns_fscatter.c:122-126: MODAL raises E_OTHER (5) when Re q == Re r and eps_t*|q| >= 1;
    fnft_nsev returns it wrapped as -5 (fnft__errwarn.h:50-57,101)."""
    q = np.full(16, 40.0 + 0j)
    rc, _ = oracle.fnft_nsev(q, [0.0, 1.0], 8, [-1.0, 1.0], kappa=-1, disc="2SPLIT2_MODAL")
    assert rc == -5


def test_validation_codes(oracle):
    q = np.ones(8, np.complex128)
    assert oracle.fnft_nsev(q[:1], [0, 1], 4, [-1, 1])[0] == 2          # D < 2
    assert oracle.fnft_nsev(q, [1, 0], 4, [-1, 1])[0] == 2              # T
    assert oracle.fnft_nsev(q, [0, 1], 4, [1, -1])[0] == 2              # XI
    assert oracle.fnft_nsev(q, [0, 1], 4, [-1, 1], kappa=2)[0] == 2     # kappa
    assert oracle.fnft_nsev(q, [0, 1], 4, [-1, 1], disc=NSE_DISC["BO"])[0] == 2        # slow scheme: not covered


def test_cstype_layouts(oracle):
    q = S.sech_focusing(256)
    T, XI, M = [-25.0, 25.0], [-1.4, 1.6], 16
    _, both = oracle.fnft_nsev(q, T, M, XI, disc="2SPLIT4B", cstype="BOTH")
    _, rho = oracle.fnft_nsev(q, T, M, XI, disc="2SPLIT4B", cstype="RHO")
    _, ab = oracle.fnft_nsev(q, T, M, XI, disc="2SPLIT4B", cstype="AB")
    assert np.array_equal(both[:M], rho) and np.array_equal(both[M:], ab)
    # rho = b/a up to the phase factors, independent of normalisation
    _, nonorm = oracle.fnft_nsev(q, T, M, XI, disc="2SPLIT4B", cstype="BOTH", normalize=False)
    assert S.rel_err(nonorm, both) < 1e-12


# ---- fnft_kdvv: every harness call of test/fnft_kdvv/*.c ---------------------------------------
def _kdvv_cases():
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "reference_fixtures.json")) as f:
        fx = json.load(f)
    return [pytest.param(b, id=b["file"].replace("fnft_kdvv_test_", "").replace(".c", ""))
            for b in fx["kdvv_error_bounds"]]


@pytest.mark.parametrize("b", _kdvv_cases())
def test_fnft_kdvv_analytic_bounds(oracle, fixtures, b):
    """fnft__kdvv_testcases.c:294-367 driven as test/fnft_kdvv/<file>.c drives it."""
    assert b["stages"]
    for st in b["stages"]:
        u, T, XI, M, exact = S.kdvv_case(fixtures, b["testcase"], st["D"])
        rc, cs = oracle.fnft_kdvv(u, T, M, XI, b["discretization"])
        assert rc == 0
        assert S.rel_err(cs, exact) <= st["bounds"][0], (st, b["file"])


# ---- discrete spectrum: bounds [3..5] of the same reference tests ---------------------------------
_DEG = {"2SPLIT2_MODAL": 1, "2SPLIT1A": 1, "2SPLIT1B": 1, "2SPLIT2A": 1, "2SPLIT2B": 1, "2SPLIT2S": 1,
        "2SPLIT3A": 3, "2SPLIT3B": 3, "2SPLIT3S": 2, "2SPLIT4A": 4, "2SPLIT4B": 2, "2SPLIT5A": 15,
        "2SPLIT5B": 15, "2SPLIT6A": 12, "2SPLIT6B": 6, "2SPLIT7A": 105, "2SPLIT7B": 105, "2SPLIT8A": 24,
        "2SPLIT8B": 12, "4SPLIT4A": 4, "4SPLIT4B": 2}


def _ds_cases():
    out = []
    for c in _bound_cases():
        b = c.values[0]
        if b["testcase"] != "SECH_FOCUSING":
            continue
        D = max(st["D"] for st in b["stages"])
        ups = 2 if b["discretization"].startswith("4SPLIT") else 1
        # degree of the polynomial whose roots the subsampled stage needs (CPU suite budget)
        if _DEG[b["discretization"]] * ups * np.sqrt(D) * np.log2(D) <= 4000:
            out.append(c)
    return out


@pytest.mark.parametrize("b", _ds_cases())
def test_fnft_nsev_discrete_spectrum_bounds(oracle, fixtures, b):
    """Default options of the harness (SUBSAMPLE_AND_REFINE, FULL filtering, 10 Newton steps,
    dstype BOTH, fnft__nsev_testcases.c:741-764) on the sech pulse with eigenvalues 0.7i, 1.7i, 2.7i:
    Hausdorff distance of the bound states, norming constants, residues against the file's bounds."""
    fx = fixtures["nsev_sech_focusing"]
    ex = [S.l2c(fx[k]) for k in ("bound_states", "normconsts", "residues")]
    # CPU budget: the first call of the file and its Richardson calls; the GPU suite replays all of them
    # ... unless the file sets options per call (user-supplied opts.Dsub / opts.niter,
    # fnft_nsev_test_adaptable_subsampling_factor.c:43-54): then every call is replayed
    per_call_opts = any(s["Dsub"] is not None or s["niter"] is not None for s in b["stages"])
    stages = b["stages"] if per_call_opts else b["stages"][:1] + [s for s in b["stages"][3:] if s["richardson"]]
    for st in stages:
        rc, bs, nc, res = oracle.fnft_nsev_ds(S.sech_focusing(st["D"]), fx["T"], b["discretization"],
                                              richardson=bool(st["richardson"]),
                                              Dsub=st["Dsub"] or 0, niter=10 if st["niter"] is None else st["niter"])
        assert rc == 0
        if not any(np.isfinite(x) for x in st["bounds_ds"]):
            continue   # fnft_nsev_test_nonregression_1.c: the call has to succeed, nothing else is checked
        assert bs.size == 3, (st, bs)
        errs = S.ds_errors(bs, nc, res, *ex)
        for e, bound in zip(errs, st["bounds_ds"]):
            if np.isfinite(bound):
                assert e <= bound, (st, errs, b["file"])


# ---- pieces under the discrete spectrum and the 4SPLIT4 front end, with the reference's own vectors ------
def _hausdorff(x, y):
    return max(max(min(abs(a - b) for b in y) for a in x), max(min(abs(a - b) for b in x) for a in y))


def test_misc_resample_golden(oracle, fixtures):
    """test/fnft__misc/fnft__misc_resample_test.c:28-66: shift of a chirped sech by four deltas, rel-L1 <= 3e-7."""
    f = fixtures["misc_resample"]
    D = f["D"]
    eps = f["span"] / (D - 1)
    t = f["T0"] + np.arange(D) * eps
    sig = lambda tt: f["amp"] / np.cosh(tt) * np.exp(1j * f["freq"] * tt)   # noqa: E731
    for delta in f["deltas"]:
        rc, qn = oracle.misc_resample(sig(t), eps, delta)
        assert rc == 0
        assert S.rel_err(qn, sig(t + delta)) <= f["tol_rel_l1"]


def test_poly_roots_golden(fixtures):
    """test/fnft__poly/fnft__poly_roots_fasteigen_test.c:27-44 (Hausdorff distance <= 100 eps) on the checker's
    root finder (numpy.roots below degree 2500, oracle/oracle.py: poly_roots)."""
    from oracle.oracle import poly_roots
    f = fixtures["poly_roots_fasteigen"]
    r = poly_roots(S.l2c(f["p"]))
    assert _hausdorff(r, S.l2c(f["roots_exact"])) <= f["tol_hausdorff"]


def test_scatter_bound_states_bo_golden(oracle, fixtures):
    """test/fnft__nse_scatter/fnft__nse_scatter_bound_states_test_bo.c:30-131.  a' agrees to round-off; a is a
    near-zero (|a| ~ 2e-5 at points that are eigenvalues of the continuous problem only), so it is compared
    with an absolute tolerance; b = phi/psi depends on the matching point at O(|a|) when a != 0, and the
    file's values come from a MATLAB rule (norm-balanced split) that differs from the C code's minimum-error
    rule (fnft__nse_scatter_bound_states.c:481-654) -- the reference's own test returns SUCCESS whatever the
    errors are (:133-140), so b is pinned only to 5e-3 here ("parity unpinned" beyond that)."""
    f = fixtures["nse_scatter_bound_states_bo"]
    D, T = f["D"], f["T"]
    eps = (T[1] - T[0]) / (D - 1)
    q = 3.0 / np.cosh(T[0] + np.arange(D) * eps) + 0j
    rc, a, ap, b = oracle.scatter_bound_states(q, T, S.l2c(f["bound_states"]), 1)
    assert rc == 0
    assert np.max(np.abs(a - S.l2c(f["a_vals"]))) < 1e-13
    assert S.rel_err(ap, S.l2c(f["aprime_vals"])) < 1e-12
    assert S.rel_err(b, S.l2c(f["b_vals"])) < 5e-3


@pytest.mark.parametrize("key", ["fmult_pow2", "fmult_nopow2"])
def test_poly_fmult_scalar_vectors(fixtures, key):
    """test/fnft__poly/fnft__poly_fmult_test_n_is_power_of_2.c:26-77 / ..._no_power_of_2.c:26-88: the MATLAB vectors
    against direct convolution of the input rule (the scalar tree of the GPU library is tested with them, -m gpu)."""
    f = fixtures[key]
    deg, n = f["deg"], f["n"]
    i = np.arange((deg + 1) * n, dtype=np.float64)
    p = np.sqrt(i + 1.0) * (np.cos(i) + 1j * np.sin(-2.0 * i))
    r = p[: deg + 1]
    for j in range(1, n):
        r = np.convolve(r, p[j * (deg + 1):(j + 1) * (deg + 1)])
    assert S.rel_err(r, S.l2c(f["result_exact"])) <= f["tol_rel_l1"]


def _finv_cases():
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "reference_fixtures.json")) as f:
        return [pytest.param(c, id=c["file"].replace("fnft__nse_finvscatter_test_", "").replace(".c", ""))
                for c in json.load(f)["nse_finvscatter"]["cases"]]


@pytest.mark.parametrize("c", _finv_cases())
def test_nse_finvscatter_round_trip(oracle, fixtures, c):
    """test/fnft__nse_finvscatter/fnft__nse_finvscatter_test.inc:28-75 as each of the four files runs it: the
    D = 16384 samples come back from their own transfer matrix within the file's bound (multiples of eps)."""
    from oracle.oracle import nse_finvscatter
    f = fixtures["nse_finvscatter"]
    D, eps_t = f["D"], f["eps_t"]
    i = np.arange(D)
    q_exact = ((i + 1) / (D + 1) / D) * np.exp(1j * i / D)
    rc, deg, tm, _ = oracle.nse_fscatter(q_exact, eps_t, c["kappa"], c["discretization"], normalize=False)
    assert rc == 0 and deg == D
    rc2, q = nse_finvscatter(tm, eps_t, c["kappa"], c["discretization"])
    assert rc2 == 0
    assert S.rel_err(q, q_exact) < c["bound_eps"] * 2.220446049250313e-16
