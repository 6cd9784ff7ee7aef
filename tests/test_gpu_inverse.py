"""fnft_nsev_inverse through the C ABI on the GPU: the reference's own tests of test/fnft_nsev_inverse (same cases,
same error bounds as tests/test_inverse_oracle.py runs against the oracle), agreement with the oracle on the same
inputs, and the fnft__poly_specfact seam."""
import numpy as np
import pytest

import inverse_cases as IC
import signals as S

pytestmark = pytest.mark.gpu
TAGS = ("2split2A", "2split2_modal")


@pytest.fixture(scope="module")
def capi():
    from fnft_amd import capi as c
    c.load()
    c.silence_errors()
    return c


@pytest.fixture(scope="module")
def INV():
    from oracle import inverse
    return inverse


@pytest.fixture(scope="module")
def orc():
    from oracle import load_oracle
    return load_oracle()


def xi_of(capi):
    def f(D, T, M):
        rc, XI = capi.nsev_inverse_XI(D, T, M)
        assert rc == 0
        return XI
    return f


def run(capi, case, INV=None, tol_vs_oracle=None, orc=None):
    cs = None if case.get("contspec") is None else np.array(case["contspec"], np.complex128)
    rc, q = capi.fnft_nsev_inverse(case["M"], cs, case.get("XI"), case.get("bound_states"), case.get("normconsts"),
                                   case["D"], case["T"], case["kappa"], case["opts"], q_seed=case.get("q_seed"))
    assert rc == 0, capi.last_error()
    err = S.rel_err(q, case["q_exact"])
    assert err < case["bound"], (err, case["bound"])
    if INV is not None:
        scatter_a = None
        if orc is not None:
            def scatter_a(qq, T, lam):
                rc2, a, _ap, _b = orc.scatter_bound_states(qq, T, lam, 1, skip_b=True)
                assert rc2 == 0
                return a
        cs2 = None if case.get("contspec") is None else np.array(case["contspec"], np.complex128)
        rc, qo = INV.fnft_nsev_inverse(case["M"], cs2, case.get("XI"), case.get("bound_states"), case.get("normconsts"),
                                       case["D"], case["T"], case["kappa"], case["opts"], q_seed=case.get("q_seed"),
                                       scatter_a=scatter_a)
        assert rc == 0
        d = S.rel_err(q, qo)
        assert d < tol_vs_oracle, d
        if cs is not None:   # the reference modifies the caller's contspec; so do the oracle and the library
            assert S.rel_err(cs, cs2) < 1e-12
    return err


def test_XI(capi, INV):
    for D, T, M in ((8, [0.0, 7.0], 10), (512, [-2.0, 2.0], 2048), (4096, [-25.0, 25.0], 4096)):
        rc, XI = capi.nsev_inverse_XI(D, T, M)
        assert rc == 0
        assert np.allclose(XI, INV.nsev_inverse_XI(D, T, M), rtol=1e-15, atol=0)


@pytest.mark.parametrize("n", (2048, 4096))
@pytest.mark.parametrize("tag", TAGS)
def test_sech_defocusing_data(capi, INV, tag, n):
    run(capi, IC.sech_defocusing(tag, n), INV, 1e-9)


@pytest.mark.parametrize("D", (512, 1024))
@pytest.mark.parametrize("tag", TAGS)
def test_truncated_soliton(capi, INV, tag, D):
    run(capi, IC.truncated_soliton(tag, D, xi_of(capi)), INV, 1e-9)


@pytest.mark.parametrize("step", range(4))
@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("kind", ("B_of_tau", "b_of_xi"))
def test_B_of_tau_or_b_of_xi(capi, INV, kind, tag, step):
    run(capi, IC.b_cases(kind, False, tag, step, xi_of(capi)), INV, 1e-9)


@pytest.mark.parametrize("step", range(4))
@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("kind", ("B_of_tau", "b_of_xi"))
def test_B_of_tau_or_b_of_xi_with_discrete_spectrum(capi, INV, kind, tag, step):
    run(capi, IC.b_cases(kind, True, tag, step, xi_of(capi)), INV, 1e-8)


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("sign", ("focusing", "defocusing"))
def test_against_forward(capi, INV, sign, tag):
    for idx in range(IC.n_against_forward(sign, tag)):
        case = IC.against_forward(sign, tag, idx)
        XI = xi_of(capi)(case["D"], case["T"], case["M"])
        rc, cs = capi.fnft_nsev(case["q_exact"], case["T"], case["M"], XI, kappa=case["kappa"],
                                discretization=case["forward"], contspec_type="REFLECTION_COEFFICIENT")
        assert rc == 0
        case.update(contspec=cs[:case["M"]], XI=XI)
        run(capi, case, INV, 1e-11)


@pytest.mark.parametrize("D", (512, 1024))
@pytest.mark.parametrize("dstype", ("NORMING_CONSTANTS", "RESIDUES"))
@pytest.mark.parametrize("tag", TAGS)
def test_against_forward_with_discrete_spectrum(capi, INV, orc, tag, dstype, D):
    case = IC.against_forward_w_discrete(tag, dstype, D)
    XI = xi_of(capi)(D, case["T"], case["M"])
    out = capi.fnft_nsev_ds(case["q_exact"], case["T"], discretization="2SPLIT4B", M=case["M"], XI=XI, K=10)
    rc, bs, nc, res, cs = out
    assert rc == 0 and bs.size == 3
    case.update(contspec=cs[:case["M"]], XI=XI, bound_states=bs, normconsts=nc if dstype == "NORMING_CONSTANTS" else res)
    run(capi, case, INV, 1e-7, orc)


@pytest.mark.parametrize("D", (512, 1024))
@pytest.mark.parametrize("dstype", ("NORMING_CONSTANTS", "RESIDUES"))
def test_addsoliton_cdt(capi, INV, D, dstype):
    run(capi, IC.addsoliton_cdt(D, dstype), INV, 1e-9)


@pytest.mark.parametrize("dstype", ("NORMING_CONSTANTS", "RESIDUES"))
def test_multisoliton_cdt(capi, INV, dstype):
    run(capi, IC.multisoliton_cdt(dstype), INV, 1e-12)


@pytest.mark.parametrize("log2D", (18, 20))
def test_full_size_round_trip(capi, log2D):
    """D = 2^18, 2^20: contspec of a sech pulse by the forward transform, back by the inverse (REFL_COEFF method,
    M = 2D)."""
    D = 1 << log2D
    T = [-32.0, 32.0]
    t = S.tgrid(T, D)
    q0 = 0.4 / np.cosh(t) * np.exp(-1j * t)
    rc, XI = capi.nsev_inverse_XI(D, T, 2 * D, "2SPLIT2_MODAL")
    assert rc == 0
    rc, cs = capi.fnft_nsev(q0, T, 2 * D, XI, kappa=1, discretization="2SPLIT2_MODAL",
                            contspec_type="REFLECTION_COEFFICIENT")
    assert rc == 0
    rc, q = capi.fnft_nsev_inverse(2 * D, cs[:2 * D].copy(), XI, None, None, D, T, 1,
                                   {"discretization": "2SPLIT2_MODAL"})
    assert rc == 0, capi.last_error()
    assert S.rel_err(q, q0) < 1e-5   # the A(z) = 1 construction is approximate: not a round-off bound


@pytest.mark.parametrize("kappa", (-1, 0, 1))
@pytest.mark.parametrize("deg", (7, 255, 1000))
def test_poly_specfact_vs_oracle(capi, INV, kappa, deg):
    rng = np.random.default_rng(deg + kappa)
    p = (rng.standard_normal(deg + 1) + 1j * rng.standard_normal(deg + 1)) * (0.3 / np.sqrt(deg + 1))
    if kappa == 0:
        p[0] += 2.0
    rc, r = capi.poly_specfact(p, 8, kappa)
    assert rc == 0, capi.last_error()
    ro, _w = INV.poly_specfact(p, 8, kappa)
    assert S.rel_err(r, ro) < 1e-11


def test_argument_checks_follow_the_reference(capi):
    q8 = np.zeros(8, np.complex128)
    f = capi.fnft_nsev_inverse
    assert f(8, None, [0, 1], None, None, 8, [0, 1], 1)[0] == 2
    assert f(9, np.zeros(9, complex), [0, 1], None, None, 8, [0, 1], 1)[0] == 2
    assert f(4, np.zeros(4, complex), [0, 1], None, None, 8, [0, 1], 1)[0] == 2
    assert f(12, np.zeros(12, complex), [0, 1], None, None, 12, [0, 1], 1)[0] == 2
    assert f(8, q8.copy(), [0, 1], None, None, 8, [1, 0], 1)[0] == 2
    assert f(8, q8.copy(), [0, 1], None, None, 8, [0, 1], 0)[0] == 2
    assert f(8, q8.copy(), [0, 1], [1j], [1.0], 8, [0, 1], -1)[0] == 7
    assert f(8, q8.copy(), [0, 1], [-1j], [1.0], 8, [0, 1], 1)[0] == 7
    assert f(0, None, None, None, None, 8, [0, 1], 1)[0] == 7
    assert f(8, q8.copy(), [0, 1], None, None, 8, [0, 1], 1, {"discretization": "2SPLIT4B"})[0] == 2
    assert f(8, q8.copy(), [0, 1], [1j, 1j], [1.0, 1.0], 8, [0, 1], 1)[0] == -7     # multiplicity, :756-761


def test_smallest_sizes_round_trip(capi, INV):
    """D = 2 and D = 4 (one leaf block far below its capacity), M = 8: forward then inverse, both inversion methods."""
    for D in (2, 4):
        T = [0.0, D - 1.0]
        q0 = np.array([0.1 + 0.05j, -0.07j, 0.02, 0.06][:D], np.complex128)
        for kappa, method, M in ((1, "TFMATRIX_CONTAINS_REFL_COEFF", 8), (-1, "TFMATRIX_CONTAINS_AB_FROM_ITER", D)):
            rc, XI = capi.nsev_inverse_XI(D, T, M, "2SPLIT2_MODAL")
            assert rc == 0
            rc, cs = capi.fnft_nsev(q0, T, M, XI, kappa=kappa, discretization="2SPLIT2_MODAL",
                                    contspec_type="REFLECTION_COEFFICIENT")
            assert rc == 0
            opts = {"discretization": "2SPLIT2_MODAL", "contspec_inversion_method": method}
            c1, c2 = cs[:M].copy(), cs[:M].copy()
            rc, q = capi.fnft_nsev_inverse(M, c1, XI, None, None, D, T, kappa, opts)
            assert rc == 0, capi.last_error()
            rco, qo = INV.fnft_nsev_inverse(M, c2, XI, None, None, D, T, kappa, opts)
            assert rco == 0
            assert S.rel_err(q, qo) < 1e-12
            if method.endswith("ITER"):
                assert S.rel_err(q, q0) < 1e-13      # M = D: the iteration reproduces the samples


def test_inner_checks_return_subroutine_codes(capi):
    """The checks of the reference's static helpers come back through CHECK_RETCODE as -(code)."""
    D = 8
    T = [0.0, 7.0]
    cs = np.full(16, 0.01 + 0.0j)
    XI = capi.nsev_inverse_XI(D, T, 16)[1]
    f = capi.fnft_nsev_inverse
    # iteration method needs M = D (src/fnft_nsev_inverse.c:393-396) and kappa = -1 (:399-400)
    assert f(16, cs.copy(), XI, None, None, D, T, -1, {"contspec_inversion_method": "TFMATRIX_CONTAINS_AB_FROM_ITER"})[0] == -2
    assert f(8, cs[:8].copy(), XI, None, None, D, T, 1, {"contspec_inversion_method": "TFMATRIX_CONTAINS_AB_FROM_ITER"})[0] == -2
    # B(tau) needs M = D and a symmetric time window (:643-647)
    assert f(16, cs.copy(), None, None, None, D, T, 1, {"contspec_type": "B_OF_TAU"})[0] == -2
    assert f(8, cs[:8].copy(), None, None, None, D, T, 1, {"contspec_type": "B_OF_TAU"})[0] == -2
    # a seed potential together with a continuous spectrum is not a combination (:890-891)
    assert f(8, cs[:8].copy(), XI, [1j], [1.0], D, T, 1, {"contspec_inversion_method": "USE_SEED_POTENTIAL_INSTEAD"})[0] == -2


@pytest.mark.parametrize("name", ("focusing", "defocusing"))
def test_nse_scatter_matrix_known_answer(capi, name):
    """The reference's known answers of fnft__nse_scatter_matrix (BO, D = 8) through the C ABI, at the file's 10 eps."""
    fx = IC.FIX["__nse_scatter_matrix__"][name]
    q = 0.4 * np.cos(np.arange(1, 9)) + 0.5j * np.sin(0.3 * np.arange(1, 9))
    rc, res = capi.nse_scatter_matrix(q, 0.13, fx["kappa"], [2.0, 1.0 + 0.5j])
    assert rc == 0, capi.last_error()
    exact = np.array([complex(a, b) for a, b in fx["result_exact"]])
    assert S.rel_err(res.ravel(), exact) < 10 * np.finfo(float).eps


@pytest.mark.parametrize("kappa", (1, -1))
def test_nse_scatter_matrix_vs_oracle(capi, INV, kappa):
    """Many chunks (D = 5000) and many lambdas; with and without the derivative; explicit r."""
    rng = np.random.default_rng(5 + kappa)
    D = 5000
    t = S.tgrid([-10.0, 10.0], D)
    q = (0.8 / np.cosh(t)) * np.exp(0.3j * t) + 0.01 * (rng.standard_normal(D) + 1j * rng.standard_normal(D))
    lam = np.concatenate([np.linspace(-2, 2, 7), np.array([0.3 + 0.4j, -0.2 + 0.9j, 0.1j])])
    eps = 20.0 / (D - 1)
    ref = INV.nse_scatter_matrix(q, eps, kappa, lam)
    rc, res = capi.nse_scatter_matrix(q, eps, kappa, lam)
    assert rc == 0, capi.last_error()
    assert S.rel_err(res, ref) < 1e-11
    rc, res4 = capi.nse_scatter_matrix(q, eps, kappa, lam, derivative=False)
    assert rc == 0 and np.array_equal(res4, res[:, :4])
    rc, resr = capi.nse_scatter_matrix(q, eps, kappa, lam, r=-kappa * np.conj(q))
    assert rc == 0 and S.rel_err(resr, res) < 1e-14
    assert capi.nse_scatter_matrix(q, eps, kappa, lam, discretization="CF4_3")[0] == 6    # not yet implemented
    assert capi.nse_scatter_matrix(q, -1.0, kappa, lam)[0] == 2


def _gridsearch_cases(search, search_ph, hausdorff):
    """The three files of test/fnft__poly/fnft__poly_roots_fftgridsearch_test_*.c with their own bounds."""
    for M, eb1, eb2 in IC.GRID_EVEN:
        r = search(IC.GRID_P_EVEN, M, [0.0, 2 * np.pi])
        assert hausdorff(r, IC.GRID_ROOTS_EVEN) <= eb1
        r = search(IC.GRID_P_EVEN, M, [1.0, 1.5])
        assert hausdorff(r, IC.GRID_ROOTS_EVEN[:1]) <= eb2
    for M, eb in IC.GRID_ODD:
        r = search(IC.GRID_P_ODD, M, [0.0, 2 * np.pi])
        assert hausdorff(r, IC.GRID_ROOTS_ODD) <= eb
    for M, eb1, eb2 in IC.GRID_PH:
        r = search_ph(IC.GRID_P_EVEN, M, [0.0, 2 * np.pi])
        assert hausdorff(r, IC.GRID_ROOTS_EVEN) <= eb1
        r = search_ph(IC.GRID_P_EVEN, M, [1.0, 1.5])
        assert hausdorff(r, IC.GRID_ROOTS_EVEN[:1]) <= eb2


def test_poly_roots_fftgridsearch_known_answers(capi, INV):
    def search(p, M, PHI):
        rc, r = capi.poly_roots_fftgridsearch(p, M, PHI)
        assert rc == 0, capi.last_error()
        return r

    def search_ph(p, M, PHI):
        rc, r = capi.poly_roots_fftgridsearch(p, M, PHI, paraherm=True)
        assert rc == 0, capi.last_error()
        return r
    _gridsearch_cases(search, search_ph, INV.hausdorff)


def test_poly_roots_fftgridsearch_vs_oracle(capi, INV):
    """A degree-44 polynomial with 20 roots on the unit circle, grid of 32 deg points (fnft_nsep's oversampling): the
    same estimates in the same order as the oracle, and every root on the circle among them."""
    rng = np.random.default_rng(11)
    on = np.exp(1j * (0.15 + 0.3 * np.arange(20)))
    off = rng.uniform(0.4, 0.7, 12) * np.exp(1j * rng.uniform(0, 2 * np.pi, 12))
    roots = np.concatenate([on, off, 1.0 / np.conj(off)])       # para-Hermitian up to a phase
    p = np.poly(roots)
    deg = p.size - 1
    M = 32 * deg
    ref = INV.poly_roots_fftgridsearch(p, M, [0.0, 2 * np.pi])
    rc, got = capi.poly_roots_fftgridsearch(p, M, [0.0, 2 * np.pi])
    assert rc == 0 and got.size == ref.size and ref.size >= 20
    assert np.max(np.abs(got - ref)) < 1e-9
    assert INV.hausdorff(got, on) < 1e-3
    assert capi.poly_roots_fftgridsearch(p[:2], 64, [0.0, 1.0])[0] == 2                      # deg < 2
    assert capi.poly_roots_fftgridsearch(p, 64, [1.0, 0.5])[0] == 2                          # PHI
    assert capi.poly_roots_fftgridsearch(np.poly(roots[:3]), 64, [0.0, 1.0], paraherm=True)[0] == 2   # odd degree


def test_inverse_b_of_xi_at_2p20(capi):
    """fnft_nsev_inverse from b(xi) at D = M = 2^20 (VERDICT r2 / ADVICE: the spectral factorization at oversampling 8 is
    an any-length DFT of 8 398 080 points, i.e. a chirp transform of 2^25 points -- beyond round 2's 2^24 ceiling):
    the sech pulse below the soliton threshold, second-order convergence to the exact signal."""
    import signals as S
    D = 1 << 20
    T = [-25.0, 25.0]
    A, t0 = 0.45, 1.2
    rc, XI = capi.nsev_inverse_XI(D, T, D, "2SPLIT2_MODAL")
    assert rc == 0
    xi = XI[0] + (XI[1] - XI[0]) / (D - 1) * np.arange(D)
    with np.errstate(over="ignore"):
        cs = 1j * np.exp(-2j * xi * t0) * np.sin(np.pi * A) / np.cosh(np.pi * xi)
    rc, q = capi.fnft_nsev_inverse(D, cs, XI, None, None, D, T, 1, {"discretization": "2SPLIT2_MODAL", "contspec_type": "B_OF_XI"})
    assert rc == 0, capi.last_error()
    exact = 1j * A / np.cosh(S.tgrid(T, D) - t0)
    assert S.rel_err(q, exact) < 5e-9   # 1.4e-9 at 2^18 (second order: ~1e-10 here, above the layer peeling's round-off)
