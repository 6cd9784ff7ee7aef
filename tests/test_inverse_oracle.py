"""The oracle's fnft_nsev_inverse (oracle/inverse.py) against the reference's own tests of test/fnft_nsev_inverse:
the sech data files (fixtures) and the analytic cases, each with its file's error bound (tests/inverse_cases.py)."""
import numpy as np
import pytest

import inverse_cases as IC
import signals as S
from oracle import inverse as INV
from oracle import load_oracle

TAGS = ("2split2A", "2split2_modal")


@pytest.fixture(scope="module")
def orc():
    return load_oracle()


def run(case, orc=None):
    scatter_a = None
    if orc is not None:
        def scatter_a(q, T, lam):
            rc, a, _ap, _b = orc.scatter_bound_states(q, T, lam, 1, skip_b=True)
            assert rc == 0
            return a
    cs = None if case.get("contspec") is None else np.array(case["contspec"], np.complex128)
    rc, q = INV.fnft_nsev_inverse(case["M"], cs, case.get("XI"), case.get("bound_states"), case.get("normconsts"),
                                  case["D"], case["T"], case["kappa"], case["opts"], q_seed=case.get("q_seed"),
                                  scatter_a=scatter_a)
    assert rc == 0
    err = S.rel_err(q, case["q_exact"])
    assert err < case["bound"], (err, case["bound"])
    return err


def test_XI_grid_is_the_fft_grid():
    D, T, M = 64, [-3.0, 5.0], 128
    XI = INV.nsev_inverse_XI(D, T, M)
    eps_t = (T[1] - T[0]) / (D - 1)
    xi = XI[0] + (XI[1] - XI[0]) / (M - 1) * np.arange(M)
    z = np.exp(2j * xi * eps_t)
    k = (np.arange(M) + M // 2 + 1) % M
    assert np.allclose(z, np.exp(2j * np.pi * k / M), atol=1e-12)


@pytest.mark.parametrize("n", (2048, 4096))
@pytest.mark.parametrize("tag", TAGS)
def test_sech_defocusing_data(tag, n):
    run(IC.sech_defocusing(tag, n))


@pytest.mark.parametrize("D", (512, 1024))
@pytest.mark.parametrize("tag", TAGS)
def test_truncated_soliton(tag, D):
    run(IC.truncated_soliton(tag, D, INV.nsev_inverse_XI))


@pytest.mark.parametrize("step", range(4))
@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("kind", ("B_of_tau", "b_of_xi"))
def test_B_of_tau_or_b_of_xi(kind, tag, step):
    run(IC.b_cases(kind, False, tag, step, INV.nsev_inverse_XI))


@pytest.mark.parametrize("step", range(4))
@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("kind", ("B_of_tau", "b_of_xi"))
def test_B_of_tau_or_b_of_xi_with_discrete_spectrum(kind, tag, step):
    run(IC.b_cases(kind, True, tag, step, INV.nsev_inverse_XI))


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("sign", ("focusing", "defocusing"))
def test_against_forward(orc, sign, tag):
    for idx in range(IC.n_against_forward(sign, tag)):
        case = IC.against_forward(sign, tag, idx)
        XI = INV.nsev_inverse_XI(case["D"], case["T"], case["M"])
        rc, cs = orc.fnft_nsev(case["q_exact"], case["T"], case["M"], XI, kappa=case["kappa"], disc=case["forward"],
                               cstype="RHO")
        assert rc == 0
        case.update(contspec=cs[:case["M"]], XI=XI)
        run(case)


@pytest.mark.parametrize("D", (512, 1024))
@pytest.mark.parametrize("dstype", ("NORMING_CONSTANTS", "RESIDUES"))
@pytest.mark.parametrize("tag", TAGS)
def test_against_forward_with_discrete_spectrum(orc, tag, dstype, D):
    case = IC.against_forward_w_discrete(tag, dstype, D)
    XI = INV.nsev_inverse_XI(D, case["T"], case["M"])
    rc, cs = orc.fnft_nsev(case["q_exact"], case["T"], case["M"], XI, kappa=1, disc="2SPLIT4B",
                           cstype="RHO")
    assert rc == 0
    rc, bs, nc, res = orc.fnft_nsev_ds(case["q_exact"], case["T"], disc="2SPLIT4B")
    assert rc == 0 and bs.size == 3
    case.update(contspec=cs[:case["M"]], XI=XI, bound_states=bs, normconsts=nc if dstype == "NORMING_CONSTANTS" else res)
    run(case, orc)


@pytest.mark.parametrize("D", (512, 1024))
@pytest.mark.parametrize("dstype", ("NORMING_CONSTANTS", "RESIDUES"))
def test_addsoliton_cdt(D, dstype):
    run(IC.addsoliton_cdt(D, dstype))


@pytest.mark.parametrize("dstype", ("NORMING_CONSTANTS", "RESIDUES"))
def test_multisoliton_cdt(dstype):
    run(IC.multisoliton_cdt(dstype))


def test_argument_checks_follow_the_reference():
    q8 = np.zeros(8, np.complex128)
    f = INV.fnft_nsev_inverse
    assert f(8, None, [0, 1], None, None, 8, [0, 1], 1)[0] == INV.EC_INVALID_ARGUMENT          # M > 0, no contspec
    assert f(9, np.zeros(9, complex), [0, 1], None, None, 8, [0, 1], 1)[0] == INV.EC_INVALID_ARGUMENT   # M odd
    assert f(4, np.zeros(4, complex), [0, 1], None, None, 8, [0, 1], 1)[0] == INV.EC_INVALID_ARGUMENT   # M < D
    assert f(12, np.zeros(12, complex), [0, 1], None, None, 12, [0, 1], 1)[0] == INV.EC_INVALID_ARGUMENT  # D not 2^k
    assert f(8, q8, [0, 1], None, None, 8, [1, 0], 1)[0] == INV.EC_INVALID_ARGUMENT            # T
    assert f(8, q8, [0, 1], None, None, 8, [0, 1], 0)[0] == INV.EC_INVALID_ARGUMENT            # kappa
    assert f(8, q8, [0, 1], [1j], [1.0], 8, [0, 1], -1)[0] == INV.EC_SANITY                    # solitons, defocusing
    assert f(8, q8, [0, 1], [-1j], [1.0], 8, [0, 1], 1)[0] == INV.EC_SANITY                    # lower half plane
    assert f(0, None, None, None, None, 8, [0, 1], 1)[0] == INV.EC_SANITY                      # nothing given
    assert f(8, q8, [0, 1], None, None, 8, [0, 1], 1, {"discretization": "2SPLIT4B"})[0] == INV.EC_INVALID_ARGUMENT


@pytest.mark.parametrize("name", ("focusing", "defocusing"))
def test_nse_scatter_matrix_known_answer(name):
    """test/fnft__nse_scatter/fnft__nse_scatter_matrix_test_{focusing,defocusing}_bo.c: D = 8, eps_t = 0.13, two lambdas,
    10 eps on the 16 values."""
    fx = IC.FIX["__nse_scatter_matrix__"][name]
    q = 0.4 * np.cos(np.arange(1, 9)) + 0.5j * np.sin(0.3 * np.arange(1, 9))
    res = INV.nse_scatter_matrix(q, 0.13, fx["kappa"], [2.0, 1.0 + 0.5j])
    exact = np.array([complex(a, b) for a, b in fx["result_exact"]])
    assert S.rel_err(res.ravel(), exact) < 10 * np.finfo(float).eps


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


def test_poly_roots_fftgridsearch_known_answers():
    _gridsearch_cases(INV.poly_roots_fftgridsearch, INV.poly_roots_fftgridsearch_paraherm, INV.hausdorff)
