"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/fnft_amd.h
declares, mirrors the reference's option defaults and argument validation (src/fnft_nsev.c:163-220)
-- all of which return before any GPU work -- and fails loudly (never falls back) without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from fnft_amd import build, capi as c
    build.build()
    c.load()
    c.silence_errors()
    return c


def test_exports_every_declared_symbol(capi):
    hdr = open(os.path.join(ROOT, "include", "fnft_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fnft_\w+|fnft__\w+)\s*\(", hdr))
    declared -= {"fnft_printf_ptr_t"}
    L = capi.load()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert set(capi.EXPORTED) <= declared


def test_default_opts_match_reference(capi):
    # src/fnft_nsev.c:26-36
    o = capi.default_opts()
    assert (o.bound_state_filtering, o.bound_state_localization, o.niter, o.Dsub) == (2, 2, 10, 0)
    assert (o.discspec_type, o.contspec_type, o.normalization_flag) == (0, 0, 1)
    assert o.discretization == capi.NSE_DISC["2SPLIT4B"]
    assert o.richardson_extrapolation_flag == 0
    import ctypes
    assert ctypes.sizeof(capi.NsevOpts) == 48


def test_max_K_and_numel(capi):
    L = capi.load()
    assert L.fnft_nsev_max_K(100, None) == 200                     # default 2SPLIT4B, degree 2
    o = capi.default_opts()
    o.discretization = capi.NSE_DISC["2SPLIT2_MODAL"]
    assert L.fnft_nsev_max_K(100, o) == 100
    o.discretization = capi.NSE_DISC["2SPLIT7A"]
    assert L.fnft_nsev_max_K(3, o) == 315
    # src/private/fnft__poly_fmult.c:40-43 : 4*(deg+1)*nextpow2(n)
    assert L.fnft__poly_fmult2x2_numel(1, 5) == 4 * 2 * 8
    assert L.fnft__nse_fscatter_numel(1000, capi.NSE_DISC["2SPLIT4B"]) == 4 * 3 * 1024
    assert L.fnft__nse_fscatter_numel(1000, capi.NSE_DISC["BO"]) == 0
    assert L.fnft__akns_fscatter_numel(8, capi.AKNS_DISC["2SPLIT4A"]) == 4 * 5 * 8


def test_validation_order_and_codes(capi):
    """Same checks in the same order as src/fnft_nsev.c:163-178; each returns
    FNFT_EC_INVALID_ARGUMENT (2) before anything else is looked at."""
    import ctypes as C
    L = capi.load()
    q = np.ones(8, np.complex128)
    T = np.array([0.0, 1.0])
    XI = np.array([-1.0, 1.0])
    cs = np.zeros(24, np.complex128)
    P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    call = lambda D, qq, TT, M, c, X, kappa, K=None, bs=None: L.fnft_nsev(  # noqa: E731
        D, qq, TT, M, c, X, K, bs, None, kappa, None)
    assert call(1, None, None, 4, P(cs), None, 5) == 2          # D first
    assert call(8, None, None, 4, P(cs), None, 5) == 2          # then q
    assert call(8, P(q), None, 4, P(cs), None, 5) == 2          # then T
    bad_T = np.array([1.0, 1.0])
    assert call(8, P(q), P(bad_T), 4, P(cs), None, 5) == 2
    assert call(8, P(q), P(T), 4, P(cs), None, 5) == 2          # XI (contspec given)
    bad_XI = np.array([2.0, 1.0])
    assert call(8, P(q), P(T), 4, P(cs), P(bad_XI), 5) == 2
    assert call(8, P(q), P(T), 4, P(cs), P(XI), 0) == 2         # kappa
    bs = np.zeros(16, np.complex128)
    assert call(8, P(q), P(T), 4, P(cs), P(XI), 1, None, P(bs)) == 2   # K_ptr with bound_states


def test_unknown_and_unsupported_options(capi):
    q = np.ones(8, np.complex128)
    o = capi.default_opts()
    o.discretization = 99
    assert capi.fnft_nsev(q, [0, 1], 4, [-1, 1], opts=o)[0] == 2                 # unknown discretization
    o = capi.default_opts()
    o.discretization = capi.NSE_DISC["BO"]                                        # slow scheme needs NEWTON
    assert capi.fnft_nsev(q, [0, 1], 4, [-1, 1], opts=o)[0] == 2
    o.bound_state_localization = 1
    assert capi.fnft_nsev(q, [0, 1], 4, [-1, 1], opts=o)[0] == capi.FNFT_EC_NOT_YET_IMPLEMENTED
    o = capi.default_opts()
    o.discretization = capi.NSE_DISC["CF4_2"]                                     # slow scheme, kappa = -1
    assert capi.fnft_nsev(q, [0, 1], 4, [-1, 1], kappa=-1, opts=o)[0] == capi.FNFT_EC_NOT_YET_IMPLEMENTED
    o = capi.default_opts()
    o.richardson_extrapolation_flag = 1                                           # accepted; needs the GPU
    assert capi.fnft_nsev(q, [0, 1], 4, [-1, 1], opts=o)[0] == capi.FNFT_EC_OTHER
    # discrete spectrum requested
    bs = np.zeros(16, np.complex128)
    assert capi.fnft_nsev(q, [0, 1], 4, [-1, 1], bound_states=bs, K=16)[0] == capi.FNFT_EC_OTHER  # needs the GPU
    o = capi.default_opts()
    o.bound_state_localization = 7
    assert capi.fnft_nsev(q, [0, 1], 4, [-1, 1], bound_states=bs, K=16, opts=o)[0] == 2
    o = capi.default_opts()
    o.contspec_type = 7
    assert capi.fnft_nsev(q, [0, 1], 4, [-1, 1], opts=o)[0] == -2                # wrapped twice like the reference


def test_private_seam_argument_checks(capi):
    import ctypes as C
    L = capi.load()
    q = np.ones(8, np.complex128)
    res = np.zeros(64, np.complex128)
    d = C.c_size_t(0)
    P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    # src/private/fnft__akns_fscatter.c:80-97
    assert L.fnft__akns_fscatter(0, P(q), P(q), 0.1, P(res), C.byref(d), None, 0) == 2
    assert L.fnft__akns_fscatter(8, None, P(q), 0.1, P(res), C.byref(d), None, 0) == 2
    assert L.fnft__akns_fscatter(8, P(q), P(q), 0.0, P(res), C.byref(d), None, 0) == 2
    assert L.fnft__akns_fscatter(8, P(q), P(q), 0.1, P(res), C.byref(d), None, 19) == 2   # BO: no degree
    # src/private/fnft__nse_fscatter.c:55-69
    assert L.fnft__nse_fscatter(8, P(q), 0.1, 3, P(res), C.byref(d), None, 0) == 2
    # src/private/fnft__poly_chirpz.c:44-49
    one = (C.c_double * 2)(1.0, 0.0)
    assert L.fnft_amd_poly_chirpz(3, None, one, one, 4, P(res)) == 2
    assert L.fnft_amd_poly_chirpz(3, P(q), one, one, 0, P(res)) == 2


def test_error_text_hook(capi):
    """src/fnft_errwarn.c:52-60 + src/private/fnft__errwarn.c:28-37: messages go through the
    user-settable printf; NULL disables them."""
    import ctypes as C
    L = capi.load()
    seen = []
    CB = C.CFUNCTYPE(C.c_int32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                     C.c_char_p)

    def hook(fmt, msg, func, line, a, b, c, suffix):
        seen.append((fmt, msg, func))
        return 0

    cb = CB(hook)
    L.fnft_errwarn_setprintf(C.cast(cb, C.c_void_p))
    assert L.fnft_errwarn_getprintf() == C.cast(cb, C.c_void_p).value
    assert L.fnft_nsev(1, None, None, 0, None, None, None, None, None, 1, None) == 2
    # the private seams raise their argument errors the same way (src/private/fnft__poly_roots_fasteigen.c:35-38,
    # fnft__nse_finvscatter.c:250-259, fnft__poly_specfact.c:32-39: E_INVALID_ARGUMENT(name))
    r4 = np.zeros(4, np.complex128)
    assert L.fnft__poly_roots_fasteigen(C.c_size_t(3), None, r4.ctypes.data_as(C.c_void_p)) == 2
    assert L.fnft__nse_finvscatter(C.c_size_t(8), r4.ctypes.data_as(C.c_void_p), r4.ctypes.data_as(C.c_void_p),
                                   C.c_double(0.0), 1, 0) == 2
    assert L.fnft__poly_specfact(C.c_size_t(4), r4.ctypes.data_as(C.c_void_p), r4.ctypes.data_as(C.c_void_p),
                                 C.c_size_t(0), 1) == 2
    L.fnft_errwarn_setprintf(None)
    assert L.fnft_errwarn_getprintf() is None
    assert seen and seen[0][0].startswith(b"FNFT Error: %s") and b"Invalid argument D" in seen[0][1]
    assert seen[0][2] == b"fnft_nsev"
    assert [(m, f) for _, m, f in seen[1:4]] == [(b"Invalid argument p.", b"fnft__poly_roots_fasteigen"),
                                                 (b"Invalid argument eps_t.", b"fnft__nse_finvscatter"),
                                                 (b"Invalid argument oversampling_factor.", b"fnft__poly_specfact")]


def test_no_gpu_means_loud_failure(capi):
    """Without a GPU the product path must fail (FNFT_EC_OTHER), never compute on the CPU."""
    L = capi.load()
    if L.fnft_amd_device_count() > 0:
        pytest.skip("a GPU is present")
    q = np.ones(8, np.complex128)
    rc, cs = capi.fnft_nsev(q, [0, 1], 4, [-1, 1])
    assert rc == capi.FNFT_EC_OTHER and not np.any(cs)
    assert capi.poly_fmult2x2(1, 2, np.ones((4, 4), np.complex128))[0] == capi.FNFT_EC_OTHER
    with pytest.raises(RuntimeError):
        capi.Plan(64, 16)


def test_kdvv_defaults_and_validation(capi):
    """src/fnft_kdvv.c:34-44 (default 2SPLIT8B) and :77-94 (checks and their order); the checks run
    before anything touches the GPU."""
    capi.silence_errors()
    assert capi.default_kdvv_opts().discretization == capi.KDV_DISC["2SPLIT8B"]
    u = np.ones(8, np.complex128)
    assert capi.fnft_kdvv(u[:1], [0, 1], 4, [-1, 1])[0] == 2                       # D
    assert capi.fnft_kdvv(u, [1, 0], 4, [-1, 1])[0] == 2                           # T
    assert capi.fnft_kdvv(u, [0, 1], 4, [-1, 1], want_contspec=False)[0] == 2      # contspec == NULL
    assert capi.fnft_kdvv(u, [0, 1], 4, [1, -1])[0] == 2                           # XI
    assert capi.fnft_kdvv(u, [0, 1], 4, [-1, 1], K=3)[0] == capi.FNFT_EC_NOT_YET_IMPLEMENTED
    bs = np.zeros(4, np.complex128)
    assert capi.fnft_kdvv(u, [0, 1], 4, [-1, 1], bound_states=bs)[0] == capi.FNFT_EC_NOT_YET_IMPLEMENTED
    assert capi.fnft_kdvv(u, [0, 1], 4, [-1, 1], normconsts=bs)[0] == capi.FNFT_EC_NOT_YET_IMPLEMENTED
    assert capi.fnft_kdvv(u, [0, 1], 4, [-1, 1], discretization="CF4_2")[0] == capi.FNFT_EC_NOT_YET_IMPLEMENTED
    assert capi.fnft_kdvv(u, [0, 1], 4, [-1, 1], discretization=99)[0] == -2
    assert capi.fnft_kdvv(u, [0, 1], 4, [-1, 1])[0] == capi.FNFT_EC_OTHER          # no GPU here


def test_inverse_validation_failure_leaves_blaschke_factors_in_contspec(capi):
    """src/fnft_nsev_inverse.c:198-199, 1013-1033: with REFLECTION_COEFFICIENT and bound states the reference has multiplied
    the Blaschke factors into the CALLER'S contspec before transfer_matrix_from_reflection_coefficient's own checks
    (:393-400: M != D, kappa != -1 for AB_FROM_ITER) fail; the drop-in leaves the same array behind (no GPU involved)."""
    capi.silence_errors()
    D, M = 16, 32                                                       # M must be even and >= D (:135-140)
    XI = np.array([-2.0, 3.0])
    xi = XI[0] + (XI[1] - XI[0]) / (M - 1) * np.arange(M)
    cs0 = np.exp(-xi ** 2) * (1.0 + 0.5j)
    bs = np.array([0.3 + 1.1j, -0.2 + 0.4j])
    want = cs0.copy()
    for lam in bs:
        want = want * ((xi - lam) / (xi - np.conj(lam)))
    opts = {"contspec_type": "REFLECTION_COEFFICIENT", "contspec_inversion_method": "TFMATRIX_CONTAINS_AB_FROM_ITER"}
    cs = cs0.copy()
    rc, _ = capi.fnft_nsev_inverse(M, cs, XI, bs, np.ones(2, np.complex128), D, [-1.0, 1.0], +1, opts)   # K > 0 needs kappa = +1
    assert rc == -capi.FNFT_EC_INVALID_ARGUMENT                        # M != D, wrapped as a subroutine failure
    assert np.allclose(cs, want, rtol=1e-15, atol=0)
    cs = cs0[:D].copy()                                                 # M == D, kappa = +1
    xiD = XI[0] + (XI[1] - XI[0]) / (D - 1) * np.arange(D)
    wantD = cs.copy()
    for lam in bs:
        wantD = wantD * ((xiD - lam) / (xiD - np.conj(lam)))
    rc, _ = capi.fnft_nsev_inverse(D, cs, XI, bs, np.ones(2, np.complex128), D, [-1.0, 1.0], +1, opts)
    assert rc == -capi.FNFT_EC_INVALID_ARGUMENT and np.allclose(cs, wantD, rtol=1e-15, atol=0)
    cs = cs0.copy()                                                     # no bound states: untouched
    rc, _ = capi.fnft_nsev_inverse(M, cs, XI, None, None, D, [-1.0, 1.0], -1, opts)
    assert rc == -capi.FNFT_EC_INVALID_ARGUMENT and np.array_equal(cs, cs0)


def test_bench_launch_breakdown_assigns_every_level_once():
    """bench.py's per-stage roofline: the stages of a 2^20 MODAL transform cover levels 0..19 exactly
    once and their algorithmic bytes add up to SURVEY 8(d)'s closed form."""
    import bench

    class FakePlan:
        seq = (["KLeafMulti<1, 3>", "KMulti<128, 3>", "KMulti<1024, 3>"]
               + sum((["KMid<2>", "KColBridge2<%d>" % (4 << i)] for i in range(7)), [])
               + ["KMid<2>", "KColInv<512>", "KFinalizeScales", "KChirpColFwd<512, false>", "KChirpRows",
                  "KChirpColInv<512, false, false>"])

        def set_launch_timing(self, on):
            pass

        def launch_times(self):
            return [(n, 0.01) for n in self.seq]

    D = 1 << 20
    st = bench.launch_breakdown(FakePlan(), lambda: None, 2, 1, D, 1, sync=lambda: None)
    tree, chirp = st[:-1], st[-1]
    assert [s["levels"] for s in tree] == [[0, 5], [6, 8], [9, 11], [12, 19]]
    assert sum(s["algorithmic_bytes"] for s in tree) == bench.bytes_tree(D, 1) == 2885680960
    assert tree[-1]["launches"]["KMid<2>"] == 8 and "KChirpRows" not in tree[-1]["launches"]
    # the chirp z-transform and the epilogue are a stage of their own (not part of the tree's byte figure)
    assert chirp["stage"] == "chirp-z + epilogue" and chirp["levels"] is None and set(chirp["launches"]) == set(FakePlan.seq[-3:])
    assert all("model_frac" in s and "frac" not in s for s in st)
