"""Index-arithmetic checks of the HIP kernel bodies WITHOUT a GPU.

tests/emu/ compiles the very same kernel bodies (fnft_amd/csrc/nft_kernels.h, fft_dev.h,
nft_plan.h) with g++ into a thread-per-lane emulator (OS thread = lane, std::barrier =
__syncthreads, heap block = LDS).  This is test infrastructure: it proves the workgroup FFT
tilings, the body/tail layout, the alias fix, the split-transform path and the chirp z-transform
before GPU minutes are spent, and it keeps doing so on machines without a GPU.  The product
library never contains or loads it.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import signals as S
from oracle.oracle import AKNS_DISC, NSE_DISC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libfnft_emu.so")
CSRC = os.path.join(ROOT, "fnft_amd", "csrc")
vp = C.c_void_p


def _P(a):
    return a.ctypes.data_as(vp)


@pytest.fixture(scope="module")
def emu():
    deps = [os.path.join(EMU_DIR, f) for f in ("emu_main.cpp", "emu_backend.h")]
    deps += [os.path.join(CSRC, f) for f in ("dev_compat.h", "fft_dev.h", "nft_kernels.h", "nft_real.h", "nft_dispatch.h",
                                             "nft_plan.h", "nft_api.h", "nft_discspec.h")]
    if not os.path.exists(EMU_LIB) or max(map(os.path.getmtime, deps)) > os.path.getmtime(EMU_LIB):
        subprocess.check_call(["g++", "-std=c++20", "-O2", "-fPIC", "-shared", "-pthread",
                               "-Wno-unknown-pragmas", "-o", EMU_LIB,
                               os.path.join(EMU_DIR, "emu_main.cpp")])
    L = C.CDLL(EMU_LIB)
    L.emu_fft_pair_cfg.argtypes = [C.c_int, C.c_int, vp, vp]
    L.emu_fft_col_cfg.argtypes = [C.c_int, C.c_int, vp, vp]
    L.emu_poly_fmult2x2.argtypes = [C.POINTER(C.c_size_t), C.c_size_t, vp, vp, C.POINTER(C.c_int32)]
    L.emu_akns_fscatter.argtypes = [C.c_size_t, vp, vp, C.c_double, C.c_int, vp, C.POINTER(C.c_size_t),
                                    C.POINTER(C.c_int32), C.c_int]
    L.emu_poly_chirpz.argtypes = [C.c_size_t, vp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.c_size_t, vp]
    L.emu_nsev_contspec.argtypes = [C.c_size_t, vp, vp, C.c_size_t, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.emu_kdvv_contspec.argtypes = [C.c_size_t, vp, vp, C.c_size_t, vp, vp, C.c_int, C.c_int]
    L.emu_nsev_discspec.argtypes = [C.c_size_t, vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                    C.c_int, C.POINTER(C.c_size_t), vp, vp]
    return L


@pytest.mark.parametrize("kind,N", [("pair", n) for n in (8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096)]
                         + [("col", n) for n in (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048)])
def test_workgroup_fft_tilings(emu, kind, N):
    """fft_wg in every (N, R, B, buffering) tiling the kernels instantiate, against numpy."""
    fn = emu.emu_fft_pair_cfg if kind == "pair" else emu.emu_fft_col_cfg
    B = fn(N, -1, None, None)
    assert B > 0
    rng = np.random.default_rng(N)
    x = (rng.standard_normal((B, N)) + 1j * rng.standard_normal((B, N))).astype(np.complex128)
    y = np.zeros_like(x)
    fn(N, -1, _P(x), _P(y))
    assert np.max(np.abs(y - np.fft.fft(x, axis=1))) < 1e-13 * N
    z = np.zeros_like(x)
    fn(N, +1, _P(x), _P(z))   # forward then inverse through the same LDS buffers
    assert np.max(np.abs(z / N - x)) < 1e-13


def _fmult(emu, deg, n, p, norm=True):
    numel = 4 * (deg + 1) * (1 << max(0, int(np.ceil(np.log2(max(n, 1))))))
    buf = np.zeros(numel, np.complex128)
    buf[: p.size] = p.ravel()
    res = np.zeros(numel, np.complex128)
    d = C.c_size_t(deg)
    W = C.c_int32(0)
    rc = emu.emu_poly_fmult2x2(C.byref(d), n, _P(buf), _P(res), C.byref(W) if norm else None)
    assert rc == 0
    return d.value, res[: 4 * (d.value + 1)].reshape(4, -1).copy(), W.value


@pytest.mark.parametrize("key", ["fmult2x2_pow2", "fmult2x2_nopow2"])
@pytest.mark.parametrize("norm", [False, True])
def test_tree_golden(emu, fixtures, key, norm):
    fx = fixtures[key]
    d, res, W = _fmult(emu, fx["deg"], fx["n"], S.fmult_test_input(fx["deg"], fx["n"]), norm)
    if norm:
        assert W != 0
        res = res * 2.0 ** W
    assert S.rel_err(res.ravel(), S.l2c(fx["result_exact"])) <= fx["tol_rel_l1"]


@pytest.mark.parametrize("deg,n", [(1, 1), (1, 2), (1, 7), (2, 13), (3, 6), (4, 5), (5, 9), (7, 33), (1, 300),
                                   (2, 100)])
def test_tree_vs_direct(emu, deg, n):
    rng = np.random.default_rng(10 * deg + n)
    p = 0.35 * (rng.standard_normal((4, n * (deg + 1))) + 1j * rng.standard_normal((4, n * (deg + 1))))
    d, res, W = _fmult(emu, deg, n, p)
    ref = S.tree_direct(p, deg, n)
    assert d == deg * n
    # random factors give products with a huge coefficient range; FFT products carry absolute
    # accuracy, so the bound grows with n (the oracle shows the same growth)
    assert S.rel_err((res * 2.0 ** W).ravel(), ref.ravel()) < (1e-13 if n < 100 else 5e-12)


@pytest.mark.parametrize("scheme", [s for s in sorted(AKNS_DISC) if s.startswith("2SPLIT")])
def test_akns_fscatter_golden(emu, oracle, fixtures, scheme):
    fx = fixtures["akns_fscatter"]["schemes"][scheme]
    q, r, z = S.akns_test_signal(fx["D"])
    res = np.zeros(int(oracle.lib.orc_akns_fscatter_numel(q.size, AKNS_DISC[scheme])), np.complex128)
    d = C.c_size_t(0)
    W = C.c_int32(0)
    rc = emu.emu_akns_fscatter(q.size, _P(q), _P(r), fx["eps_t"], 1, _P(res), C.byref(d), C.byref(W),
                               AKNS_DISC[scheme])
    assert rc == 0
    tm = res[: 4 * (d.value + 1)].reshape(4, -1) * 2.0 ** W.value
    vals = np.concatenate([oracle.poly_eval(tm[e], z) for e in range(4)])
    assert S.rel_err(vals, S.l2c(fx["result_exact"])) <= fx["tol_rel_l1"]  # err_bnd of the reference test of this scheme


def test_chirpz_golden(emu, fixtures):
    fx = fixtures["chirpz"]
    p = S.l2c(fx["p"])
    A, W = complex(*fx["A"]), np.exp(1j * fx["W_arg"])
    for M, key in ((3, "result_M3"), (6, "result_M6")):
        out = np.zeros(M, np.complex128)
        rc = emu.emu_poly_chirpz(p.size - 1, _P(p), (C.c_double * 2)(A.real, A.imag),
                                 (C.c_double * 2)(W.real, W.imag), M, _P(out))
        assert rc == 0
        assert S.rel_err(out, S.l2c(fx[key])) <= fx["tol_rel_l1"]


@pytest.mark.parametrize("D,M,disc,kappa", [(256, 16, "2SPLIT2_MODAL", 1), (1000, 37, "2SPLIT4B", 1),
                                             (300, 50, "2SPLIT3A", -1), (4097, 64, "2SPLIT4B", 1),
                                             (16384, 48, "2SPLIT2_MODAL", 1), (8192, 40, "2SPLIT3A", 1),
                                             (300, 20, "2SPLIT5B", 1), (256, 16, "2SPLIT7A", 1),
                                             (257, 16, "2SPLIT8A", -1), (128, 16, "2SPLIT6B", 1),
                                             (100, 16, "2SPLIT6A", 1), (200, 16, "2SPLIT7B", -1),
                                             (200, 16, "2SPLIT8B", 1), (96, 16, "2SPLIT5A", -1),
                                             (512, 16, "4SPLIT4A", 1), (511, 16, "4SPLIT4B", 1),
                                             (300, 16, "4SPLIT4A", -1)])
def test_nsev_vs_oracle(emu, oracle, D, M, disc, kappa):
    """Whole pipeline in the emulator; D = 4097 with degree 2 reaches a split transform in the
    top level of the tree; D = 16384 (degree 1) and D = 8192 (degree 3, lengths that are not 2d)
    have consecutive split levels joined by the bridge kernel."""
    T, XI = np.array([-25.0, 25.0]), np.array([-1.4, 1.6])
    q = S.sech_focusing(D, amp=3.2 if kappa == 1 else 1.1)
    out = np.zeros(3 * M, np.complex128)
    rc = emu.emu_nsev_contspec(D, _P(q), _P(T), M, _P(out), _P(XI), kappa, NSE_DISC[disc], 2, 1)
    assert rc == 0
    rc2, ref = oracle.fnft_nsev(q, T, M, XI, kappa=kappa, disc=disc, cstype="BOTH")
    assert rc2 == 0
    assert S.rel_err(out, ref) < S.contspec_tol(oracle, q, T, kappa, disc, 5e-12)


@pytest.mark.parametrize("testcase,D,disc", [("SECH", 256, "2SPLIT4B"), ("SECH", 300, "2SPLIT2A"), ("RECT", 4, "2SPLIT8B"),
                                              ("NEGATIVE_RECT", 4, "2SPLIT2A"), ("SECH", 128, "2SPLIT8B"),
                                              ("SECH", 200, "2SPLIT3A"), ("RECT", 64, "2SPLIT1A")])
@pytest.mark.parametrize("real", [0, 1])
def test_kdvv_vs_oracle(emu, oracle, fixtures, testcase, D, disc, real):
    """fnft_kdvv pipeline (general 2x2 tree with r = -1, entries 12/22, -xi grid) in the emulator; real = 1: the
    real-coefficient path (folded negacyclic transforms, nft_real.h)."""
    from oracle.oracle import KDV_DISC
    u, T, XI, M, exact = S.kdvv_case(fixtures, testcase, D)
    T, XI = np.array(T, np.float64), np.array(XI, np.float64)
    out = np.zeros(M, np.complex128)
    rc = emu.emu_kdvv_contspec(D, _P(u), _P(T), M, _P(out), _P(XI), KDV_DISC[disc], real)
    assert rc == 0
    rc2, ref = oracle.fnft_kdvv(u, T, M, XI, disc)
    assert rc2 == 0
    # order 5..8 schemes on coarse grids: ill-conditioned coefficient form, see signals.contspec_tol
    assert S.rel_err(out, ref) < (1e-9 if disc[6] in "5678" else 1e-11)


@pytest.mark.parametrize("D,disc,bsloc", [(700, "2SPLIT2_MODAL", 2), (300, "2SPLIT4B", 2), (256, "2SPLIT4B", 0),
                                          (256, "4SPLIT4A", 2), (256, "2SPLIT2A", 1)])
def test_discrete_spectrum_vs_oracle(emu, oracle, fixtures, D, disc, bsloc):
    """Bound states (Aberth root finder + chunk-parallel Newton), norming constants and residues in
    the emulator against the oracle (numpy.roots + sequential scatterer) on the sech pulse."""
    T = np.array([-25.0, 25.0])
    q = S.sech_focusing(D)
    K = C.c_size_t(4 * D)
    bs = np.zeros(4 * D, np.complex128)
    nc = np.zeros(8 * D, np.complex128)
    loc = {0: "FAST_EIGENVALUE", 1: "NEWTON", 2: "SUBSAMPLE_AND_REFINE"}[bsloc]
    guesses = np.array([0.6j, 1.8j, 2.6j])
    if bsloc == 1:
        bs[:3] = guesses
        K = C.c_size_t(3)
    rc = emu.emu_nsev_discspec(D, _P(q), _P(T), 2, bsloc, 10, 0, 2, NSE_DISC[disc], 0, C.byref(K), _P(bs), _P(nc))
    assert rc == 0
    k = K.value
    rc2, bs_o, nc_o, res_o = oracle.fnft_nsev_ds(q, T, disc, bsloc=loc, guesses=guesses)
    assert rc2 == 0 and k == bs_o.size == 3, (rc2, k, bs[:k], bs_o)
    order = [int(np.argmin(np.abs(bs[:k] - v))) for v in bs_o]
    assert sorted(order) == [0, 1, 2]
    tol = 1e-6 if bsloc == 0 else 1e-10   # raw polynomial roots are conditioned like the polynomial
    eps_t = (T[1] - T[0]) / (D - 1)
    ups = 2 if disc.startswith("4SPLIT") else 1
    rcp, qp, _, _ = oracle.preprocess(q, eps_t, D, disc)
    rcs, a_o, _, _ = oracle.scatter_bound_states(qp, T, bs_o, ups, skip_b=True)
    for j, i in enumerate(order):
        assert abs(bs[i] - bs_o[j]) < tol
        # b = phi/psi is independent of the grid point only at a zero of a: compare it where Newton converged
        if bsloc != 0 and abs(a_o[j]) < 1e-9:
            assert abs(nc[i] - nc_o[j]) < 1e-8 * abs(nc_o[j])
            assert abs(nc[k + i] - res_o[j]) < 1e-8 * abs(res_o[j])
