"""Deterministic test signals and closed-form spectra shared by the CPU and GPU tests.

The signal definitions restate the inputs of the reference's own test cases
(src/private/fnft__nsev_testcases.c:142-567 and test/fnft__akns_fscatter/*.c); values come from
tests/golden/reference_fixtures.json.
"""
import numpy as np


def l2c(lst):
    a = np.asarray(lst, dtype=np.float64)
    return a[..., 0] + 1j * a[..., 1]


def rel_err(numer, exact):
    """misc_rel_err, src/private/fnft__misc.c:41-51: sum|d| / sum|exact|."""
    numer = np.asarray(numer)
    exact = np.asarray(exact)
    return float(np.sum(np.abs(numer - exact)) / np.sum(np.abs(exact)))


def tgrid(T, D):
    return T[0] + np.arange(D) * (T[1] - T[0]) / (D - 1)


def sech(x):
    return 1.0 / np.cosh(x)


def sech_focusing(D, T=(-25.0, 25.0), amp=3.2):
    """fnft__nsev_testcases.c:176-177"""
    return (1j * amp * sech(tgrid(T, D))).astype(np.complex128)


def sech_defocusing(D, T=(-2.0, 1.5)):
    """fnft__nsev_testcases.c:489-493"""
    Q, GAM, F = 1.0, 1.0 / 25.0, 1.5
    t = tgrid(T, D)
    return (-np.conj(Q / GAM * sech(t / GAM).astype(np.complex128) ** (1 - 2j * F))).astype(np.complex128)


def truncated_soliton(D, T=(0.0, 15.0)):
    """fnft__nsev_testcases.c:534-541"""
    be = 0.55
    q = (-2.0 * be * sech(2.0 * be * tgrid(T, D))).astype(np.complex128)
    q[0] *= 0.5
    return q


def truncated_soliton_contspec(XI, M):
    be = 0.55
    xi = XI[0] + np.arange(M) * (XI[1] - XI[0]) / (M - 1)
    return -1j * be / xi * (xi + 1j * be) / (xi - 1j * be)


def sech_focusing_analytic(xi, amp=3.2):
    """Satsuma-Yajima a(xi), b(xi) for q = i*A*sech(t); fnft__nsev_testcases.c:148-166.
    double precision through scipy's complex gamma (good to ~1e-14 on this grid)."""
    from scipy.special import gamma
    xi = np.asarray(xi, dtype=np.float64)
    a = gamma(-1j * xi + 0.5) ** 2 / (gamma(-1j * xi + amp + 0.5) * gamma(-1j * xi - amp + 0.5))
    b = 1j * np.sin(np.pi * amp) / np.cosh(np.pi * xi)
    return a, b


def akns_test_signal(D=8):
    """test/fnft__akns_fscatter/*.c: q, r, and the 5 evaluation points."""
    n = np.arange(1, D + 1, dtype=np.float64)
    q = (0.41 * np.cos(n) + 0.59j * np.sin(0.28 * n)) * 50
    r = (0.33 * np.sin(n) + 0.85j * np.cos(0.43 * n)) * 25
    z = np.exp(1j * np.array([0.0, np.pi / 4, 9 * np.pi / 14, 4 * np.pi / 3, -np.pi / 5]))
    return q.astype(np.complex128), r.astype(np.complex128), z


def fmult_test_input(deg, n):
    """test/fnft__poly/fnft__poly_fmult2x2_test_*.c input rule; returns p[4, n*(deg+1)]."""
    i = np.arange(n * (deg + 1), dtype=np.float64)
    return np.stack([np.sqrt(i + 1.0) * (np.cos(i + 0.1 * e) + 1j * np.sin(-2.0 * i + 0.1 * e))
                     for e in range(4)]).astype(np.complex128)


def splitmix64(seed):
    """splitmix64 stream (SURVEY 8d cfg 3): yields uniform doubles in [0,1)."""
    mask = (1 << 64) - 1
    state = seed & mask
    while True:
        state = (state + 0x9E3779B97F4A7C15) & mask
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z = z ^ (z >> 31)
        yield (z >> 11) * (1.0 / 9007199254740992.0)


def batch_signal(k, D, T=(-25.0, 25.0)):
    """cfg 3 signal k: A*sech(t-tau)*exp(i*w*t), (A,tau,w) from splitmix64(0x5EED0000+k)."""
    g = splitmix64(0x5EED0000 + k)
    A = 0.5 + 3.0 * next(g)
    tau = -5.0 + 10.0 * next(g)
    w = -2.0 + 4.0 * next(g)
    t = tgrid(T, D)
    return (A * sech(t - tau) * np.exp(1j * w * t)).astype(np.complex128)


def poly_matmul_direct(P, Q):
    """Exact-arithmetic-order reference product of two 2x2 polynomial matrices given as
    [4, d+1] arrays (entry-major 11,12,21,22; highest power first) via numpy convolution."""
    c = np.convolve
    return np.stack([
        c(P[0], Q[0]) + c(P[1], Q[2]),
        c(P[0], Q[1]) + c(P[1], Q[3]),
        c(P[2], Q[0]) + c(P[3], Q[2]),
        c(P[2], Q[1]) + c(P[3], Q[3]),
    ])


def tree_direct(p, deg, n):
    """Ordered product P_0 P_1 ... P_{n-1} by direct convolution in long double-free numpy
    (balanced tree like the reference).  p: [4, n*(deg+1)]."""
    p = np.asarray(p)
    mats = [p[:, j * (deg + 1):(j + 1) * (deg + 1)] for j in range(n)]
    while len(mats) > 1:
        nxt = []
        for i in range(0, len(mats) - 1, 2):
            nxt.append(poly_matmul_direct(mats[i], mats[i + 1]))
        if len(mats) % 2:
            nxt.append(mats[-1])
        mats = nxt
    return mats[0]


def contspec_tol(oracle, q, T, kappa, disc, floor):
    """Tolerance for comparing two FFT-tree evaluations of the same spectrum.  Both carry an
    absolute error of a few eps * max|c| per coefficient c of the transfer matrix (times 2^W);
    evaluating a polynomial of degree deg adds these up to ~sqrt(deg) * eps * max|c|.  For the
    low-order schemes on fine grids max|c| ~ 1 and the floor applies; the order 5..8 schemes on
    coarse grids have max|c| up to 1e4 (Richardson weights, large steps) and the oracle itself is
    that far from a direct long-double evaluation."""
    eps_t = (T[1] - T[0]) / (len(q) - 1)
    rc, deg, tm, W = oracle.nse_fscatter(q, eps_t, kappa, disc)
    cmax = float(np.max(np.abs(tm))) * 2.0 ** W
    return max(floor, 100 * 2.220446049250313e-16 * cmax * np.sqrt(deg))


# ---- KdV test signals (src/private/fnft__kdvv_testcases.c:88-262) --------------------------------
def kdvv_sech(D, T=(-16.0, 15.0)):
    return (3.2 * sech(tgrid(T, D)) ** 2).astype(np.complex128)


def kdvv_rect(D, T=(-1.0, 2.0), sign=1.0):
    eps_t = (T[1] - T[0]) / (D - 1)
    t = T[0] + np.arange(D) * eps_t
    u = np.where(np.abs(t) == 0.5, 0.5, np.where(np.abs(t) < 0.5, 1.0, 0.0))
    return (sign * u).astype(np.complex128)


def kdvv_case(fixtures, testcase, D):
    """(u, T, XI, M, exact contspec) of one reference KdV test case."""
    key = {"SECH": "kdvv_sech", "RECT": "kdvv_rect", "NEGATIVE_RECT": "kdvv_negative_rect"}[testcase]
    fx = fixtures[key]
    T = fx["T"]
    XI = [float(fx["XI"][0]), 15.0 * np.pi / 32.0 if isinstance(fx["XI"][1], str) else float(fx["XI"][1])]
    if testcase == "SECH":
        u = kdvv_sech(D, T)
    else:
        u = kdvv_rect(D, T, 1.0 if testcase == "RECT" else -1.0)
    return u, T, XI, fx["M"], l2c(fx["contspec"])


# ---- discrete spectrum error measures of the reference harness ------------------------------------
def hausdorff(a, b):
    """fnft__misc.c:53-83"""
    a, b = np.asarray(a), np.asarray(b)
    d = np.abs(a[:, None] - b[None, :])
    return float(max(d.min(axis=1).max(), d.min(axis=0).max()))


def ds_errors(bs, nc, res, bs_exact, nc_exact, res_exact):
    """fnft__nsev_testcases.c:650-700: Hausdorff distance of the eigenvalues; norming constants and
    residues matched to the nearest exact eigenvalue, summed and normalised."""
    bs, bs_exact = np.asarray(bs), np.asarray(bs_exact)
    if bs.size == 0 or bs_exact.size == 0:
        return [float("nan")] * 3
    out = [hausdorff(bs, bs_exact)]
    for v, ex in ((nc, nc_exact), (res, res_exact)):
        num = nrm = 0.0
        for i in range(bs.size):
            j = int(np.argmin(np.abs(bs[i] - bs_exact)))
            num += abs(v[i] - ex[j])
            nrm += abs(ex[i]) if i < len(ex) else 0.0
        out.append(num / nrm if nrm > 0 else num)
    return out
