"""CPU restatement of fnft_nsev_inverse (TEST INFRASTRUCTURE ONLY -- tests/, smoke and bench's cpu_baseline leg).

Follows the reference file by file (citations into /root/reference, FNFT 0.4.1):
  fnft_nsev_inverse_XI            src/fnft_nsev_inverse.c:40-65
  fnft_nsev_inverse               src/fnft_nsev_inverse.c:121-248 (argument checks in the reference's order)
  remove_boundary_conds_and_reorder_for_fft                     :251-296
  transfer matrix from rho (A = 1)                               :302-369
  transfer matrix from rho by iteration (defocusing)             :375-508
  transfer matrix from b(xi) / B(tau)                            :560-676
  add_discrete_spectrum (pure solitons / Darboux on a seed)      :680-903
  compute_eigenfunctions                                         :908-1007
  precompensate_for_cdt_phaseshifts                              :1013-1033
  poly_specfact                   src/private/fnft__poly_specfact.c:25-140
  nse_scatter_matrix (BO)         src/private/fnft__nse_scatter_matrix.c:33-86, fnft__akns_scatter_matrix.c
  poly_roots_fftgridsearch(_paraherm), misc_hausdorff_dist
                                  src/private/fnft__poly_roots_fftgridsearch.c:35-217, fnft__misc.c:53-83
FFTs are numpy's (any length); the reference uses KissFFT at the same lengths (fft_wrapper_next_fft_length = the
next 2-3-5-smooth length), so results agree to round-off, not bit for bit.
Pinned by the reference's own tests of test/fnft_nsev_inverse (tests/test_inverse_oracle.py): the sech data files as
fixtures, the analytic cases with the files' error bounds.
"""
import numpy as np

from .oracle import nse_finvscatter

SUCCESS, EC_INVALID_ARGUMENT, EC_SANITY = 0, 2, 7   # fnft_errwarn.h:44-108 ordinals used by the checks below
CSTYPES = ("REFLECTION_COEFFICIENT", "B_OF_XI", "B_OF_TAU")
CSMETHODS = ("DEFAULT", "TFMATRIX_CONTAINS_REFL_COEFF", "TFMATRIX_CONTAINS_AB_FROM_ITER", "USE_SEED_POTENTIAL_INSTEAD")
DSTYPES = ("NORMING_CONSTANTS", "RESIDUES")


def next_fast_size(n):
    """kiss_fft_next_fast_size: smallest m >= n with only the factors 2, 3, 5."""
    while True:
        m = n
        for f in (2, 3, 5):
            while m % f == 0:
                m //= f
        if m <= 1:
            return n
        n += 1


def default_opts():
    return {"discretization": "2SPLIT2A", "contspec_type": "REFLECTION_COEFFICIENT",
            "contspec_inversion_method": "DEFAULT", "discspec_type": "NORMING_CONSTANTS", "max_iter": 100,
            "oversampling_factor": 8}


def nsev_inverse_XI(D, T, M):
    """:40-65; both admissible discretizations have degree 1 and no upsampling."""
    eps_t = (T[1] - T[0]) / (D - 1)
    z = np.array([np.exp(2j * np.pi * (M // 2 + 1) / M), -1.0 + 0j])
    lam = np.log(z) / (2j * eps_t)
    return [float(lam[0].real), float(lam[1].real)]


def phase_factor_rho(eps_t, T1):        # fnft__nse_discretization.c:240-258, degree 1, boundary coefficient 0.5
    return -2.0 * (T1 + eps_t * 0.5) + eps_t


def phase_factor_b(eps_t, D, T):        # fnft__nse_discretization.c:319-379
    return -eps_t * D - (T[1] + eps_t * 0.5) - (T[0] - eps_t * 0.5) + eps_t


def poly_specfact(poly, oversampling_factor, kappa):
    """fnft__poly_specfact.c:25-140.  poly: deg+1 coefficients; returns (result[deg+1], ill_posed_warning)."""
    poly = np.asarray(poly, np.complex128)
    deg = poly.size - 1
    M = next_fast_size((deg + 1) * oversampling_factor)
    buf = np.zeros(M, np.complex128)
    buf[:deg + 1] = poly
    P = np.fft.fft(buf)
    tol = np.sqrt(np.finfo(float).eps)
    warn = False
    if kappa == 0:
        a = np.abs(P)
        warn = bool(np.any(a < tol))
        x = np.log(a).astype(np.complex128)
    elif kappa == -1:
        x = (0.5 * np.log(1.0 + np.abs(P) ** 2)).astype(np.complex128)
    else:
        a2 = np.abs(P) ** 2
        warn = bool(np.any(a2 > 1.0 - tol))
        x = 0.5 * np.log((1.0 - a2).astype(np.complex128))          # CLOG of a possibly negative real
    X = np.fft.fft(x)
    X[0] = 0.0
    X[1:M // 2 - 1] *= -1j / M
    X[M // 2 - 1] = 0.0
    X[M // 2:] *= 1j / M
    y = np.fft.ifft(X) * M                                           # un-normalised inverse, as the reference's plan
    resp = np.exp(x - 1j * y) / M
    out = np.fft.ifft(resp) * M
    return np.conj(out[deg::-1][:deg + 1]), warn


def _remove_bc_and_reorder(contspec, XI, D, T, cstype):
    """:251-296 -- multiplies contspec IN PLACE (the reference does) and returns the FFT-ordered copy."""
    M = contspec.size
    eps_t = (T[1] - T[0]) / (D - 1)
    eps_xi = (XI[1] - XI[0]) / (M - 1)
    pf = phase_factor_rho(eps_t, T[1]) if cstype == "REFLECTION_COEFFICIENT" else phase_factor_b(eps_t, D, T)
    xi = XI[0] + np.arange(M) * eps_xi
    contspec *= np.exp(-1j * xi * pf)
    return _reorder(contspec)


def _reorder(c):
    M = c.size
    out = np.empty(M, np.complex128)
    out[:M // 2 + 1] = c[M // 2 - 1:M]
    out[M // 2 + 1:] = c[:M // 2 - 1]
    return out


def _tm_refl(contspec, XI, D, T, deg, kappa):
    M = contspec.size
    b = np.fft.fft(_remove_bc_and_reorder(contspec, XI, D, T, "REFLECTION_COEFFICIENT"))
    tm = np.zeros((4, deg + 1), np.complex128)
    i0 = 0 if deg <= M - 1 else deg - (M - 1)
    i = np.arange(i0, deg + 1)
    tm[1, i] = -kappa * np.conj(b[M - 1 - deg + i] / M)
    tm[2, i] = b[deg - i] / M
    tm[0, deg] = 1.0
    tm[3, 0] = 1.0
    return tm


def _tm_refl_iter(contspec, XI, D, T, deg, kappa, max_iter):
    M = contspec.size
    if D < 2 or (D & (D - 1)) or M != D or kappa != -1:
        return EC_INVALID_ARGUMENT, None
    cr = _remove_bc_and_reorder(contspec, XI, D, T, "REFLECTION_COEFFICIENT")
    prev_change, prev_diff = np.inf, np.inf
    a = bco = None
    for _ in range(max_iter):
        fin = cr / np.sqrt(1.0 + kappa * np.abs(cr) ** 2) / D
        bco = np.fft.fft(fin)[::-1].copy()
        a, _w = poly_specfact(bco, 32, kappa)
        ph = np.angle(np.fft.ifft(a[::-1]) * D)
        cur = float(np.sum(np.abs(ph)) / D)
        cr = _reorder(contspec) * np.exp(1j * ph)
        diff = abs(cur - prev_change)
        if diff < 10 * np.finfo(float).eps:
            break
        prev_change = cur
        if diff > 0.9 * prev_diff:
            break
        prev_diff = diff
    tm = np.zeros((4, deg + 1), np.complex128)
    tm[0, 1:D + 1] = a
    tm[1, :D] = -kappa * np.conj(bco[::-1])
    tm[2, 1:D + 1] = bco
    tm[3, :D] = a[::-1]
    return SUCCESS, tm


def _tm_b_of_xi(contspec, XI, D, T, deg, kappa, oversampling):
    M = contspec.size
    b = np.fft.fft(_remove_bc_and_reorder(contspec, XI, D, T, "B_OF_XI"))
    tm = np.zeros((4, deg + 1), np.complex128)
    i0 = 0 if deg <= M - 1 else deg - (M - 1)
    i = np.arange(i0, deg + 1)
    tm[1, i] = -kappa * np.conj(b[M - 1 - deg + i] / M)
    tm[2, i] = b[deg - i] / M
    tm[0], _w = poly_specfact(tm[2], oversampling, kappa)
    tm[3] = tm[0][::-1]
    return tm


def _tm_B_of_tau(contspec, D, T, deg, kappa, oversampling):
    eps_t = (T[1] - T[0]) / (D - 1)
    b = 2.0 * eps_t * np.asarray(contspec, np.complex128).copy()
    b[0] *= 0.5
    b[D - 1] *= 0.5
    a, _w = poly_specfact(b, oversampling, kappa)
    tm = np.zeros((4, deg + 1), np.complex128)
    tm[0, 1:D + 1] = a
    tm[2, 1:D + 1] = b
    tm[1, :D] = -kappa * np.conj(b[::-1])
    tm[3, :D] = a[::-1]
    return tm


def compute_eigenfunctions(bound_states, q, T):
    """:908-1007: phi, psi [K, 2, D] by half steps of the exponential (Boffetta-Osborne) integrator."""
    D, K = q.size, len(bound_states)
    h = ((T[1] - T[0]) / (D - 1)) / 2
    phi = np.zeros((K, 2, D), np.complex128)
    psi = np.zeros((K, 2, D), np.complex128)
    for i, l in enumerate(bound_states):
        ks = -np.abs(q) ** 2 - l * l
        k = np.sqrt(ks.astype(np.complex128))
        ch = np.cosh(k * h)
        with np.errstate(all="ignore"):
            sh = np.where(ks != 0, np.sinh(k * h) / np.where(ks != 0, k, 1.0), 0.0)
        u1 = 1j * l * sh
        nz = ks != 0
        p1, p2 = np.exp(-1j * l * T[0]), 0.0j
        phi[i, 0, 0], phi[i, 1, 0] = p1, p2
        for n in range(1, D):
            if nz[n - 1]:
                p1, p2 = ((ch[n - 1] - u1[n - 1]) * p1 + q[n - 1] * sh[n - 1] * p2,
                          -np.conj(q[n - 1]) * sh[n - 1] * p1 + (ch[n - 1] + u1[n - 1]) * p2)
            if nz[n]:
                p1, p2 = ((ch[n] - u1[n]) * p1 + q[n] * sh[n] * p2,
                          -np.conj(q[n]) * sh[n] * p1 + (ch[n] + u1[n]) * p2)
            phi[i, 0, n], phi[i, 1, n] = p1, p2
        s1, s2 = 0.0j, np.exp(1j * l * T[1])
        psi[i, 0, D - 1], psi[i, 1, D - 1] = s1, s2
        for n in range(D - 1, 0, -1):
            for m in (n, n - 1):
                if nz[m]:
                    scl = (ch[m] - u1[m]) * (ch[m] + u1[m]) - (-np.conj(q[m]) * sh[m]) * (q[m] * sh[m])
                    s1, s2 = (((ch[m] + u1[m]) * s1 - q[m] * sh[m] * s2) / scl,
                              (np.conj(q[m]) * sh[m] * s1 + (ch[m] - u1[m]) * s2) / scl)
            psi[i, 0, n - 1], psi[i, 1, n - 1] = s1, s2
    return phi, psi


def add_discrete_spectrum(bound_states, normconsts_or_residues, q, T, contspec_flag, opts, scatter_a=None):
    """:680-903; q is modified in place.  scatter_a(q, T, lam) -> a(lam) of the seed (BO scheme), needed for residues on
    top of a continuous spectrum (:776-781)."""
    D, K = q.size, len(bound_states)
    eps_t = (T[1] - T[0]) / (D - 1)
    t = T[0] + eps_t * np.arange(D)
    zc = int(np.argmax(t >= 0.0)) if np.any(t >= 0.0) else 0
    order = list(range(K))
    bs = list(np.asarray(bound_states, np.complex128))
    nc = list(np.asarray(normconsts_or_residues, np.complex128))
    for i in range(K):                                                # the reference's exchange sort, :742-754
        for j in range(i + 1, K):
            if bs[i].imag < bs[j].imag:
                bs[i], bs[j] = bs[j], bs[i]
                nc[i], nc[j] = nc[j], nc[i]
    del order
    for i in range(K - 1):
        if bs[i + 1] == bs[i]:
            return EC_SANITY
    bs, nc = np.array(bs), np.array(nc)
    diff = 2j * bs.imag
    method = opts["contspec_inversion_method"]
    if opts["discspec_type"] == "RESIDUES":
        acs = scatter_a(q, T, bs) if contspec_flag else np.ones(K, np.complex128)
        for i in range(K):
            tmp = acs[i]
            for j in range(K):
                if j != i:
                    tmp = tmp * (bs[i] - bs[j]) / (bs[i] - np.conj(bs[j]))
            nc[i] = (nc[i] / diff[i]) * tmp
    if contspec_flag == 0 and method != "USE_SEED_POTENTIAL_INSTEAD":
        def solitons(tt, ncv, sign):
            rhok = ncv[:, None] * np.exp(sign * 2j * bs[:, None] * tt[None, :])
            qt = np.zeros(tt.size, np.complex128)
            for i in range(K):
                rho = rhok[i].copy()
                rhoc = np.conj(rho)
                f = diff[i] / (1.0 + np.abs(rho) ** 2)
                qt = qt + 2j * rhoc * f
                for j in range(i + 1, K):
                    rhok[j] = ((bs[j] - bs[i]) * rhok[j] + (rhok[j] - rho) * f) / (
                        bs[j] - np.conj(bs[i]) - (1.0 + rhoc * rhok[j]) * f)
            return qt
        with np.errstate(all="ignore"):
            q[zc:] = solitons(t[zc:], nc, +1.0)
            q[:zc] = np.conj(solitons(t[:zc], 1.0 / nc, -1.0))
        return SUCCESS
    if (contspec_flag == 0 and method == "USE_SEED_POTENTIAL_INSTEAD") or (
            contspec_flag == 1 and method != "USE_SEED_POTENTIAL_INSTEAD"):
        phi, psi = compute_eigenfunctions(bs, q, T)
        qn = q.copy()
        S1 = np.zeros((K, D), np.complex128)
        S2 = np.zeros((K, D), np.complex128)
        for i in range(K):
            p1, p2, s1, s2 = phi[i, 0].copy(), phi[i, 1].copy(), psi[i, 0].copy(), psi[i, 1].copy()
            for j in range(i):
                p1, p2 = (bs[i] - S1[j]) * p1 - S2[j] * p2, np.conj(S2[j]) * p1 + (bs[i] - np.conj(S1[j])) * p2
                s1, s2 = (bs[i] - S1[j]) * s1 - S2[j] * s2, np.conj(S2[j]) * s1 + (bs[i] - np.conj(S1[j])) * s2
            beta = (p1 - nc[i] * s1) / (p2 - nc[i] * s2)
            ab = np.abs(beta) ** 2
            S1[i] = (ab * bs[i] + np.conj(bs[i])) / (1.0 + ab)
            S2[i] = (2j * bs[i].imag * beta) / (1.0 + ab)
            qn = qn - 2j * S2[i]
        q[:] = qn
        return SUCCESS
    return EC_INVALID_ARGUMENT


def fnft_nsev_inverse(M, contspec, XI, bound_states, normconsts_or_residues, D, T, kappa, opts=None, q_seed=None,
                      scatter_a=None):
    """:121-248.  Returns (rc, q).  contspec (if given) is modified in place like the reference's argument."""
    o = default_opts()
    o.update(opts or {})
    K = 0 if bound_states is None else len(bound_states)
    if M > 0 and contspec is None:
        return EC_INVALID_ARGUMENT, None
    if contspec is not None and M % 2 != 0:
        return EC_INVALID_ARGUMENT, None
    if contspec is not None and M < D:
        return EC_INVALID_ARGUMENT, None
    if D < 2 or (D & (D - 1)) != 0:
        return EC_INVALID_ARGUMENT, None
    if T is None or not (T[0] < T[1]):
        return EC_INVALID_ARGUMENT, None
    if kappa not in (1, -1):
        return EC_INVALID_ARGUMENT, None
    if K > 0 and kappa != 1:
        return EC_SANITY, None
    if K > 0 and any(complex(b).imag <= 0 for b in bound_states):
        return EC_SANITY, None
    if K > 0 and normconsts_or_residues is None:
        return EC_INVALID_ARGUMENT, None
    if o["discretization"] not in ("2SPLIT2A", "2SPLIT2_MODAL"):
        return EC_INVALID_ARGUMENT, None
    if contspec is None and K == 0:
        return EC_SANITY, None
    if XI is None and contspec is not None and o["contspec_type"] != "B_OF_TAU":
        return EC_INVALID_ARGUMENT, None
    q = np.zeros(D, np.complex128) if q_seed is None else np.array(q_seed, np.complex128)
    flag = 0
    if contspec is not None:
        flag = 1
        deg = D
        ct, method = o["contspec_type"], o["contspec_inversion_method"]
        if ct == "REFLECTION_COEFFICIENT":
            if K > 0:                                                 # :1013-1033
                eps_xi = (XI[1] - XI[0]) / (M - 1)
                xi = XI[0] + eps_xi * np.arange(M)
                for b in bound_states:
                    contspec *= (xi - b) / (xi - np.conj(b))
            if method in ("DEFAULT", "TFMATRIX_CONTAINS_REFL_COEFF"):
                tm = _tm_refl(contspec, XI, D, T, deg, kappa)
            elif method == "TFMATRIX_CONTAINS_AB_FROM_ITER":
                rc, tm = _tm_refl_iter(contspec, XI, D, T, deg, kappa, o["max_iter"])
                if rc:
                    return rc, None
            else:
                return EC_INVALID_ARGUMENT, None
        elif ct == "B_OF_XI":
            tm = _tm_b_of_xi(contspec, XI, D, T, deg, kappa, o["oversampling_factor"])
        elif ct == "B_OF_TAU":
            if M != D or T[0] != -T[1] or method != "DEFAULT":
                return EC_INVALID_ARGUMENT, None
            tm = _tm_B_of_tau(contspec, D, T, deg, kappa, o["oversampling_factor"])
        else:
            return EC_INVALID_ARGUMENT, None
        eps_t = (T[1] - T[0]) / (D - 1)
        rc, qq = nse_finvscatter(tm, eps_t, kappa, o["discretization"])
        if rc:
            return rc, None
        q[:] = qq
    if K > 0:
        rc = add_discrete_spectrum(bound_states, normconsts_or_residues, q, T, flag, o, scatter_a)
        if rc:
            return rc, None
    return SUCCESS, q


def nse_scatter_matrix(q, eps_t, kappa, lam, derivative=True):
    """fnft__nse_scatter_matrix, BO scheme (src/private/fnft__akns_scatter_matrix.c: T_n = [[U, 0], [U', U]], S = T_{D-1}
    ... T_0): [K, 8] = [S11 S12 S21 S22 S11' S12' S21' S22'] per lambda (4 columns without the derivative)."""
    q = np.asarray(q, np.complex128)
    r = -kappa * np.conj(q)
    out = []
    for l in np.asarray(lam, np.complex128):
        S = np.eye(4, dtype=np.complex128)
        for n in range(q.size - 1, -1, -1):
            ks = q[n] * r[n] - l * l
            k = np.sqrt(ks + 0j)
            ch, sh = np.cosh(k * eps_t), np.sinh(k * eps_t) / k
            U = np.array([[ch - 1j * l * sh, q[n] * sh], [r[n] * sh, ch + 1j * l * sh]])
            g = l * (eps_t * ch - sh) / ks
            Ud = np.array([[1j * eps_t * l * l * ch / ks - (l * eps_t + 1j + 1j * l * l / ks) * sh, -q[n] * g],
                           [-r[n] * g, -1j * eps_t * l * l * ch / ks - (l * eps_t - 1j - 1j * l * l / ks) * sh]])
            Tn = np.zeros((4, 4), np.complex128)
            Tn[:2, :2] = U
            Tn[2:, 2:] = U
            Tn[2:, :2] = Ud
            S = S @ Tn
        row = [S[0, 0], S[0, 1], S[1, 0], S[1, 1]]
        if derivative:
            row += [S[2, 0], S[2, 1], S[3, 0], S[3, 1]]
        out.append(row)
    return np.array(out)


def _chirpz(p, A, W, M):
    """fnft__poly_chirpz.c:28-29: p(1/(A W^-m)), m < M, p highest power first."""
    m = np.arange(M)
    z = 1.0 / (A * W ** (-m.astype(float)))
    return np.polyval(np.asarray(p, np.complex128), z)


def poly_roots_fftgridsearch(p, M, PHI):
    """src/private/fnft__poly_roots_fftgridsearch.c:35-151."""
    p = np.asarray(p, np.complex128)
    eps = (PHI[1] - PHI[0]) / (M - 1)
    W = np.exp(1j * eps)
    vals = np.concatenate([_chirpz(p, (1.0 + k * eps) * np.exp(-1j * PHI[0]), W, M) for k in (-1, 0, 1)])
    roots = []
    for i in range(1, M - 1):
        tmp = abs(vals[M + i])
        nb = [vals[i - 1], vals[i], vals[i + 1], vals[M + i - 1], vals[M + i + 1], vals[2 * M + i - 1], vals[2 * M + i],
              vals[2 * M + i + 1]]
        if any(tmp > abs(v) for v in nb):
            continue
        z0 = np.exp(1j * (PHI[0] + i * eps))
        y0 = vals[M + i]
        c, den = 0.0j, 0.0
        for j in range(i - 1, i + 2):
            for k in (-1, 0, 1):
                if j == 0 and k == 0:
                    continue
                zi = (1 - k * eps) * np.exp(1j * (PHI[0] + j * eps))
                yi = vals[(k + 1) * M + j]
                c += np.conj(zi - z0) * (yi - y0)
                den += abs(zi - z0) ** 2
        c /= den
        if c == 0:
            if y0 != 0:
                continue
            zr = z0
        else:
            zr = z0 - y0 / c
            if abs(zr - z0) > eps:
                continue
        roots.append(zr)
    return np.array(roots, np.complex128)


def poly_roots_fftgridsearch_paraherm(p, M, PHI):
    """:159-217 (even degree)."""
    p = np.asarray(p, np.complex128)
    deg = p.size - 1
    eps = (PHI[1] - PHI[0]) / (M - 1)
    v = _chirpz(p, np.exp(-1j * PHI[0]), np.exp(1j * eps), M)
    N = deg // 2 + 1
    phi_all = PHI[0] + eps * np.arange(M)
    v = v * np.exp(-1j * phi_all * (N - 1))
    roots = []
    for i in range(1, M):
        if v[i - 1].real * v[i].real <= 0.0:
            phi1 = PHI[0] + eps * (i - 1)
            phi2 = phi1 + eps
            if v[i - 1] != v[i]:
                phi = phi1 - (v[i - 1] * (phi2 - phi1) / (v[i] - v[i - 1])).real
            else:
                phi = 0.5 * (phi1 + phi2)
            roots.append(np.exp(1j * phi))
    return np.array(roots, np.complex128)


def hausdorff(a, b):
    """misc_hausdorff_dist, src/private/fnft__misc.c:53-83."""
    a, b = np.asarray(a, np.complex128), np.asarray(b, np.complex128)
    if a.size == 0 or b.size == 0:
        return np.inf
    d = np.abs(a[:, None] - b[None, :])
    return float(max(d.min(axis=1).max(), d.min(axis=0).max()))
