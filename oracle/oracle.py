"""ctypes front-end of oracle/liboracle.so (the plain-C restatement of the reference algorithm).

TEST INFRASTRUCTURE ONLY -- see oracle/fnft_oracle.h.  All arrays are numpy complex128/float64.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

NSE_DISC = {
    "2SPLIT2_MODAL": 0, "BO": 1, "2SPLIT1A": 2, "2SPLIT1B": 3, "2SPLIT2A": 4, "2SPLIT2B": 5,
    "2SPLIT2S": 6, "2SPLIT3A": 7, "2SPLIT3B": 8, "2SPLIT3S": 9, "2SPLIT4A": 10, "2SPLIT4B": 11,
    "2SPLIT5A": 12, "2SPLIT5B": 13, "2SPLIT6A": 14, "2SPLIT6B": 15, "2SPLIT7A": 16,
    "2SPLIT7B": 17, "2SPLIT8A": 18, "2SPLIT8B": 19, "4SPLIT4A": 20, "4SPLIT4B": 21,
}
AKNS_DISC = {
    "2SPLIT2_MODAL": 0, "2SPLIT1A": 1, "2SPLIT1B": 2, "2SPLIT2A": 3, "2SPLIT2B": 4, "2SPLIT2S": 5,
    "2SPLIT3A": 6, "2SPLIT3B": 7, "2SPLIT3S": 8, "2SPLIT4A": 9, "2SPLIT4B": 10,
    "2SPLIT5A": 11, "2SPLIT5B": 12, "2SPLIT6A": 13, "2SPLIT6B": 14, "2SPLIT7A": 15, "2SPLIT7B": 16,
    "2SPLIT8A": 17, "2SPLIT8B": 18, "4SPLIT4A": 20, "4SPLIT4B": 21,
}
KDV_DISC = {n: i for i, n in enumerate(
    ["2SPLIT1A", "2SPLIT1B", "2SPLIT2A", "2SPLIT2B", "2SPLIT2S", "2SPLIT3A", "2SPLIT3B", "2SPLIT3S",
     "2SPLIT4A", "2SPLIT4B", "2SPLIT5A", "2SPLIT5B", "2SPLIT6A", "2SPLIT6B", "2SPLIT7A", "2SPLIT7B",
     "2SPLIT8A", "2SPLIT8B"])}
CSTYPE = {"RHO": 0, "AB": 1, "BOTH": 2}


def poly_roots(c, dense_max=2500):
    """All roots of the polynomial with coefficients c (highest power first).  The reference calls
    eiscor's structured QR (fnft__poly_roots_fasteigen.c:29-48, Fortran, not rebuilt here); the checker
    uses LAPACK through numpy.roots up to degree dense_max and a vectorised Ehrlich-Aberth iteration
    (Bini's start values, Horner in z or 1/z) above, where the dense companion matrix is too slow."""
    c = np.asarray(c, np.complex128)
    nz = np.flatnonzero(c != 0)
    if nz.size == 0:
        return np.zeros(0, np.complex128)
    lead, trail = nz[0], c.size - 1 - nz[-1]
    cc = c[nz[0]:nz[-1] + 1]
    n = cc.size - 1
    extra = np.concatenate([np.full(lead, np.inf + 0j), np.zeros(trail, np.complex128)])
    if n <= dense_max:
        return np.concatenate([np.roots(cc), extra])
    a = np.abs(cc[::-1])
    la = np.log(np.maximum(a, 1e-300))
    hull = []
    for k in range(n + 1):
        while len(hull) >= 2:
            k1, k2 = hull[-2], hull[-1]
            if (la[k2] - la[k1]) * (k - k1) <= (la[k] - la[k1]) * (k2 - k1):
                hull.pop()
            else:
                break
        hull.append(k)
    z = np.zeros(n, np.complex128)
    pos = 0
    for h in range(len(hull) - 1):
        k1, k2 = hull[h], hull[h + 1]
        m = k2 - k1
        r = np.exp((la[k1] - la[k2]) / m)
        z[pos:pos + m] = r * np.exp(1j * (2 * np.pi * np.arange(m) / m + 2 * np.pi * h / n + 0.7))
        pos += m
    cr = cc[::-1]
    with np.errstate(all="ignore"):
        for _ in range(100):
            ins = np.abs(z) <= 1
            w = np.zeros(n, np.complex128)
            zi = z[ins]
            p = np.full(zi.shape, cc[0], np.complex128)
            dp = np.zeros(zi.shape, np.complex128)
            for k in range(1, n + 1):
                dp = dp * zi + p
                p = p * zi + cc[k]
            w[ins] = p / dp
            zo = z[~ins]
            y = 1.0 / zo
            qv = np.full(y.shape, cr[0], np.complex128)
            dq = np.zeros(y.shape, np.complex128)
            for k in range(1, n + 1):
                dq = dq * y + qv
                qv = qv * y + cr[k]
            w[~ins] = 1.0 / (n / zo - y * y * dq / qv)
            s = np.zeros(n, np.complex128)
            B = 256
            for i0 in range(0, n, B):
                d = z[i0:i0 + B, None] - z[None, :]
                d[np.arange(d.shape[0]), np.arange(i0, i0 + d.shape[0])] = np.inf
                s[i0:i0 + B] = np.sum(1.0 / d, axis=1)
            corr = w / (1 - w * s)
            corr[~np.isfinite(corr)] = 0.0
            z = z - corr
            if np.max(np.abs(corr) / np.maximum(np.abs(z), 1e-300)) < 1e-14:
                break
    return np.concatenate([z, extra])


def build_oracle(force=False):
    """Compile oracle/liboracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "fnft_oracle.c")
    if (not force and os.path.exists(_LIB)
            and os.path.getmtime(_LIB) >= max(os.path.getmtime(src),
                                              os.path.getmtime(os.path.join(_HERE, "fnft_oracle.h")))):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


def _conv2x2(A, B):
    """Product of two 2x2 polynomial matrices [4, n] x [4, m] (highest power first) by FFT convolution."""
    n = A.shape[1] + B.shape[1] - 1
    L = 1 << int(np.ceil(np.log2(max(n, 1))))
    FA, FB = np.fft.fft(A, L, axis=1), np.fft.fft(B, L, axis=1)
    FC = np.stack([FA[0] * FB[0] + FA[1] * FB[2], FA[0] * FB[1] + FA[1] * FB[3],
                   FA[2] * FB[0] + FA[3] * FB[2], FA[2] * FB[1] + FA[3] * FB[3]])
    return np.fft.ifft(FC, axis=1)[:, :n]


def nse_finvscatter(tm, eps_t, kappa, disc):
    """Fast inverse scattering by layer peeling, src/private/fnft__nse_finvscatter.c:66-232 (recursion) and
    :234-366 (driver): tm [4, deg+1] (highest power first, as fnft__nse_fscatter returns it un-normalised) ->
    (rc, q[deg]).  Only the two degree-1 schemes with a base case (:164-211): 2SPLIT2_MODAL and 2SPLIT2A; the
    number of samples must be a power of two (:259-260).  Products by FFT convolution as in the reference."""
    tm = np.asarray(tm, np.complex128)
    deg = tm.shape[1] - 1
    if deg < 2 or (deg & (deg - 1)) != 0:
        return 5, None
    if disc not in ("2SPLIT2_MODAL", "2SPLIT2A"):
        return 2, None
    modal = disc == "2SPLIT2_MODAL"
    q = np.zeros(deg, np.complex128)
    state = {"rc": 0}

    def peel(T, want_inverse, q_out):
        """T [4, d+1] -> inverse up to a power of z ([4, d+1]) if wanted; fills q_out (d samples)."""
        d = T.shape[1] - 1
        if state["rc"]:
            return None
        if d == 1:                                                       # :158-218
            Q = -kappa * np.conj(T[2, 1] / T[0, 1])
            den = 1.0 + kappa * abs(Q) ** 2
            if den <= 0.0:
                state["rc"] = 5
                return None
            scl = 1.0 / np.sqrt(den)
            q_out[0] = Q / eps_t if modal else np.arctan(abs(Q)) * np.exp(1j * np.angle(Q)) / eps_t
            if not want_inverse:
                return None
            return np.array([[scl, 0.0], [-scl * Q, 0.0], [0.0, scl * kappa * np.conj(Q)], [0.0, scl]], np.complex128)
        h = d // 2
        T2i_low = peel(T[:, h:], True, q_out[h:])                         # step 1, :107-116
        if state["rc"]:
            return None
        T2i = np.zeros((4, d + 1), np.complex128)
        T2i[:, h:] = T2i_low                                              # degree-d array, upper half zero
        T1 = _conv2x2(T2i, T)                                             # step 2, :120-127 (2d+1 coefficients)
        T1i = peel(T1[:, d:d + h + 1], True, q_out[:h])                   # step 3, :131-140
        if state["rc"] or not want_inverse:
            return None
        return _conv2x2(T1i, T2i_low)                                     # step 4, :144-156 (d+1 coefficients)

    peel(tm, False, q)
    return state["rc"], (q if state["rc"] == 0 else None)


def _c128(a):
    return np.ascontiguousarray(a, dtype=np.complex128)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, path=None):
        self.lib = C.CDLL(path or build_oracle())
        L = self.lib
        L.orc_next_fast_size.restype = C.c_size_t
        L.orc_next_fast_size.argtypes = [C.c_size_t]
        L.orc_fft.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_poly_eval.argtypes = [C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_poly_fmult2x2_numel.restype = C.c_size_t
        L.orc_poly_fmult2x2_numel.argtypes = [C.c_size_t, C.c_size_t]
        L.orc_poly_fmult2x2.argtypes = [C.POINTER(C.c_size_t), C.c_size_t, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_int32)]
        L.orc_poly_chirpz_p.argtypes = [C.c_size_t, C.c_void_p, C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.c_size_t, C.c_void_p]
        L.orc_akns_degree.restype = C.c_size_t
        L.orc_akns_degree.argtypes = [C.c_int]
        L.orc_akns_fscatter_numel.restype = C.c_size_t
        L.orc_akns_fscatter_numel.argtypes = [C.c_size_t, C.c_int]
        L.orc_akns_fscatter.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_int32), C.c_int]
        L.orc_akns_coeffs.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                      C.c_int]
        L.orc_nse_fscatter_numel.restype = C.c_size_t
        L.orc_nse_fscatter_numel.argtypes = [C.c_size_t, C.c_int]
        L.orc_nse_fscatter.argtypes = [C.c_size_t, C.c_void_p, C.c_double, C.c_int, C.c_void_p,
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_int32), C.c_int]
        L.orc_fnft_nsev.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_fnft_nsev_ex.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_fnft_kdvv.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                    C.c_int]
        L.orc_kdv_fscatter_numel.restype = C.c_size_t
        L.orc_kdv_fscatter_numel.argtypes = [C.c_size_t, C.c_int]
        L.orc_kdv_fscatter.argtypes = [C.c_size_t, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int]
        L.orc_nse_scatter_bound_states.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_l2norm2.restype = C.c_double
        L.orc_l2norm2.argtypes = [C.c_size_t, C.c_void_p, C.c_double, C.c_double]
        L.orc_nse_preprocess.argtypes = [C.c_size_t, C.c_void_p, C.c_double, C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int]
        L.orc_nse_upsampling.restype = C.c_size_t
        L.orc_nse_upsampling.argtypes = [C.c_int]
        L.orc_last_timings.argtypes = [C.c_double * 2]
        L.orc_last_timings.restype = None

    # -- FFT ---------------------------------------------------------------------------------
    def next_fast_size(self, n):
        return int(self.lib.orc_next_fast_size(n))

    def fft(self, x, sign=-1):
        x = _c128(x)
        y = np.empty_like(x)
        rc = self.lib.orc_fft(x.size, _ptr(x), _ptr(y), sign)
        if rc:
            raise RuntimeError("orc_fft rc=%d" % rc)
        return y

    # -- polynomials -------------------------------------------------------------------------
    def poly_eval(self, p, z):
        p = _c128(p)
        z = _c128(z).copy()
        rc = self.lib.orc_poly_eval(p.size - 1, _ptr(p), z.size, _ptr(z))
        if rc:
            raise RuntimeError("orc_poly_eval rc=%d" % rc)
        return z

    def poly_fmult2x2(self, deg, n, p, normalize=True):
        """p: 4*n*(deg+1) entries, entry-major.  Returns (deg_out, result[4, deg_out+1], W)."""
        numel = int(self.lib.orc_poly_fmult2x2_numel(deg, n))
        buf = np.zeros(numel, np.complex128)
        p = _c128(p).ravel()
        assert p.size == 4 * n * (deg + 1)
        buf[: p.size] = p
        res = np.zeros(numel, np.complex128)
        d = C.c_size_t(deg)
        W = C.c_int32(0)
        rc = self.lib.orc_poly_fmult2x2(C.byref(d), n, _ptr(buf), _ptr(res),
                                        C.byref(W) if normalize else None)
        if rc:
            raise RuntimeError("orc_poly_fmult2x2 rc=%d" % rc)
        dd = d.value
        return dd, res[: 4 * (dd + 1)].reshape(4, dd + 1).copy(), int(W.value)

    def poly_chirpz(self, p, A, W, M):
        p = _c128(p)
        out = np.empty(M, np.complex128)
        A = complex(A)
        W = complex(W)
        rc = self.lib.orc_poly_chirpz_p(p.size - 1, _ptr(p), (C.c_double * 2)(A.real, A.imag),
                                        (C.c_double * 2)(W.real, W.imag), M, _ptr(out))
        if rc:
            raise RuntimeError("orc_poly_chirpz rc=%d" % rc)
        return out

    # -- scattering --------------------------------------------------------------------------
    def akns_coeffs(self, q, r, eps_t, disc):
        q = _c128(q)
        r = _c128(r)
        a = AKNS_DISC[disc] if isinstance(disc, str) else int(disc)
        deg = int(self.lib.orc_akns_degree(a))
        p = np.zeros(4 * q.size * (deg + 1), np.complex128)
        rc = self.lib.orc_akns_coeffs(q.size, _ptr(q), _ptr(r), eps_t, _ptr(p), a)
        return rc, deg, p

    def akns_fscatter(self, q, r, eps_t, disc, normalize=True):
        q = _c128(q)
        r = _c128(r)
        a = AKNS_DISC[disc] if isinstance(disc, str) else int(disc)
        numel = int(self.lib.orc_akns_fscatter_numel(q.size, a))
        res = np.zeros(max(numel, 1), np.complex128)
        d = C.c_size_t(0)
        W = C.c_int32(0)
        rc = self.lib.orc_akns_fscatter(q.size, _ptr(q), _ptr(r), eps_t, _ptr(res), C.byref(d),
                                        C.byref(W) if normalize else None, a)
        dd = d.value
        return rc, dd, res[: 4 * (dd + 1)].reshape(4, dd + 1).copy(), int(W.value)

    def nse_fscatter(self, q, eps_t, kappa, disc, normalize=True):
        q = _c128(q)
        n = NSE_DISC[disc] if isinstance(disc, str) else int(disc)
        numel = int(self.lib.orc_nse_fscatter_numel(q.size, n))
        res = np.zeros(max(numel, 1), np.complex128)
        d = C.c_size_t(0)
        W = C.c_int32(0)
        rc = self.lib.orc_nse_fscatter(q.size, _ptr(q), eps_t, kappa, _ptr(res), C.byref(d),
                                       C.byref(W) if normalize else None, n)
        dd = d.value
        return rc, dd, res[: 4 * (dd + 1)].reshape(4, dd + 1).copy(), int(W.value)

    def fnft_nsev(self, q, T, M, XI, kappa=1, disc="2SPLIT4B", cstype="BOTH", normalize=True,
                  richardson=False):
        """Continuous spectrum only.  Returns (rc, contspec) with contspec of length M*{1,2,3}."""
        q = _c128(q)
        T = np.ascontiguousarray(T, np.float64)
        XI = np.ascontiguousarray(XI, np.float64)
        n = NSE_DISC[disc] if isinstance(disc, str) else int(disc)
        c = CSTYPE[cstype] if isinstance(cstype, str) else int(cstype)
        out = np.zeros(M * {0: 1, 1: 2, 2: 3}[c], np.complex128)
        rc = self.lib.orc_fnft_nsev_ex(q.size, _ptr(q), _ptr(T), M, _ptr(out), _ptr(XI), kappa, n, c,
                                       1 if normalize else 0, 1 if richardson else 0)
        return rc, out

    def fnft_kdvv(self, u, T, M, XI, disc="2SPLIT8B"):
        """fnft_kdvv, reflection coefficient on M points.  Returns (rc, contspec)."""
        u = _c128(u)
        T = np.ascontiguousarray(T, np.float64)
        XI = np.ascontiguousarray(XI, np.float64)
        k = KDV_DISC[disc] if isinstance(disc, str) else int(disc)
        out = np.zeros(M, np.complex128)
        rc = self.lib.orc_fnft_kdvv(u.size, _ptr(u), _ptr(T), M, _ptr(out), _ptr(XI), k)
        return rc, out

    def kdv_fscatter(self, u, eps_t, disc, normalize=False):
        u = _c128(u)
        k = KDV_DISC[disc] if isinstance(disc, str) else int(disc)
        numel = int(self.lib.orc_kdv_fscatter_numel(u.size, k))
        res = np.zeros(max(numel, 1), np.complex128)
        d = C.c_size_t(0)
        W = C.c_int32(0)
        rc = self.lib.orc_kdv_fscatter(u.size, _ptr(u), eps_t, _ptr(res), C.byref(d),
                                       C.byref(W) if normalize else None, k)
        dd = d.value
        return rc, dd, res[: 4 * (dd + 1)].reshape(4, dd + 1).copy(), int(W.value)

    # ---- discrete spectrum (kappa = +1), fnft_nsev.c:276-309, 567-741, 895-1038 -----------------
    def preprocess(self, q, eps_t, Dsub, disc):
        """fnft__nse_discretization_preprocess_signal -> (rc, q_pre, Dsub, first_last)."""
        q = _c128(q)
        n = NSE_DISC[disc] if isinstance(disc, str) else int(disc)
        ds = C.c_size_t(Dsub)
        out = C.c_void_p()
        fl = (C.c_size_t * 2)()
        rc = self.lib.orc_nse_preprocess(q.size, _ptr(q), eps_t, C.byref(ds), C.byref(out), fl, n)
        if rc != 0:
            return rc, None, 0, (0, 0)
        ups = int(self.lib.orc_nse_upsampling(n))
        cnt = ds.value * ups
        buf = (C.c_double * (2 * cnt)).from_address(out.value)
        arr = np.frombuffer(buf, dtype=np.complex128, count=cnt).copy()
        C.CDLL(None).free(out)
        return 0, arr, ds.value, (fl[0], fl[1])

    def misc_resample(self, q, eps_t, delta):
        """fnft__misc_resample (fnft__misc.c:326-407) -> (rc, q_new)."""
        q = _c128(q)
        out = np.zeros(q.size, np.complex128)
        self.lib.orc_misc_resample.argtypes = [C.c_size_t, C.c_double, C.c_void_p, C.c_double, C.c_void_p]
        rc = self.lib.orc_misc_resample(q.size, float(eps_t), _ptr(q), float(delta), _ptr(out))
        return int(rc), out

    def scatter_bound_states(self, q_pre, T, lam, ups, skip_b=False):
        q_pre = _c128(q_pre)
        lam = _c128(lam)
        T = np.ascontiguousarray(T, np.float64)
        K = lam.size
        a, ap, b = (np.zeros(max(K, 1), np.complex128) for _ in range(3))
        rc = self.lib.orc_nse_scatter_bound_states(q_pre.size, _ptr(q_pre), _ptr(T), K, _ptr(lam), _ptr(a),
                                                   _ptr(ap), _ptr(b), ups, 1 if skip_b else 0)
        return rc, a[:K], ap[:K], b[:K]

    def _bounding_box(self, q_pre, T, eps_t, deg, ups, bsfilt):
        inf = float("inf")
        if bsfilt == "BASIC":
            return [-inf, inf, 0.0, inf]
        if bsfilt == "FULL":
            re_b = 0.9 * np.pi / abs(2.0 / deg * eps_t)                  # fnft_nsev.c:569-579
            qt = q_pre if ups == 1 else np.ascontiguousarray(ups * q_pre[1::ups])  # :626-640
            im_b = 1.5 * 0.25 * self.lib.orc_l2norm2(qt.size, _ptr(_c128(qt)), T[0], T[1])
            return [-re_b, re_b, 0.0, im_b]
        return [-inf, inf, -inf, inf]

    @staticmethod
    def _filter_merge(vals, box):
        keep = [v for v in vals if (v.real >= box[0]) and (v.real <= box[1]) and (v.imag >= box[2])
                and (v.imag <= box[3])]                                   # fnft__misc.c:114-157
        vals = list(keep)
        n = len(vals)
        if n == 0:
            return np.zeros(0, np.complex128)
        tol = np.sqrt(2.220446049250313e-16)
        nf = 1
        for i in range(1, n):                                             # fnft__misc.c:228-259, in place
            dist = -1.0
            for j in range(i):
                dist = abs(vals[j] - vals[i])
                if dist < tol:
                    break
            if dist < tol:
                continue
            vals[nf] = vals[i]
            nf += 1
        return np.array(vals[:nf], np.complex128)

    def _base_ds(self, q_pre, T, kappa, disc, bsloc, bsfilt, niter, dstype, guesses):
        """fnft_nsev_base, discrete part.  Returns (rc, bound_states, normconsts, residues, aprimes)."""
        n = NSE_DISC[disc] if isinstance(disc, str) else int(disc)
        ups = int(self.lib.orc_nse_upsampling(n))
        deg = int(self.lib.orc_akns_degree(self.lib.orc_nse_to_akns(n)))
        Dg = q_pre.size // ups
        eps_t = (T[1] - T[0]) / (Dg - 1)
        box = self._bounding_box(q_pre, T, eps_t, deg, ups, bsfilt)
        if bsloc == "NEWTON":
            bs = np.array(guesses, np.complex128).copy()
            eprec = 100 * 2.220446049250313e-16
            for i in range(bs.size):                                      # fnft_nsev.c:971-1038
                it = 0
                while True:
                    rc, a, ap, _ = self.scatter_bound_states(q_pre, T, bs[i:i + 1], ups, skip_b=True)
                    if rc != 0:
                        return -abs(rc), None, None, None, None
                    if a[0] == 0.0:
                        break
                    if ap[0] == 0.0:
                        return 3, None, None, None, None
                    err = a[0] / ap[0]
                    bs[i] -= err
                    it += 1
                    if bs[i].imag > box[3] or bs[i].real > box[1] or bs[i].real < box[0] or bs[i].imag < box[2]:
                        break
                    if not (abs(err) > eprec and it < niter):
                        break
        elif bsloc == "FAST_EIGENVALUE":
            rc, d, tm, W = self.nse_fscatter_pre(q_pre, eps_t, kappa, n)
            if rc != 0:
                return -abs(rc), None, None, None, None
            with np.errstate(all="ignore"):
                z = poly_roots(tm[0])                                     # fnft__poly_roots_fasteigen.c
                bs = np.log(z.astype(np.complex128)) / (2j * eps_t / (deg * ups))
        else:
            return 2, None, None, None, None
        if bsfilt != "NONE":
            bs = self._filter_merge(list(bs), box)
        K = bs.size
        if K == 0:
            e = np.zeros(0, np.complex128)
            return 0, e, e, e, e
        rc, a, ap, b = self.scatter_bound_states(q_pre, T, bs, ups, skip_b=False)   # :895-968
        if rc != 0:
            return -abs(rc), None, None, None, None
        return 0, bs, b, b / ap, ap

    def nse_fscatter_pre(self, q_pre, eps_t, kappa, n):
        numel = int(self.lib.orc_nse_fscatter_numel(q_pre.size, n))
        res = np.zeros(max(numel, 1), np.complex128)
        d = C.c_size_t(0)
        W = C.c_int32(0)
        rc = self.lib.orc_nse_fscatter(q_pre.size, _ptr(_c128(q_pre)), eps_t, kappa, _ptr(res), C.byref(d),
                                       C.byref(W), n)
        dd = d.value
        return rc, dd, res[: 4 * (dd + 1)].reshape(4, dd + 1).copy(), int(W.value)

    def fnft_nsev_ds(self, q, T, disc="2SPLIT4B", bsloc="SUBSAMPLE_AND_REFINE", bsfilt="FULL", niter=10,
                     Dsub=0, guesses=None, richardson=False):
        """Discrete spectrum of fnft_nsev for kappa = +1.
        Returns (rc, bound_states, normconsts, residues)."""
        q = _c128(q)
        T = [float(T[0]), float(T[1])]
        D = q.size
        n = NSE_DISC[disc] if isinstance(disc, str) else int(disc)
        ups = int(self.lib.orc_nse_upsampling(n))
        eps_t = (T[1] - T[0]) / (D - 1)
        rc, q_pre, _, _ = self.preprocess(q, eps_t, D, n)
        if rc != 0:
            return -abs(rc), None, None, None
        if bsloc == "SUBSAMPLE_AND_REFINE":                               # fnft_nsev.c:276-304
            ds = Dsub if Dsub else int(np.sqrt(D * np.log2(D) * np.log2(D)))
            nskip = int(np.floor(D / ds + 0.5))
            ds = int(np.floor(D / nskip + 0.5))
            rc, qsub, ds, fl = self.preprocess(q, eps_t, ds, n)
            if rc != 0:
                return -abs(rc), None, None, None
            Tsub = [T[0] + fl[0] * eps_t, T[0] + fl[1] * eps_t]
            rc, bs0, _, _, _ = self._base_ds(qsub, Tsub, 1, n, "FAST_EIGENVALUE", bsfilt, niter, "BOTH", None)
            if rc != 0:
                return rc, None, None, None
            rc, bs, nc, res, ap = self._base_ds(q_pre, T, 1, n, "NEWTON", bsfilt, niter, "BOTH", bs0)
        else:
            rc, bs, nc, res, ap = self._base_ds(q_pre, T, 1, n, bsloc, bsfilt, niter, "BOTH", guesses)
        if rc != 0 or not richardson or bs.size == 0:
            return rc, bs, nc, res
        # Richardson extrapolation of the discrete spectrum, fnft_nsev.c:340-364, 376-392, 406-441
        ds = D // 2
        rc, qsub, ds, fl = self.preprocess(q, eps_t, ds, n)
        if rc != 0:
            return -abs(rc), None, None, None
        Tsub = [T[0] + fl[0] * eps_t, T[0] + fl[1] * eps_t]
        eps_sub = (Tsub[1] - Tsub[0]) / (ds - 1)
        rc, bs_s, nc_s, res_s, ap_s = self._base_ds(qsub, Tsub, 1, n, "NEWTON", bsfilt, niter, "BOTH", bs.copy())
        if rc != 0:
            return rc, None, None, None
        order = 4.0 if ups == 2 else 2.0
        sn = (eps_sub / eps_t) ** order
        sd = sn - 1.0
        bs, res = bs.copy(), res.copy()
        for i in range(bs.size):
            loc, thr = bs_s.size, eps_t
            for j in range(bs_s.size):
                e = abs(bs[i] - bs_s[j]) / abs(bs[i])
                if e < thr:
                    thr, loc = e, j
            if loc < bs_s.size:
                bs[i] = (sn * bs[i] - bs_s[loc]) / sd
                ap_i = (sn * (nc[i] / res[i]) - (nc_s[loc] / res_s[loc])) / sd
                res[i] = nc[i] / ap_i
        return 0, bs, nc, res

    def last_timings(self):
        t = (C.c_double * 2)()
        self.lib.orc_last_timings(t)
        return float(t[0]), float(t[1])


_ORACLE = None


def load_oracle():
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle()
    return _ORACLE
