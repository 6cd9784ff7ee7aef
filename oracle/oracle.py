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


def build_oracle(force=False):
    """Compile oracle/liboracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "fnft_oracle.c")
    if (not force and os.path.exists(_LIB)
            and os.path.getmtime(_LIB) >= max(os.path.getmtime(src),
                                              os.path.getmtime(os.path.join(_HERE, "fnft_oracle.h")))):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


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
