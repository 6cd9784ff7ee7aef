"""ctypes bindings of libfnft_amd.so -- the same C ABI a C / MATLAB-mex / FNFTpy caller of the
reference's libfnft.so uses (include/fnft_amd.h).  Names, argument meaning and return codes
follow the reference (include/fnft_nsev.h:371-376, include/private/fnft__poly_fmult.h:223-224,
include/private/fnft__poly_chirpz.h:61, include/private/fnft__akns_fscatter.h:89-90,
include/private/fnft__nse_fscatter.h:81-84).

There is no CPU fallback: if the HIP library has not been built, load() raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfnft_amd.so")

# return codes, include/fnft_errwarn.h:44-94
FNFT_SUCCESS = 0
FNFT_EC_NOMEM = 1
FNFT_EC_INVALID_ARGUMENT = 2
FNFT_EC_DIV_BY_ZERO = 3
FNFT_EC_OTHER = 5
FNFT_EC_NOT_YET_IMPLEMENTED = 6

# fnft_nse_discretization_t, include/fnft_nse_discretization_t.h:104-133
NSE_DISC = {
    "2SPLIT2_MODAL": 0, "BO": 1, "2SPLIT1A": 2, "2SPLIT1B": 3, "2SPLIT2A": 4, "2SPLIT2B": 5,
    "2SPLIT2S": 6, "2SPLIT3A": 7, "2SPLIT3B": 8, "2SPLIT3S": 9, "2SPLIT4A": 10, "2SPLIT4B": 11,
    "2SPLIT5A": 12, "2SPLIT5B": 13, "2SPLIT6A": 14, "2SPLIT6B": 15, "2SPLIT7A": 16,
    "2SPLIT7B": 17, "2SPLIT8A": 18, "2SPLIT8B": 19, "4SPLIT4A": 20, "4SPLIT4B": 21,
    "CF4_2": 22, "CF4_3": 23, "CF5_3": 24, "CF6_4": 25, "ES4": 26, "TES4": 27,
}
# fnft__akns_discretization_t, include/private/fnft__akns_discretization_t.h:104-134
AKNS_DISC = {
    "2SPLIT2_MODAL": 0, "2SPLIT1A": 1, "2SPLIT1B": 2, "2SPLIT2A": 3, "2SPLIT2B": 4, "2SPLIT2S": 5,
    "2SPLIT3A": 6, "2SPLIT3B": 7, "2SPLIT3S": 8, "2SPLIT4A": 9, "2SPLIT4B": 10, "2SPLIT5A": 11,
    "2SPLIT5B": 12, "2SPLIT6A": 13, "2SPLIT6B": 14, "2SPLIT7A": 15, "2SPLIT7B": 16,
    "2SPLIT8A": 17, "2SPLIT8B": 18, "BO": 19, "4SPLIT4A": 20, "4SPLIT4B": 21,
}
KDV_DISC = {n: i for i, n in enumerate(
    ["2SPLIT1A", "2SPLIT1B", "2SPLIT2A", "2SPLIT2B", "2SPLIT2S", "2SPLIT3A", "2SPLIT3B", "2SPLIT3S",
     "2SPLIT4A", "2SPLIT4B", "2SPLIT5A", "2SPLIT5B", "2SPLIT6A", "2SPLIT6B", "2SPLIT7A", "2SPLIT7B",
     "2SPLIT8A", "2SPLIT8B", "4SPLIT4A", "4SPLIT4B", "BO", "CF4_2", "CF4_3", "CF5_3", "CF6_4"])}
CSTYPE = {"RHO": 0, "REFLECTION_COEFFICIENT": 0, "AB": 1, "BOTH": 2}
CS_FACTOR = {0: 1, 1: 2, 2: 3}


class KdvvOpts(C.Structure):
    """fnft_kdvv_opts_t, include/fnft_kdvv.h:60-62."""
    _fields_ = [("discretization", C.c_int)]


class NsevOpts(C.Structure):
    """fnft_nsev_opts_t, include/fnft_nsev.h:198-208 (same field order = same ABI)."""
    _fields_ = [
        ("bound_state_filtering", C.c_int),
        ("bound_state_localization", C.c_int),
        ("niter", C.c_size_t),
        ("Dsub", C.c_size_t),
        ("discspec_type", C.c_int),
        ("contspec_type", C.c_int),
        ("normalization_flag", C.c_int32),
        ("discretization", C.c_int),
        ("richardson_extrapolation_flag", C.c_size_t),
    ]


class NsevInverseOpts(C.Structure):
    """fnft_nsev_inverse_opts_t, include/fnft_nsev_inverse.h:155-162 (same field order = same ABI)."""
    _fields_ = [
        ("discretization", C.c_int),
        ("contspec_type", C.c_int),
        ("contspec_inversion_method", C.c_int),
        ("discspec_type", C.c_int),
        ("max_iter", C.c_size_t),
        ("oversampling_factor", C.c_size_t),
    ]


INV_CSTYPE = {"REFLECTION_COEFFICIENT": 0, "B_OF_XI": 1, "B_OF_TAU": 2}
INV_DSTYPE = {"NORMING_CONSTANTS": 0, "RESIDUES": 1}
INV_CSMETHOD = {"DEFAULT": 0, "TFMATRIX_CONTAINS_REFL_COEFF": 1, "TFMATRIX_CONTAINS_AB_FROM_ITER": 2,
                "USE_SEED_POTENTIAL_INSTEAD": 3}

PRINTF_T = C.CFUNCTYPE(C.c_int32, C.c_char_p)  # variadic in C; used only to silence output

EXPORTED = [
    "fnft_nsev", "fnft_nsev_default_opts", "fnft_nsev_max_K", "fnft_errwarn_setprintf",
    "fnft_errwarn_getprintf", "fnft__poly_fmult2x2_numel", "fnft__poly_fmult2x2",
    "fnft_amd_poly_chirpz", "fnft__poly_chirpz", "fnft__akns_fscatter_numel", "fnft__akns_fscatter",
    "fnft__nse_fscatter_numel", "fnft__nse_fscatter", "fnft_amd_device_count",
    "fnft_amd_last_error", "fnft_amd_plan_create", "fnft_amd_plan_create_sub", "fnft_amd_plan_destroy",
    "fnft_amd_plan_workspace_bytes", "fnft_amd_nsev_contspec_device", "fnft_amd_plan_finish",
    "fnft_amd_plan_last_ms", "fnft_amd_plan_set_timing", "fnft_amd_plan_set_launch_timing",
    "fnft_amd_plan_launch_count", "fnft_amd_plan_launch_ms", "fnft_amd_plan_get_transfer_matrix",
    "fnft_amd_plan_get_transfer_matrix_device", "fnft_amd_plan_device", "fnft_amd_current_device",
    "fnft_amd_nsev_contspec_from_tm_device", "fnft__misc_resample", "fnft__poly_roots_fasteigen",
    "fnft__nse_scatter_bound_states", "fnft__poly_fmult_numel", "fnft__poly_fmult", "fnft__poly_fmult_two_polys_len",
    "fnft__poly_fmult_two_polys", "fnft__poly_fmult_two_polys2x2", "fnft__nse_finvscatter",
    "fnft_amd_plan_last_warnings", "fnft_amd_discspec_stage_ms",
    "fnft_nsev_inverse", "fnft_nsev_inverse_default_opts", "fnft_nsev_inverse_XI", "fnft__poly_specfact",
    "fnft__nse_scatter_matrix", "fnft__poly_roots_fftgridsearch", "fnft__poly_roots_fftgridsearch_paraherm",
    "fnft_kdvv", "fnft_kdvv_default_opts", "fnft__kdv_fscatter_numel", "fnft__kdv_fscatter",
    "fnft_amd_kdvv_plan_create", "fnft_amd_kdvv_contspec_device", "fnft_amd_kdvv_plan_set_real_mode",
    "fnft_amd_release_cached", "fnft_nsep", "fnft_nsep_default_opts", "fnft_amd_poly_fmult2x2_device",
]

_lib = None


def load(path=None):
    """Load libfnft_amd.so.  Raises (never falls back) when the library is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # PyTorch wheels bundle their own copy of the HIP/HSA runtime.  Two copies in one process
    # cannot both own the GPU, so when torch is installed it is imported FIRST: the loader then
    # resolves this library's libamdhip64.so.7 to the copy torch already mapped.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(p):
        raise RuntimeError(
            "libfnft_amd.so not found at %s -- run `python -m fnft_amd.build` (hipcc, gfx950); "
            "this package has no CPU fallback" % p)
    L = C.CDLL(p)
    vp, sz, i32, dbl = C.c_void_p, C.c_size_t, C.c_int32, C.c_double
    L.fnft_nsev_default_opts.restype = NsevOpts
    L.fnft_nsev_default_opts.argtypes = []
    L.fnft_nsev_max_K.restype = sz
    L.fnft_nsev_max_K.argtypes = [sz, C.POINTER(NsevOpts)]
    L.fnft_nsev.restype = i32
    L.fnft_nsev.argtypes = [sz, vp, vp, sz, vp, vp, C.POINTER(sz), vp, vp, i32, C.POINTER(NsevOpts)]
    L.fnft_errwarn_setprintf.restype = None
    L.fnft_errwarn_setprintf.argtypes = [vp]
    L.fnft_errwarn_getprintf.restype = vp
    L.fnft__poly_fmult2x2_numel.restype = sz
    L.fnft__poly_fmult2x2_numel.argtypes = [sz, sz]
    L.fnft__poly_fmult2x2.restype = i32
    L.fnft__poly_fmult2x2.argtypes = [C.POINTER(sz), sz, vp, vp, C.POINTER(i32)]
    L.fnft_amd_poly_chirpz.restype = i32
    L.fnft_amd_poly_chirpz.argtypes = [sz, vp, C.POINTER(dbl), C.POINTER(dbl), sz, vp]
    L.fnft__akns_fscatter_numel.restype = sz
    L.fnft__akns_fscatter_numel.argtypes = [sz, C.c_int]
    L.fnft__akns_fscatter.restype = i32
    L.fnft__akns_fscatter.argtypes = [sz, vp, vp, dbl, vp, C.POINTER(sz), C.POINTER(i32), C.c_int]
    L.fnft__nse_fscatter_numel.restype = sz
    L.fnft__nse_fscatter_numel.argtypes = [sz, C.c_int]
    L.fnft__nse_fscatter.restype = i32
    L.fnft__nse_fscatter.argtypes = [sz, vp, dbl, i32, vp, C.POINTER(sz), C.POINTER(i32), C.c_int]
    L.fnft_amd_device_count.restype = C.c_int
    L.fnft_amd_last_error.restype = C.c_char_p
    L.fnft_amd_plan_create.restype = i32
    L.fnft_amd_plan_create.argtypes = [C.POINTER(vp), sz, sz, sz, C.c_int, C.c_int]
    L.fnft_kdvv_default_opts.restype = KdvvOpts
    L.fnft_kdvv_default_opts.argtypes = []
    L.fnft_kdvv.restype = i32
    L.fnft_kdvv.argtypes = [sz, vp, vp, sz, vp, vp, vp, vp, vp, C.POINTER(KdvvOpts)]
    L.fnft__kdv_fscatter_numel.restype = sz
    L.fnft__kdv_fscatter_numel.argtypes = [sz, C.c_int]
    L.fnft__kdv_fscatter.restype = i32
    L.fnft__kdv_fscatter.argtypes = [sz, vp, dbl, vp, C.POINTER(sz), C.POINTER(i32), C.c_int]
    L.fnft_amd_kdvv_plan_create.restype = i32
    L.fnft_amd_kdvv_plan_create.argtypes = [C.POINTER(vp), sz, sz, sz, C.c_int, C.c_int]
    L.fnft_amd_release_cached.restype = None
    L.fnft_amd_release_cached.argtypes = [C.c_int]
    L.fnft_amd_kdvv_plan_set_real_mode.restype = i32
    L.fnft_amd_kdvv_plan_set_real_mode.argtypes = [vp, C.c_int]
    L.fnft_amd_kdvv_contspec_device.restype = i32
    L.fnft_amd_kdvv_contspec_device.argtypes = [vp, vp, vp, C.POINTER(dbl), C.POINTER(dbl), vp]
    L.fnft_amd_plan_create_sub.restype = i32
    L.fnft_amd_plan_create_sub.argtypes = [C.POINTER(vp), sz, sz, sz, C.c_int, C.c_int, sz]
    L.fnft_amd_plan_destroy.restype = None
    L.fnft_amd_plan_destroy.argtypes = [vp]
    L.fnft_amd_plan_workspace_bytes.restype = sz
    L.fnft_amd_plan_workspace_bytes.argtypes = [vp]
    L.fnft_amd_nsev_contspec_device.restype = i32
    L.fnft_amd_nsev_contspec_device.argtypes = [vp, vp, vp, C.POINTER(dbl), C.POINTER(dbl), i32,
                                                C.c_int, i32, vp]
    L.fnft_amd_plan_finish.restype = i32
    L.fnft_amd_plan_finish.argtypes = [vp, vp]
    L.fnft_amd_plan_last_ms.restype = dbl
    L.fnft_amd_plan_last_ms.argtypes = [vp, C.c_int]
    L.fnft_amd_plan_set_timing.restype = None
    L.fnft_amd_plan_set_timing.argtypes = [vp, C.c_int]
    L.fnft_amd_plan_set_launch_timing.restype = None
    L.fnft_amd_plan_set_launch_timing.argtypes = [vp, C.c_int]
    L.fnft_amd_plan_launch_count.restype = C.c_size_t
    L.fnft_amd_plan_launch_count.argtypes = [vp]
    L.fnft_amd_plan_launch_ms.restype = dbl
    L.fnft_amd_plan_launch_ms.argtypes = [vp, C.c_size_t, C.c_char_p, C.c_size_t]
    L.fnft_amd_plan_get_transfer_matrix.restype = i32
    L.fnft_amd_plan_get_transfer_matrix.argtypes = [vp, sz, vp, C.POINTER(sz), C.POINTER(i32)]
    L.fnft_amd_plan_get_transfer_matrix_device.restype = i32
    L.fnft_amd_plan_get_transfer_matrix_device.argtypes = [vp, sz, vp, C.POINTER(sz), C.POINTER(i32), vp]
    L.fnft_amd_nsev_contspec_from_tm_device.restype = i32
    L.fnft_amd_nsev_contspec_from_tm_device.argtypes = [vp, vp, i32, vp, C.POINTER(dbl), C.POINTER(dbl), C.c_int, vp]
    L.fnft__misc_resample.restype = i32
    L.fnft__misc_resample.argtypes = [sz, dbl, vp, dbl, vp]
    L.fnft__poly_roots_fasteigen.restype = i32
    L.fnft__poly_roots_fasteigen.argtypes = [sz, vp, vp]
    L.fnft__nse_scatter_bound_states.restype = i32
    L.fnft__nse_scatter_bound_states.argtypes = [sz, vp, vp, vp, sz, vp, vp, vp, vp, C.c_int, sz]
    L.fnft__poly_fmult_numel.restype = sz
    L.fnft__poly_fmult_numel.argtypes = [sz, sz]
    L.fnft__poly_fmult.restype = i32
    L.fnft__poly_fmult.argtypes = [C.POINTER(sz), sz, vp, C.POINTER(i32)]
    L.fnft__poly_fmult_two_polys_len.restype = i32
    L.fnft__poly_fmult_two_polys_len.argtypes = [sz]
    L.fnft__poly_fmult_two_polys.restype = i32
    L.fnft__poly_fmult_two_polys.argtypes = [sz, vp, vp, vp, vp, vp, vp, vp, vp, sz]
    L.fnft__poly_fmult_two_polys2x2.restype = i32
    L.fnft__poly_fmult_two_polys2x2.argtypes = [sz, vp, sz, vp, sz, vp, sz, vp, vp, vp, vp, vp, sz]
    L.fnft_nsev_inverse_default_opts.restype = NsevInverseOpts
    L.fnft_nsev_inverse_XI.restype = i32
    L.fnft_nsev_inverse_XI.argtypes = [sz, vp, sz, vp, C.c_int]
    L.fnft_nsev_inverse.restype = i32
    L.fnft_nsev_inverse.argtypes = [sz, vp, vp, sz, vp, vp, sz, vp, vp, i32, vp]
    for fn in (L.fnft__poly_roots_fftgridsearch, L.fnft__poly_roots_fftgridsearch_paraherm):
        fn.restype = i32
        fn.argtypes = [sz, vp, vp, vp, vp]
    L.fnft__nse_scatter_matrix.restype = i32
    L.fnft__nse_scatter_matrix.argtypes = [sz, vp, vp, dbl, i32, sz, vp, vp, C.c_int, sz]
    L.fnft__poly_specfact.restype = i32
    L.fnft__poly_specfact.argtypes = [sz, vp, vp, sz, i32]
    L.fnft__nse_finvscatter.restype = i32
    L.fnft__nse_finvscatter.argtypes = [sz, vp, vp, dbl, i32, C.c_int]
    L.fnft_amd_discspec_stage_ms.restype = dbl
    L.fnft_amd_discspec_stage_ms.argtypes = [sz, C.c_char_p, sz]
    L.fnft_amd_plan_last_warnings.restype = C.c_int
    L.fnft_amd_plan_last_warnings.argtypes = [vp]
    L.fnft_amd_plan_device.restype = C.c_int
    L.fnft_amd_plan_device.argtypes = [vp]
    L.fnft_amd_current_device.restype = C.c_int
    if path is None:
        _lib = L
    return L


def _c128(a):
    return np.ascontiguousarray(a, dtype=np.complex128)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _d2(v):
    return (C.c_double * 2)(float(v[0]), float(v[1]))


def last_error():
    return load().fnft_amd_last_error().decode()


def silence_errors(lib=None):
    """fnft_errwarn_setprintf(NULL): the reference's way of disabling error text."""
    (lib or load()).fnft_errwarn_setprintf(None)


def default_kdvv_opts():
    return load().fnft_kdvv_default_opts()


def default_opts():
    return load().fnft_nsev_default_opts()


# --------------------------------------------------------------------------------------------
# host-pointer calls (drop-in boundary)
# --------------------------------------------------------------------------------------------
def fnft_nsev(q, T, M, XI, kappa=1, discretization="2SPLIT4B", contspec_type="REFLECTION_COEFFICIENT",
              normalization_flag=1, opts=None, want_contspec=True, bound_states=None, K=None,
              richardson=False, normconsts=None, out=None):
    """fnft_nsev() through the C ABI with host (numpy) buffers.  Returns (rc, contspec).
    bound_states / normconsts: caller-allocated complex128 arrays (K = capacity of bound_states);
    see fnft_nsev_ds() for the discrete spectrum with managed buffers."""
    L = load()
    q = _c128(q)
    if opts is None:
        opts = L.fnft_nsev_default_opts()
        opts.discretization = NSE_DISC[discretization] if isinstance(discretization, str) else int(discretization)
        opts.contspec_type = CSTYPE[contspec_type] if isinstance(contspec_type, str) else int(contspec_type)
        opts.normalization_flag = int(normalization_flag)
        opts.richardson_extrapolation_flag = 1 if richardson else 0
    Tn = None if T is None else np.ascontiguousarray(T, np.float64)
    XIn = None if XI is None else np.ascontiguousarray(XI, np.float64)
    fac = CS_FACTOR.get(int(opts.contspec_type), 3)
    # out: a caller-owned result array that is reused between calls (what a C caller has: the reference's caller
    # allocates contspec once); without it a fresh zeroed array is made per call (its pages are first touched by the copy)
    if out is not None:
        assert out.dtype == np.complex128 and out.flags.c_contiguous and out.size >= max(M * fac, 1)
    cs = (out if out is not None else np.zeros(max(M * fac, 1), np.complex128)) if want_contspec else None
    Kc = C.c_size_t(K if K is not None else 0)
    rc = L.fnft_nsev(q.size, _ptr(q), None if Tn is None else _ptr(Tn), M,
                     None if cs is None else _ptr(cs), None if XIn is None else _ptr(XIn),
                     C.byref(Kc) if K is not None else None,
                     None if bound_states is None else _ptr(bound_states),
                     None if normconsts is None else _ptr(normconsts), int(kappa), C.byref(opts))
    if K is not None:
        fnft_nsev.last_K = int(Kc.value)
    return int(rc), (cs[: M * fac] if cs is not None else None)


BSLOC = {"FAST_EIGENVALUE": 0, "NEWTON": 1, "SUBSAMPLE_AND_REFINE": 2}
BSFILT = {"NONE": 0, "BASIC": 1, "FULL": 2}
DSTYPE = {"NORMING_CONSTANTS": 0, "RESIDUES": 1, "BOTH": 2}


def fnft_nsev_ds(q, T, discretization="2SPLIT4B", bsloc="SUBSAMPLE_AND_REFINE", bsfilt="FULL", niter=10,
                 Dsub=0, dstype="BOTH", guesses=None, richardson=False, M=0, XI=None, K=None, bufs=None):
    """Discrete spectrum (kappa = +1) through the drop-in fnft_nsev().
    Returns (rc, bound_states, normconsts, residues[, contspec if M > 0])."""
    L = load()
    q = _c128(q)
    opts = L.fnft_nsev_default_opts()
    opts.discretization = NSE_DISC[discretization] if isinstance(discretization, str) else int(discretization)
    opts.bound_state_localization = BSLOC[bsloc] if isinstance(bsloc, str) else int(bsloc)
    opts.bound_state_filtering = BSFILT[bsfilt] if isinstance(bsfilt, str) else int(bsfilt)
    opts.niter = int(niter)
    opts.Dsub = int(Dsub)
    opts.discspec_type = DSTYPE[dstype] if isinstance(dstype, str) else int(dstype)
    opts.contspec_type = CSTYPE["BOTH"]
    opts.richardson_extrapolation_flag = 1 if richardson else 0
    cap = int(K) if K is not None else int(L.fnft_nsev_max_K(q.size, C.byref(opts)))
    # bufs: caller-owned arrays {"bs", "nc", "cs"} reused between calls (what a C caller of the reference has)
    bs = bufs["bs"] if bufs else np.zeros(max(cap, 1), np.complex128)
    if guesses is not None:
        g = _c128(guesses)
        bs[: g.size] = g
        cap = g.size
    nc = bufs["nc"] if bufs else np.zeros(2 * max(cap, 1), np.complex128)
    cs = (bufs["cs"] if bufs else np.zeros(3 * M, np.complex128)) if M > 0 else None
    Tn = np.ascontiguousarray(T, np.float64)
    XIn = None if XI is None else np.ascontiguousarray(XI, np.float64)
    Kc = C.c_size_t(cap)
    rc = L.fnft_nsev(q.size, _ptr(q), _ptr(Tn), M, None if cs is None else _ptr(cs),
                     None if XIn is None else _ptr(XIn), C.byref(Kc), _ptr(bs), _ptr(nc), 1, C.byref(opts))
    k = int(Kc.value)
    d = int(opts.discspec_type)
    ncs = nc[:k] if d in (0, 2) else None
    res = nc[k:2 * k] if d == 2 else (nc[:k] if d == 1 else None)
    out = (int(rc), bs[:k].copy(), None if ncs is None else ncs.copy(), None if res is None else res.copy())
    return out + ((cs,) if M > 0 else ())


def discspec_stages():
    """[(stage, ms)] of this thread's last discrete-spectrum call (host wall clock)."""
    L = load()
    out, buf, i = [], C.create_string_buffer(64), 0
    while True:
        ms = float(L.fnft_amd_discspec_stage_ms(i, buf, len(buf)))
        if ms < 0:
            return out
        out.append((buf.value.decode(), ms))
        i += 1


def poly_fmult(deg, n, p, normalize=True):
    """fnft__poly_fmult: n scalar polynomials of degree deg (p[n*(deg+1)]) -> (rc, deg_out, product[deg_out+1], W)."""
    L = load()
    buf = np.zeros(max(int(L.fnft__poly_fmult_numel(deg, n)), 1), np.complex128)
    p = _c128(p).ravel()
    buf[: p.size] = p
    d = C.c_size_t(deg)
    W = C.c_int32(0)
    rc = L.fnft__poly_fmult(C.byref(d), n, _ptr(buf), C.byref(W) if normalize else None)
    return int(rc), int(d.value), buf[: d.value + 1].copy(), int(W.value)


def poly_fmult_two_polys(p1, p2, result=None, mode=0, bufs=None):
    """fnft__poly_fmult_two_polys: (rc, result[len]); p1 / p2 None = the previous call's factor (kept in bufs);
    result / bufs are the caller's persistent buffers (allocated on first use)."""
    L = load()
    deg = (len(p1) if p1 is not None else len(p2)) - 1
    ln = int(L.fnft__poly_fmult_two_polys_len(deg))
    if bufs is None:
        bufs = [np.zeros(ln, np.complex128) for _ in range(3)]
    if result is None:
        result = np.zeros(ln, np.complex128)
    a = None if p1 is None else _c128(p1)
    b = None if p2 is None else _c128(p2)
    rc = L.fnft__poly_fmult_two_polys(deg, None if a is None else _ptr(a), None if b is None else _ptr(b), _ptr(result),
                                      None, None, _ptr(bufs[0]), _ptr(bufs[1]), _ptr(bufs[2]), mode)
    return int(rc), result, bufs


def poly_fmult_two_polys2x2(p1, p2):
    """fnft__poly_fmult_two_polys2x2: p1, p2 [4, deg+1] -> (rc, result[4, 2*deg+1])."""
    L = load()
    p1, p2 = _c128(p1), _c128(p2)
    deg = p1.shape[1] - 1
    res = np.zeros((4, 2 * deg + 1), np.complex128)
    rc = L.fnft__poly_fmult_two_polys2x2(deg, _ptr(p1), deg + 1, _ptr(p2), deg + 1, _ptr(res), 2 * deg + 1, None, None,
                                         None, None, None, 0)
    return int(rc), res


def nse_finvscatter(tm, eps_t, kappa, discretization):
    """fnft__nse_finvscatter: tm [4, deg+1] -> (rc, q[deg])."""
    L = load()
    tm = _c128(tm)
    deg = tm.shape[1] - 1
    q = np.zeros(max(deg, 1), np.complex128)
    d = NSE_DISC[discretization] if isinstance(discretization, str) else int(discretization)
    rc = L.fnft__nse_finvscatter(deg, _ptr(tm), _ptr(q), float(eps_t), int(kappa), d)
    return int(rc), q[:deg]


def nsev_inverse_XI(D, T, M, discretization="2SPLIT2A"):
    """fnft_nsev_inverse_XI: (rc, [XI0, XI1])."""
    L = load()
    Tn = np.ascontiguousarray(T, np.float64)
    XI = np.zeros(2, np.float64)
    d = NSE_DISC[discretization] if isinstance(discretization, str) else int(discretization)
    rc = L.fnft_nsev_inverse_XI(int(D), _ptr(Tn), int(M), _ptr(XI), d)
    return int(rc), [float(XI[0]), float(XI[1])]


def fnft_nsev_inverse(M, contspec, XI, bound_states, normconsts_or_residues, D, T, kappa, opts=None, q_seed=None):
    """fnft_nsev_inverse() through the C ABI with host (numpy) buffers: (rc, q[D]).  opts: dict with the names of the
    reference's option fields (strings for the enums).  contspec, if a complex128 array, is modified in place as the
    reference modifies its argument; q_seed: the seed potential of USE_SEED_POTENTIAL_INSTEAD."""
    L = load()
    o = L.fnft_nsev_inverse_default_opts()
    for k, v in (opts or {}).items():
        if k == "discretization":
            v = NSE_DISC[v] if isinstance(v, str) else int(v)
        elif k == "contspec_type":
            v = INV_CSTYPE[v] if isinstance(v, str) else int(v)
        elif k == "contspec_inversion_method":
            v = INV_CSMETHOD[v] if isinstance(v, str) else int(v)
        elif k == "discspec_type":
            v = INV_DSTYPE[v] if isinstance(v, str) else int(v)
        setattr(o, k, v)
    cs = None if contspec is None else (contspec if (isinstance(contspec, np.ndarray) and contspec.dtype == np.complex128
                                                     and contspec.flags.c_contiguous) else _c128(contspec))
    Tn = None if T is None else np.ascontiguousarray(T, np.float64)
    XIn = None if XI is None else np.ascontiguousarray(XI, np.float64)
    bs = None if bound_states is None else _c128(bound_states)
    nc = None if normconsts_or_residues is None else _c128(normconsts_or_residues)
    K = 0 if bs is None else bs.size
    q = np.zeros(max(int(D), 1), np.complex128) if q_seed is None else _c128(q_seed).copy()
    rc = L.fnft_nsev_inverse(int(M), None if cs is None else _ptr(cs), None if XIn is None else _ptr(XIn), K,
                             None if bs is None else _ptr(bs), None if nc is None else _ptr(nc), int(D), _ptr(q),
                             None if Tn is None else _ptr(Tn), int(kappa), C.byref(o))
    return int(rc), q[:int(D)]


def poly_roots_fftgridsearch(p, M, PHI, paraherm=False):
    """fnft__poly_roots_fftgridsearch(_paraherm): (rc, estimates of the roots on the arc PHI of the unit circle)."""
    L = load()
    p = _c128(p)
    Mc = C.c_size_t(int(M))
    ph = np.ascontiguousarray(PHI, np.float64)
    roots = np.zeros(max(int(M), 1), np.complex128)
    fn = L.fnft__poly_roots_fftgridsearch_paraherm if paraherm else L.fnft__poly_roots_fftgridsearch
    rc = fn(p.size - 1, _ptr(p), C.byref(Mc), _ptr(ph), _ptr(roots))
    return int(rc), roots[: int(Mc.value)].copy() if rc == 0 else roots[:0]


def nse_scatter_matrix(q, eps_t, kappa, lam, derivative=True, r=None, discretization="BO"):
    """fnft__nse_scatter_matrix: (rc, result [K, 8 or 4]) = [S11 S12 S21 S22 (S11' S12' S21' S22')] per lambda."""
    L = load()
    q = _c128(q)
    lam = _c128(lam)
    rr = None if r is None else _c128(r)
    w = 8 if derivative else 4
    out = np.zeros(max(lam.size, 1) * w, np.complex128)
    d = NSE_DISC[discretization] if isinstance(discretization, str) else int(discretization)
    rc = L.fnft__nse_scatter_matrix(q.size, _ptr(q), None if rr is None else _ptr(rr), float(eps_t), int(kappa),
                                    lam.size, _ptr(lam), _ptr(out), d, 1 if derivative else 0)
    return int(rc), out[: lam.size * w].reshape(lam.size, w)


def poly_specfact(poly, oversampling_factor, kappa):
    """fnft__poly_specfact: poly [deg+1] -> (rc, result[deg+1])."""
    L = load()
    poly = _c128(poly)
    out = np.zeros(poly.size, np.complex128)
    rc = L.fnft__poly_specfact(poly.size - 1, _ptr(poly), _ptr(out), int(oversampling_factor), int(kappa))
    return int(rc), out


def misc_resample(q, eps_t, delta):
    """fnft__misc_resample: (rc, q shifted by delta on the band-limited periodic grid)."""
    L = load()
    q = _c128(q)
    out = np.zeros(q.size, np.complex128)
    rc = L.fnft__misc_resample(q.size, float(eps_t), _ptr(q), float(delta), _ptr(out))
    return int(rc), out


def poly_roots_fasteigen(p):
    """fnft__poly_roots_fasteigen: p highest power first -> (rc, roots[deg])."""
    L = load()
    p = _c128(p)
    roots = np.zeros(max(p.size - 1, 1), np.complex128)
    rc = L.fnft__poly_roots_fasteigen(p.size - 1, _ptr(p), _ptr(roots))
    return int(rc), roots[: p.size - 1]


def nse_scatter_bound_states(q, T, lam, skip_b=False, discretization="BO"):
    """fnft__nse_scatter_bound_states: (rc, a, a', b) at the points lam."""
    L = load()
    q = _c128(q)
    lam = _c128(lam)
    Tn = np.ascontiguousarray(T, np.float64)
    K = lam.size
    a, ap, b = (np.zeros(max(K, 1), np.complex128) for _ in range(3))
    d = NSE_DISC[discretization] if isinstance(discretization, str) else int(discretization)
    rc = L.fnft__nse_scatter_bound_states(q.size, _ptr(q), None, _ptr(Tn), K, _ptr(lam), _ptr(a), _ptr(ap), _ptr(b), d,
                                          1 if skip_b else 0)
    return int(rc), a[:K], ap[:K], b[:K]


def poly_fmult2x2(deg, n, p, normalize=True):
    """fnft__poly_fmult2x2: p[4, n*(deg+1)] -> (rc, deg_out, result[4, deg_out+1], W)."""
    L = load()
    numel = int(L.fnft__poly_fmult2x2_numel(deg, n))
    buf = np.zeros(max(numel, 1), np.complex128)
    p = _c128(p).ravel()
    buf[: p.size] = p
    res = np.zeros(max(numel, 1), np.complex128)
    d = C.c_size_t(deg)
    W = C.c_int32(0)
    rc = L.fnft__poly_fmult2x2(C.byref(d), n, _ptr(buf), _ptr(res), C.byref(W) if normalize else None)
    dd = d.value
    return int(rc), dd, res[: 4 * (dd + 1)].reshape(4, dd + 1).copy(), int(W.value)


def poly_chirpz(p, A, W, M):
    L = load()
    p = _c128(p)
    out = np.zeros(M, np.complex128)
    A, W = complex(A), complex(W)
    rc = L.fnft_amd_poly_chirpz(p.size - 1, _ptr(p), _d2((A.real, A.imag)), _d2((W.real, W.imag)), M,
                                _ptr(out))
    return int(rc), out


def akns_fscatter(q, r, eps_t, discretization, normalize=True):
    L = load()
    q, r = _c128(q), _c128(r)
    a = AKNS_DISC[discretization] if isinstance(discretization, str) else int(discretization)
    numel = int(L.fnft__akns_fscatter_numel(q.size, a))
    res = np.zeros(max(numel, 1), np.complex128)
    d = C.c_size_t(0)
    W = C.c_int32(0)
    rc = L.fnft__akns_fscatter(q.size, _ptr(q), _ptr(r), eps_t, _ptr(res), C.byref(d),
                               C.byref(W) if normalize else None, a)
    dd = d.value
    return int(rc), dd, res[: 4 * (dd + 1)].reshape(4, dd + 1).copy(), int(W.value)


def nse_fscatter(q, eps_t, kappa, discretization, normalize=True):
    L = load()
    q = _c128(q)
    n = NSE_DISC[discretization] if isinstance(discretization, str) else int(discretization)
    numel = int(L.fnft__nse_fscatter_numel(q.size, n))
    res = np.zeros(max(numel, 1), np.complex128)
    d = C.c_size_t(0)
    W = C.c_int32(0)
    rc = L.fnft__nse_fscatter(q.size, _ptr(q), eps_t, int(kappa), _ptr(res), C.byref(d),
                              C.byref(W) if normalize else None, n)
    dd = d.value
    return int(rc), dd, res[: 4 * (dd + 1)].reshape(4, dd + 1).copy(), int(W.value)


def fnft_kdvv(u, T, M, XI, discretization="2SPLIT8B", opts=None, want_contspec=True, K=None,
              bound_states=None, normconsts=None):
    """fnft_kdvv() through the C ABI with host buffers.  Returns (rc, contspec)."""
    L = load()
    u = _c128(u)
    if opts is None:
        opts = L.fnft_kdvv_default_opts()
        opts.discretization = KDV_DISC[discretization] if isinstance(discretization, str) else int(discretization)
    Tn = None if T is None else np.ascontiguousarray(T, np.float64)
    XIn = None if XI is None else np.ascontiguousarray(XI, np.float64)
    cs = np.zeros(max(M, 1), np.complex128) if want_contspec else None
    Kc = C.c_size_t(K if K is not None else 0)
    rc = L.fnft_kdvv(u.size, _ptr(u), None if Tn is None else _ptr(Tn), M, None if cs is None else _ptr(cs),
                     None if XIn is None else _ptr(XIn), C.byref(Kc) if K is not None else None,
                     None if bound_states is None else _ptr(bound_states),
                     None if normconsts is None else _ptr(normconsts), C.byref(opts))
    return int(rc), (cs[:M] if cs is not None else None)


def kdv_fscatter(u, eps_t, discretization, normalize=True):
    L = load()
    u = _c128(u)
    k = KDV_DISC[discretization] if isinstance(discretization, str) else int(discretization)
    numel = int(L.fnft__kdv_fscatter_numel(u.size, k))
    res = np.zeros(max(numel, 1), np.complex128)
    d = C.c_size_t(0)
    W = C.c_int32(0)
    rc = L.fnft__kdv_fscatter(u.size, _ptr(u), eps_t, _ptr(res), C.byref(d),
                              C.byref(W) if normalize else None, k)
    dd = d.value
    return int(rc), dd, res[: 4 * (dd + 1)].reshape(4, dd + 1).copy(), int(W.value)


class KdvvPlan:
    """Device-resident KdV transform (fnft_amd_kdvv_plan_create)."""

    def __init__(self, D, M, batch=1, discretization="2SPLIT8B", device=0):
        self.L = load()
        self.D, self.M, self.batch = int(D), int(M), int(batch)
        self.disc = KDV_DISC[discretization] if isinstance(discretization, str) else int(discretization)
        self.h = C.c_void_p()
        rc = self.L.fnft_amd_kdvv_plan_create(C.byref(self.h), self.D, self.M, self.batch, self.disc, int(device))
        if rc != FNFT_SUCCESS:
            raise RuntimeError("fnft_amd_kdvv_plan_create rc=%d (%s)" % (rc, last_error()))

    def close(self):
        if self.h:
            self.L.fnft_amd_plan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_real_mode(self, mode):
        """-1: ask the device whether u is real (default); 1: u is real (real-coefficient tree); 0: complex tree."""
        return int(self.L.fnft_amd_kdvv_plan_set_real_mode(self.h, int(mode)))

    def set_timing(self, on=True):
        self.L.fnft_amd_plan_set_timing(self.h, 1 if on else 0)

    def set_launch_timing(self, on=True):
        self.L.fnft_amd_plan_set_launch_timing(self.h, 1 if on else 0)

    def launch_times(self):
        """[(kernel name, ms)] of every launch since set_launch_timing(True); the stream must be idle."""
        out = []
        buf = C.create_string_buffer(128)
        for i in range(int(self.L.fnft_amd_plan_launch_count(self.h))):
            ms = float(self.L.fnft_amd_plan_launch_ms(self.h, i, buf, len(buf)))
            out.append((buf.value.decode(), ms))
        return out

    def last_ms(self, which=2):
        return float(self.L.fnft_amd_plan_last_ms(self.h, which))

    def contspec_device(self, u_ptr, out_ptr, T, XI, stream=0):
        return int(self.L.fnft_amd_kdvv_contspec_device(self.h, C.c_void_p(u_ptr), C.c_void_p(out_ptr),
                                                        _d2(T), _d2(XI), C.c_void_p(stream)))

    def finish(self, stream=0):
        return int(self.L.fnft_amd_plan_finish(self.h, C.c_void_p(stream)))


# --------------------------------------------------------------------------------------------
# device-resident plan (inputs already in HBM)
# --------------------------------------------------------------------------------------------
class Plan:
    """fnft_amd_plan_t: `batch` signals of D samples, M spectral points, one discretization."""

    def __init__(self, D, M, batch=1, discretization="2SPLIT2_MODAL", device=0, nskip=1):
        self.L = load()
        self.D, self.M, self.batch = int(D), int(M), int(batch)
        self.disc = NSE_DISC[discretization] if isinstance(discretization, str) else int(discretization)
        self.h = C.c_void_p()
        rc = self.L.fnft_amd_plan_create_sub(C.byref(self.h), self.D, self.M, self.batch, self.disc,
                                             int(device), int(nskip))
        if rc != FNFT_SUCCESS:
            raise RuntimeError("fnft_amd_plan_create rc=%d (%s)" % (rc, last_error()))

    def close(self):
        if self.h:
            self.L.fnft_amd_plan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self):
        return int(self.L.fnft_amd_plan_workspace_bytes(self.h))

    def cs_len(self, contspec_type):
        c = CSTYPE[contspec_type] if isinstance(contspec_type, str) else int(contspec_type)
        return self.M * CS_FACTOR[c]

    def set_timing(self, on=True):
        self.L.fnft_amd_plan_set_timing(self.h, 1 if on else 0)

    def set_launch_timing(self, on=True):
        self.L.fnft_amd_plan_set_launch_timing(self.h, 1 if on else 0)

    def launch_times(self):
        """[(kernel name, ms)] of every launch since set_launch_timing(True); the stream must be idle."""
        out = []
        buf = C.create_string_buffer(128)
        for i in range(int(self.L.fnft_amd_plan_launch_count(self.h))):
            ms = float(self.L.fnft_amd_plan_launch_ms(self.h, i, buf, len(buf)))
            out.append((buf.value.decode(), ms))
        return out

    def last_ms(self, which=2):
        return float(self.L.fnft_amd_plan_last_ms(self.h, which))

    def contspec_device(self, q_ptr, out_ptr, T, XI, kappa=1, contspec_type="BOTH",
                        normalization_flag=1, stream=0):
        """Enqueue one pass.  q_ptr/out_ptr are raw device addresses (e.g. tensor.data_ptr())."""
        c = CSTYPE[contspec_type] if isinstance(contspec_type, str) else int(contspec_type)
        return int(self.L.fnft_amd_nsev_contspec_device(
            self.h, C.c_void_p(q_ptr), C.c_void_p(out_ptr), _d2(T), _d2(XI), int(kappa), c,
            int(normalization_flag), C.c_void_p(stream)))

    def finish(self, stream=0):
        return int(self.L.fnft_amd_plan_finish(self.h, C.c_void_p(stream)))

    def contspec_from_tm_device(self, tm_ptr, W, out_ptr, T, XI, contspec_type="BOTH", stream=0):
        """Continuous spectrum from a transfer matrix in device memory (fnft_amd_nsev_contspec_from_tm_device)."""
        c = CSTYPE[contspec_type] if isinstance(contspec_type, str) else int(contspec_type)
        return int(self.L.fnft_amd_nsev_contspec_from_tm_device(self.h, C.c_void_p(tm_ptr), int(W), C.c_void_p(out_ptr),
                                                                _d2(T), _d2(XI), c, C.c_void_p(stream)))

    def transfer_matrix_device(self, out_ptr, b=0, stream=0):
        """Transfer matrix of signal b into a device buffer of 4*(deg+1) complex128; returns (rc, deg, W)."""
        deg = C.c_size_t(0)
        W = C.c_int32(0)
        rc = self.L.fnft_amd_plan_get_transfer_matrix_device(self.h, b, C.c_void_p(out_ptr), C.byref(deg), C.byref(W),
                                                             C.c_void_p(stream))
        return int(rc), int(deg.value), int(W.value)

    @property
    def device(self):
        return int(self.L.fnft_amd_plan_device(self.h))

    def transfer_matrix(self, b=0):
        deg = C.c_size_t(0)
        W = C.c_int32(0)
        # deg is D*deg0; allocate for the largest supported deg0 = 4
        buf = np.zeros(4 * (4 * self.D + 1), np.complex128)
        rc = self.L.fnft_amd_plan_get_transfer_matrix(self.h, b, _ptr(buf), C.byref(deg), C.byref(W))
        d = deg.value
        return int(rc), d, buf[: 4 * (d + 1)].reshape(4, d + 1).copy(), int(W.value)


def poly_fmult2x2_device(deg, n, p_ptr, out_ptr, stream=0):
    """fnft_amd_poly_fmult2x2_device: n matrices of degree deg in device memory (reference input layout) -> their product
    in device memory (reference result layout, normalised).  Returns (rc, deg_out, W)."""
    L = load()
    L.fnft_amd_poly_fmult2x2_device.restype = C.c_int32
    L.fnft_amd_poly_fmult2x2_device.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t),
                                                C.POINTER(C.c_int32), C.c_void_p]
    d, W = C.c_size_t(0), C.c_int32(0)
    rc = L.fnft_amd_poly_fmult2x2_device(int(deg), int(n), C.c_void_p(p_ptr), C.c_void_p(out_ptr), C.byref(d), C.byref(W),
                                         C.c_void_p(stream))
    return int(rc), int(d.value), int(W.value)


def release_cached(device=-1):
    """fnft_amd_release_cached: give the host-pointer entry points' cached plans / work arrays back to the driver."""
    load().fnft_amd_release_cached(int(device))


# --------------------------------------------------------------------------------------------
# fnft_nsep (include/fnft_amd.h section 5)
# --------------------------------------------------------------------------------------------
NSEP_LOC = {"SUBSAMPLE_AND_REFINE": 0, "GRIDSEARCH": 1, "MIXED": 2}
NSEP_FILT = {"NONE": 0, "MANUAL": 1, "AUTO": 2}


class NsepOpts(C.Structure):
    """fnft_nsep_opts_t, include/fnft_nsep.h:140-151 (same field order)."""
    _fields_ = [("localization", C.c_int), ("filtering", C.c_int), ("bounding_box", C.c_double * 4),
                ("max_evals", C.c_size_t), ("discretization", C.c_int), ("normalization_flag", C.c_int32),
                ("floquet_range", C.c_double * 2), ("points_per_spine", C.c_size_t), ("Dsub", C.c_size_t),
                ("tol", C.c_double)]


def nsep_default_opts():
    L = load()
    L.fnft_nsep_default_opts.restype = NsepOpts
    L.fnft_nsep_default_opts.argtypes = []
    return L.fnft_nsep_default_opts()


def nsep_opts(d=None):
    """NsepOpts from a dict with the keys of oracle.nsep.default_opts() (missing keys: library defaults)."""
    o = nsep_default_opts()
    for k, v in (d or {}).items():
        if k == "localization":
            o.localization = NSEP_LOC[v] if isinstance(v, str) else int(v)
        elif k == "filtering":
            o.filtering = NSEP_FILT[v] if isinstance(v, str) else int(v)
        elif k == "discretization":
            o.discretization = NSE_DISC[v] if isinstance(v, str) else int(v)
        elif k == "bounding_box":
            for i in range(4):
                if v[i] is not None:
                    o.bounding_box[i] = float(v[i])
        elif k == "floquet_range":
            o.floquet_range[0], o.floquet_range[1] = float(v[0]), float(v[1])
        elif k == "tol":
            o.tol = float(v)
        else:
            setattr(o, k, int(v))
    return o


def fnft_nsep(q, T, phase_shift=0.0, kappa=1, opts=None, K=None, M=None, want_main=True, want_aux=True,
              sheet_indices=False):
    """fnft_nsep() through the C ABI with host buffers.  opts: dict or NsepOpts.  K / M: capacities of the result arrays
    (default 2*deg*D + 1 like the reference's harness).  Returns (rc, main_spec, aux_spec)."""
    L = load()
    L.fnft_nsep.restype = C.c_int32
    L.fnft_nsep.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.c_size_t), C.c_void_p,
                            C.POINTER(C.c_size_t), C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(NsepOpts)]
    q = _c128(q)
    o = opts if isinstance(opts, NsepOpts) else nsep_opts(opts)
    cap = 2 * 105 * q.size + 1
    Kc = C.c_size_t(cap if K is None else int(K))
    Mc = C.c_size_t(cap if M is None else int(M))
    ms = np.zeros(max(Kc.value, 1), np.complex128) if want_main else None
    au = np.zeros(max(Mc.value, 1), np.complex128) if want_aux else None
    Tn = None if T is None else np.ascontiguousarray(T, np.float64)
    sh = np.zeros(4) if sheet_indices else None
    rc = L.fnft_nsep(q.size, _ptr(q), None if Tn is None else _ptr(Tn), float(phase_shift), C.byref(Kc),
                     None if ms is None else _ptr(ms), C.byref(Mc), None if au is None else _ptr(au),
                     None if sh is None else _ptr(sh), int(kappa), C.byref(o) if opts is not None else None)
    return (int(rc), (ms[: Kc.value].copy() if (ms is not None and rc == 0) else None),
            (au[: Mc.value].copy() if (au is not None and rc == 0) else None))
