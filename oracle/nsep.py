"""oracle/nsep.py -- CPU restatement of fnft_nsep (TEST INFRASTRUCTURE ONLY; see oracle/fnft_oracle.h).

Follows the reference line by line (file:line relative to the FNFT source tree):
    fnft_nsep               src/fnft_nsep.c:82-220   (checks, phase-shift removal, localization switch)
    gridsearch              src/fnft_nsep.c:222-439  (Floquet polynomials, roots on the unit circle)
    subsample_and_refine    src/fnft_nsep.c:441-706  (all roots of the subsampled polynomials, Newton refinement)
    refine_mainspec         src/fnft_nsep.c:708-792
    refine_auxspec          src/fnft_nsep.c:794-836
    update_bounding_box...  src/fnft_nsep.c:838-
on the oracle's pieces: the C restatements of fnft__nse_discretization_preprocess_signal, fnft__nse_fscatter and
fnft__nse_scatter_matrix (oracle/fnft_oracle.c), numpy's LAPACK root finder in place of eiscor, and the numpy
restatement of fnft__poly_roots_fftgridsearch (oracle/inverse.py).

The refinement loops are written once per spectral point, exactly as the reference's sequential loops, but as
coroutines that YIELD the lambda they want the scattering matrix at; a small driver collects the requests of all
points and answers them with one vectorised evaluation.  The numbers are those of the sequential loops.

Pinned by the reference's own tests: the 10 analytic files of test/fnft_nsep (plane wave / constant signal with their
error bounds) and the 5 numerical files (signals and expected spectra extracted as numbers into
tests/golden/nsep_fixtures.npz).
"""
import ctypes as C
import math

import numpy as np

from .inverse import poly_roots_fftgridsearch
from .oracle import NSE_DISC, _c128, _ptr, poly_roots

LOC = {"SUBSAMPLE_AND_REFINE": 0, "GRIDSEARCH": 1, "MIXED": 2}
FILT = {"NONE": 0, "MANUAL": 1, "AUTO": 2}
OVERSAMPLING = 32   # src/fnft_nsep.c:41
EC_INVALID_ARGUMENT, EC_DIV_BY_ZERO, EC_OTHER, EC_NOT_YET_IMPLEMENTED, EC_ASSERTION_FAILED = 2, 3, 5, 6, 8


def default_opts():
    """src/fnft_nsep.c:26-39"""
    return {"localization": "MIXED", "filtering": "AUTO", "max_evals": 20,
            "bounding_box": [-math.inf, math.inf, -math.inf, math.inf], "normalization_flag": 1,
            "discretization": "2SPLIT2A", "floquet_range": [-1.0, 1.0], "points_per_spine": 2, "Dsub": 0, "tol": -1.0}


def _degree(orc, disc):
    n = NSE_DISC[disc]
    return int(orc.lib.orc_akns_degree(orc.lib.orc_nse_to_akns(n))), int(orc.lib.orc_nse_upsampling(n))


def scatter_matrix(orc, q_pre, eps_t, kappa, lam, ups):
    """fnft__nse_scatter_matrix with derivative (BO for ups = 1, CF4_2 for ups = 2) -> [K, 8]."""
    lam = _c128(np.atleast_1d(lam))
    q_pre = _c128(q_pre)
    out = np.zeros((lam.size, 8), np.complex128)
    orc.lib.orc_nse_scatter_matrix.argtypes = [C.c_size_t, C.c_void_p, C.c_double, C.c_int, C.c_size_t, C.c_void_p,
                                               C.c_void_p, C.c_int, C.c_int]
    rc = orc.lib.orc_nse_scatter_matrix(q_pre.size, _ptr(q_pre), float(eps_t), int(kappa), lam.size, _ptr(lam), _ptr(out),
                                        int(ups), 1)
    if rc != 0:
        raise RuntimeError("orc_nse_scatter_matrix rc=%d" % rc)
    return out


def gridsearch_roots(orc, p, M, PHI):
    """fnft__poly_roots_fftgridsearch (src/private/fnft__poly_roots_fftgridsearch.c:35-151) with the three rings from
    the C oracle's chirp z-transform and the candidate test / linear fit as array operations; the same numbers as
    oracle/inverse.py's point-by-point restatement (tests/test_nsep_oracle.py checks that), fast enough for the grids
    of 32*deg points fnft_nsep uses."""
    p = np.asarray(p, np.complex128)
    eps = (PHI[1] - PHI[0]) / (M - 1)
    W = complex(math.cos(eps), math.sin(eps))
    e0 = complex(math.cos(PHI[0]), -math.sin(PHI[0]))
    vals = np.stack([orc.poly_chirpz(p, (1.0 + k * eps) * e0, W, M) for k in (-1, 0, 1)])
    if M < 3:
        return np.zeros(0, np.complex128)
    i = np.arange(1, M - 1)
    a = np.abs(vals)
    tmp = a[1, i]
    nb = np.stack([a[0, i - 1], a[0, i], a[0, i + 1], a[1, i - 1], a[1, i + 1], a[2, i - 1], a[2, i], a[2, i + 1]])
    cand = ~np.any(tmp[None, :] > nb, axis=0)
    i = i[cand]
    if i.size == 0:
        return np.zeros(0, np.complex128)
    z0 = np.exp(1j * (PHI[0] + i * eps))
    y0 = vals[1, i]
    c = np.zeros(i.size, np.complex128)
    den = np.zeros(i.size)
    for dj in (-1, 0, 1):
        j = i + dj
        for k in (-1, 0, 1):
            skip = (j == 0) & (k == 0)   # the reference's test, :113 (it only ever skips at the left end)
            zi = (1 - k * eps) * np.exp(1j * (PHI[0] + j * eps))
            yi = vals[k + 1, j]
            c += np.where(skip, 0.0, np.conj(zi - z0) * (yi - y0))
            den += np.where(skip, 0.0, np.abs(zi - z0) ** 2)
    c = c / den
    with np.errstate(all="ignore"):
        zr = np.where(c == 0, z0, z0 - y0 / c)
    keep = np.where(c == 0, y0 == 0, np.abs(zr - z0) <= eps)
    return zr[keep]


def _z_to_lambda(z, eps_t, deg1, ups):
    """src/private/fnft__akns_discretization.c:224-239"""
    return np.log(np.asarray(z, np.complex128)) / (2j * eps_t / (deg1 * ups))


def _filter(vals, box):
    """misc_filter, src/private/fnft__misc.c:114-157"""
    if not (box[0] <= box[1]) or not (box[2] <= box[3]):
        raise ValueError("bounding_box")
    v = np.asarray(vals, np.complex128)
    with np.errstate(invalid="ignore"):
        keep = (v.real >= box[0]) & (v.real <= box[1]) & (v.imag >= box[2]) & (v.imag <= box[3])
    return v[keep]


def _filter_nonreal(vals, tol_im):
    """misc_filter_nonreal, :205-226"""
    v = np.asarray(vals, np.complex128)
    return v[np.abs(v.imag) > tol_im]


def _update_bounding_box_if_auto(eps_t, map_coeff, o):
    if FILT[o["filtering"]] == FILT["AUTO"]:
        o["bounding_box"][1] = 0.9 * math.pi / (abs(map_coeff) * eps_t)
        o["bounding_box"][0] = -o["bounding_box"][1]
        o["bounding_box"][3] = -math.log(0.1) / (abs(map_coeff) * eps_t)
        o["bounding_box"][2] = -o["bounding_box"][3]


class _DivByZero(Exception):
    pass


def _refine_main_point(lam, max_evals, rhs, tol):
    """src/fnft_nsep.c:708-792 for ONE estimate; yields the lambdas it evaluates, returns the refined point."""
    M = yield lam
    next_f, next_fp = M[0] + M[3] + rhs, M[4] + M[7]
    nevals = 1
    while nevals <= max_evals:
        f, fp = next_f, next_fp
        if fp == 0:
            raise _DivByZero()
        incr = f / fp
        min_abs, best_m = math.inf, 1
        for m in (1, 2):   # the root may be single or double: the better of the two trial points is kept
            M = yield lam - m * incr
            nevals += 1
            tmp = M[0] + M[3] + rhs
            cur = abs(tmp)
            if cur < min_abs:
                min_abs, best_m, next_f, next_fp = cur, m, tmp, M[4] + M[7]
                if cur < tol:
                    break
        lam = lam - best_m * incr
        if min_abs < tol:
            if next_fp == 0:
                raise _DivByZero()
            lam = lam - next_f / next_fp
            break
    return lam


def _refine_aux_point(lam, max_evals, tol):
    """src/fnft_nsep.c:794-836 for ONE estimate."""
    nevals = 0
    while nevals < max_evals:
        M = yield lam
        nevals += 1
        f, fp = M[1], M[5]
        if fp == 0:
            raise _DivByZero()
        lam = lam - f / fp
        if abs(f) < tol:
            break
    return lam


def _run_together(gens, evaluate):
    """Advance the per-point coroutines in lock step: one vectorised evaluation answers the pending request of every
    unfinished point."""
    out = [None] * len(gens)
    pending = {}
    for i, g in enumerate(gens):
        try:
            pending[i] = next(g)
        except StopIteration as e:
            out[i] = e.value
    while pending:
        idx = list(pending)
        S = evaluate(np.array([pending[i] for i in idx], np.complex128))
        nxt = {}
        for j, i in enumerate(idx):
            try:
                nxt[i] = gens[i].send(S[j])
            except StopIteration as e:
                out[i] = e.value
        pending = nxt
    return np.array(out, np.complex128)


def _refine_mainspec(orc, q_pre, eps_t, ups, pts, max_evals, rhs, tol, kappa):
    if max_evals == 0 or len(pts) == 0:
        return np.asarray(pts, np.complex128)
    return _run_together([_refine_main_point(complex(p), max_evals, rhs, tol) for p in pts],
                         lambda lam: scatter_matrix(orc, q_pre, eps_t, kappa, lam, ups))


def _refine_auxspec(orc, q_pre, eps_t, ups, pts, max_evals, tol, kappa):
    if max_evals == 0 or len(pts) == 0:
        return np.asarray(pts, np.complex128)
    return _run_together([_refine_aux_point(complex(p), max_evals, tol) for p in pts],
                         lambda lam: scatter_matrix(orc, q_pre, eps_t, kappa, lam, ups))


def _gridsearch(orc, q, T, K_cap, want_main, M_cap, want_aux, kappa, o):
    """src/fnft_nsep.c:222-439 -> (rc, main, aux)"""
    disc = o["discretization"]
    deg1, ups = _degree(orc, disc)
    D = q.size
    eps_t = (T[1] - T[0]) / D
    rc, q_pre, _, _ = orc.preprocess(q, eps_t, D, disc)
    if rc != 0:
        return -abs(rc), None, None
    rc, deg, tm, W = orc.nse_fscatter_pre(q_pre, eps_t, kappa, NSE_DISC[disc])
    if rc != 0:
        return -abs(rc), None, None
    if not o["normalization_flag"]:
        tm, W = tm * 2.0 ** W, 0
    map_coeff = 2.0 / deg1
    _update_bounding_box_if_auto(eps_t, map_coeff, o)
    box = o["bounding_box"]
    PHI = sorted([map_coeff * eps_t * box[0], map_coeff * eps_t * box[1]])
    filt = FILT[o["filtering"]] != FILT["NONE"]
    main, aux = np.zeros(0, np.complex128), np.zeros(0, np.complex128)
    if want_main:
        p = tm[0] + np.conj(tm[0][::-1])     # p(z) ~ z^(deg/2) (Delta(z) -/+ 2)
        found = []
        for shift in (2.0, -2.0):
            pp = p.copy()
            pp[deg // 2] += shift * 2.0 ** (-W)
            r = gridsearch_roots(orc, pp, OVERSAMPLING * deg, PHI)
            if r.size > deg:
                return EC_OTHER, None, None
            r = _z_to_lambda(r, eps_t, deg1, ups)
            if filt:
                r = _filter(r, box)
            found.append(r)
        main = np.concatenate(found)[:K_cap]
    if want_aux:
        r = gridsearch_roots(orc, tm[1], OVERSAMPLING * deg, PHI)
        r = _z_to_lambda(r, eps_t, deg1, ups)
        if filt:
            r = _filter(r, box)
        aux = r[:M_cap]
    return 0, main, aux


def _subsample_and_refine(orc, q, T, K_cap, want_main, M_cap, want_aux, kappa, o, skip_real):
    """src/fnft_nsep.c:441-706 -> (rc, main, aux)"""
    disc = o["discretization"]
    deg1, ups = _degree(orc, disc)
    D = q.size
    eps_t = (T[1] - T[0]) / D
    rc, q_full, _, _ = orc.preprocess(q, eps_t, D, disc)
    if rc != 0:
        return -abs(rc), None, None
    Dsub = o["Dsub"]
    if Dsub == 0:
        Dsub = int(2.0 ** math.ceil(0.5 * math.log2(D * math.log2(D) * math.log2(D))))
    else:
        Dsub = int(2.0 ** round(math.log2(Dsub)))
    rc, q_sub, Dsub, fl = orc.preprocess(q, eps_t, Dsub, disc)
    if rc != 0:
        return -abs(rc), None, None
    nskip = D // Dsub
    if fl[0] != 0 or fl[1] + nskip != D:
        return EC_ASSERTION_FAILED, None, None
    refine_tol = math.sqrt(np.finfo(float).eps) if o["tol"] < 0 else o["tol"]
    eps_t_sub = nskip * eps_t
    rc, deg, tm, W = orc.nse_fscatter_pre(q_sub, eps_t_sub, kappa, NSE_DISC[disc])
    if rc != 0:
        return -abs(rc), None, None
    if not o["normalization_flag"]:
        tm, W = tm * 2.0 ** W, 0
    map_coeff = 2.0 / deg1
    _update_bounding_box_if_auto(eps_t_sub, map_coeff, o)
    box = o["bounding_box"]
    tol_im = (box[1] - box[0]) / (OVERSAMPLING * (D - 1))
    filt = FILT[o["filtering"]] != FILT["NONE"]
    main, aux = [], np.zeros(0, np.complex128)
    try:
        if want_main:
            p = tm[0] + np.conj(tm[0][::-1])
            rhs_0, rhs_1 = o["floquet_range"]
            nvals = o["points_per_spine"]
            step = (rhs_1 - rhs_0) / (nvals - 1) if nvals > 1 else (rhs_1 - rhs_0)
            center = p[deg // 2]
            K = 0
            for nval in range(nvals):
                rhs = 2.0 * (rhs_0 + nval * step)
                pp = p.copy()
                pp[deg // 2] = center - rhs * 2.0 ** (-W)
                r = _z_to_lambda(poly_roots(pp), eps_t_sub, deg1, ups)
                if filt:
                    r = _filter(r, box)
                if skip_real:
                    r = _filter_nonreal(r, tol_im)
                r = _refine_mainspec(orc, q_full, eps_t, ups, r, o["max_evals"], -rhs, refine_tol, kappa)
                if filt:
                    r = _filter(r, box)
                if skip_real:
                    r = _filter_nonreal(r, tol_im)
                full = K + r.size > K_cap
                r = r[: K_cap - K]
                main.append(r)
                K += r.size
                if full:
                    break
        if want_aux:
            r = _z_to_lambda(poly_roots(tm[1]), eps_t_sub, deg1, ups)
            if filt:
                r = _filter(r, box)
            r = _refine_auxspec(orc, q_full, eps_t, ups, r, o["max_evals"], refine_tol, kappa)
            if filt:
                r = _filter(r, box)
            if skip_real:
                r = _filter_nonreal(r, tol_im)
            aux = r[:M_cap]
    except _DivByZero:
        return -EC_DIV_BY_ZERO, None, None
    main = np.concatenate(main) if main else np.zeros(0, np.complex128)
    return 0, main, aux


def fnft_nsep(orc, q, T, phase_shift=0.0, kappa=+1, opts=None, K_cap=None, M_cap=None, want_main=True, want_aux=True):
    """src/fnft_nsep.c:82-220.  Returns (rc, main_spec, aux_spec); opts as default_opts() (bounding_box is updated in
    place when filtering is AUTO, as the reference does)."""
    q = _c128(q)
    D = q.size
    if D < 2 or (D & (D - 1)) != 0:
        return EC_INVALID_ARGUMENT, None, None
    if T is None or not (T[0] < T[1]):
        return EC_INVALID_ARGUMENT, None, None
    if abs(kappa) != 1:
        return EC_INVALID_ARGUMENT, None, None
    o = default_opts() if opts is None else opts
    if FILT[o["filtering"]] != FILT["NONE"] and not want_main and want_aux:
        return EC_INVALID_ARGUMENT, None, None
    deg1, _ = _degree(orc, o["discretization"])
    K_cap = 2 * deg1 * D + 1 if K_cap is None else K_cap
    M_cap = 2 * deg1 * D + 1 if M_cap is None else M_cap
    Lam_shift = phase_shift / (-2.0 * (T[1] - T[0]))
    eps_t = (T[1] - T[0]) / D
    qp = q * np.exp(2j * Lam_shift * (T[0] + eps_t * np.arange(D)))
    manual = FILT[o["filtering"]] == FILT["MANUAL"]
    if manual:
        o["bounding_box"][0] -= Lam_shift
        o["bounding_box"][1] -= Lam_shift
    try:
        loc = LOC[o["localization"]]
        if loc == LOC["MIXED"]:
            if kappa == +1:
                rc, m1, a1 = _subsample_and_refine(orc, qp, T, K_cap, want_main, M_cap, want_aux, kappa, o, True)
            else:
                rc, m1, a1 = _subsample_and_refine(orc, qp, T, 0, False, M_cap, want_aux, kappa, o, True)
            if rc != 0:
                return -abs(rc), None, None
            rc, m2, a2 = _gridsearch(orc, qp, T, K_cap - m1.size, want_main, M_cap - a1.size, want_aux, kappa, o)
            if rc != 0:
                return -abs(rc), None, None
            main, aux = np.concatenate([m1, m2]), np.concatenate([a1, a2])
        elif loc == LOC["SUBSAMPLE_AND_REFINE"]:
            rc, main, aux = _subsample_and_refine(orc, qp, T, K_cap, want_main, M_cap, want_aux, kappa, o, False)
            if rc != 0:
                return -abs(rc), None, None
        else:
            rc, main, aux = _gridsearch(orc, qp, T, K_cap, want_main, M_cap, want_aux, kappa, o)
            if rc != 0:
                return -abs(rc), None, None
        return 0, main + Lam_shift, aux + Lam_shift
    finally:
        if manual:
            o["bounding_box"][0] += Lam_shift
            o["bounding_box"][1] += Lam_shift
