"""The analytic test cases of the reference's fnft_nsep harness, restated (numbers and formulas only):
src/private/fnft__nsep_testcases.c:29-230 (signals and exact spectra) and :283-400 (nsep_testcases_test_fnft: what is
compared with what).  Used by tests/test_nsep_oracle.py (CPU, oracle) and tests/test_gpu_nsep.py (GPU, C ABI)."""
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "nsep_fixtures.json")))
ARR = np.load(os.path.join(HERE, "golden", "nsep_fixtures.npz"))
DEG = {"2SPLIT2_MODAL": 1, "2SPLIT2A": 1, "2SPLIT4A": 4, "2SPLIT4B": 2, "4SPLIT4A": 4, "4SPLIT4B": 2}


def testcase(tc, D):
    """-> q, T, phase_shift, mainspec_exact, auxspec_exact, kappa, remove_box  (fnft__nsep_testcases.c:64-230)"""
    if tc == "PLANE_WAVE_FOCUSING":
        K = 100
        M = K - 2
        T = [0.0, 2.0 * math.pi]
        eps_t = 2.0 * math.pi / D
        q = 2.0 * np.exp(3.0j * (T[0] + np.arange(D) * eps_t))
        j = np.arange(K // 2)
        ms = np.empty(K, np.complex128)
        ms[0::2] = -1.5 + 1j * np.sqrt(4.0 - j * j / 4.0 + 0j)
        ms[1::2] = -1.5 - 1j * np.sqrt(4.0 - j * j / 4.0 + 0j)
        j = np.arange(M // 2) + 1   # "+1": the points with the largest imaginary parts are skipped
        au = np.empty(M, np.complex128)
        au[0::2] = -1.5 + 1j * np.sqrt(4.0 - j * j / 4.0 + 0j)
        au[1::2] = -1.5 - 1j * np.sqrt(4.0 - j * j / 4.0 + 0j)
        return q, T, 0.0, ms, au, +1, [-1.6, -1.4, -0.1, 0.1]
    if tc == "CONSTANT_DEFOCUSING":
        K = 100
        T = [0.0, 1.0]
        q = np.full(D, (1.0 + 2.0j) / 5.0, np.complex128)
        pi2 = math.pi ** 2
        ms = np.zeros(K, np.complex128)
        ms[0] = 1.0 / math.sqrt(5.0)
        ms[1] = -ms[0]
        ms[2] = math.sqrt(5.0 * pi2 + 1.0) / math.sqrt(5.0)
        ms[3] = -ms[2]
        j = 1
        while True:
            i = 3 + 4 * j
            if i >= K:
                break
            ms[i - 3] = math.sqrt(20.0 * pi2 * j * j + 1.0) / math.sqrt(5.0)
            ms[i - 2] = -ms[i - 3]
            ms[i - 1] = math.sqrt(20.0 * pi2 * j * j + 20.0 * pi2 * j + 5.0 * pi2 + 1.0) / math.sqrt(5.0)
            ms[i] = -ms[i - 1]
            j += 1
        Kx = i - 4
        au = np.zeros(K, np.complex128)
        au[0] = math.sqrt(5.0 * pi2 + 1.0) / math.sqrt(5.0)
        au[1] = -au[0]
        j = 1
        while True:
            i = 1 + 4 * j
            if i >= Kx:
                break
            au[i - 3] = math.sqrt(20.0 * pi2 * j * j + 1.0) / math.sqrt(5.0)
            au[i - 2] = -au[i - 3]
            au[i - 1] = math.sqrt(20.0 * pi2 * j * j + 20.0 * pi2 * j + 5.0 * pi2 + 1.0) / math.sqrt(5.0)
            au[i] = -au[i - 1]
            j += 1
        Mx = i - 4
        return q, T, 0.0, ms[:Kx], au[:Mx], -1, [0.0, 0.0, 0.0, 0.0]
    raise KeyError(tc)


def box_filter(v, box):
    """misc_filter, src/private/fnft__misc.c:114-157"""
    v = np.asarray(v, np.complex128)
    return v[(v.real >= box[0]) & (v.real <= box[1]) & (v.imag >= box[2]) & (v.imag <= box[3])]


def box_filter_inv(v, box):
    """misc_filter_inv (src/private/fnft__misc.c:159-203): drop the values strictly INSIDE the box"""
    v = np.asarray(v, np.complex128)
    inside = (v.real > box[0]) & (v.real < box[1]) & (v.imag > box[2]) & (v.imag < box[3])
    return v[~inside]


def hausdorff(a, b):
    """misc_hausdorff_dist, src/private/fnft__misc.c:53-83"""
    a, b = np.asarray(a, np.complex128), np.asarray(b, np.complex128)
    if a.size == 0 or b.size == 0:
        return math.inf
    d = np.abs(a[:, None] - b[None, :])
    return float(max(d.min(axis=1).max(), d.min(axis=0).max()))


def compare(main, aux, ms_exact, au_exact, bounding_box, remove_box):
    """fnft__nsep_testcases.c:345-371: exact spectra filtered by the bounding box, everything inside remove_box dropped
    on both sides, Hausdorff distances (0 when both sides are empty, NaN when one is)."""
    ms_exact = box_filter_inv(box_filter(ms_exact, bounding_box), remove_box)
    au_exact = box_filter_inv(box_filter(au_exact, bounding_box), remove_box)
    main = box_filter_inv(main, remove_box)
    aux = box_filter_inv(aux, remove_box)

    def dist(x, y):
        if x.size == 0 and y.size == 0:
            return 0.0
        if x.size == 0 or y.size == 0:
            return math.nan
        return hausdorff(x, y)
    return dist(main, ms_exact), dist(aux, au_exact)


def analytic_stages():
    """(file, stage index, testcase, D, bounds, opts) of every harness call of the 10 analytic files"""
    out = []
    for name, rec in sorted(FIX["analytic"].items()):
        for i, st in enumerate(rec["stages"]):
            out.append((name, i, rec["testcase"], st["D"], st["bounds"], st["opts"]))
    return out


def capacity(opts, D):
    """K = M = 2 * degree * D + 1 (fnft__nsep_testcases.c:309-310)"""
    return 2 * DEG[opts["discretization"]] * D + 1


def nonregression_signal():
    """test/fnft_nsep/fnft_nsep_test_nonregression_1.c:529-547"""
    D = 512
    T = [0.0, 2.0 * math.pi / 0.822]
    eps_t = (T[1] - T[0]) / D
    q = 1.0 + 0.22 * np.exp(-1j * 0.822 * (T[0] + eps_t * np.arange(D)))
    return q, T


def spine_check(spines, tol, real_eps):
    """test/fnft_nsep/fnft_nsep_test_numerical_focusing_1.c:141-190: every point on one of the three imaginary spines
    [-5i, -2i], [-i, i], [2i, 5i], and a point near the centre of each spine found.  -> (all on spines, flags)"""
    ok = [False, False, False]
    eps = np.finfo(float).eps
    for lam in spines:
        if abs(lam.real) > real_eps * eps:
            return False, ok
        li = lam.imag
        if -4.5 < li < -2.5:
            ok[0] = True
        if -5 - tol <= li <= -2 + tol:
            continue
        if abs(li) < 0.5:
            ok[1] = True
        if abs(li) <= 1 + tol:
            continue
        if 2.5 < li < 4.5:
            ok[2] = True
        if 2 - tol <= li <= 5 + tol:
            continue
        return False, ok
    return True, ok
