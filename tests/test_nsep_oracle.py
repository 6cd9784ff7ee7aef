"""The oracle's fnft_nsep (oracle/nsep.py) against the reference's own 15 test files (test/fnft_nsep/*.c): this is what
pins the restatement the GPU path is compared with (tests/test_gpu_nsep.py)."""
import copy
import math

import numpy as np
import pytest

import nsep_cases as NC
from oracle import nsep as ON
from oracle.inverse import poly_roots_fftgridsearch

# Harness calls whose subsampled polynomial has a degree above 1024 (LAPACK's dense companion matrix: 10 s per root
# finding) or whose signal is longer than 2048 samples are left to the GPU suite, which replays EVERY harness call of
# every file (tests/test_gpu_nsep.py asserts that); 22 of the 30 analytic harness calls run here.
SLOW_D = 2048
SLOW_DEG = 1024


def _sub_degree(opts, D):
    Dsub = int(2.0 ** math.ceil(0.5 * math.log2(D * math.log2(D) * math.log2(D))))
    return NC.DEG[opts["discretization"]] * (2 if opts["discretization"].startswith("4SPLIT") else 1) * min(Dsub, D)


def is_slow(stage):
    return stage[3] > SLOW_D or _sub_degree(stage[5], stage[3]) > SLOW_DEG


def _opts(d):
    o = ON.default_opts()
    for k, v in d.items():
        if k == "bounding_box":
            o["bounding_box"] = [o["bounding_box"][i] if v[i] is None else v[i] for i in range(4)]
        else:
            o[k] = copy.deepcopy(v)
    return o


@pytest.fixture(scope="module")
def orc():
    from oracle.oracle import load_oracle
    return load_oracle()


@pytest.mark.parametrize("name,i,tc,D,bounds,opts", [pytest.param(*s, id="%s-%d" % (s[0], s[1])) for s in NC.analytic_stages()
                                                     if not is_slow(s)])
def test_analytic_files(orc, name, i, tc, D, bounds, opts):
    """every harness call nsep_testcases_test_fnft(tc, D, error_bounds, &opts) of the 10 analytic files, with the
    file's own bounds (src/private/fnft__nsep_testcases.c:283-400)"""
    q, T, ps, ms_x, au_x, kappa, remove_box = NC.testcase(tc, D)
    o = _opts(opts)
    cap = NC.capacity(opts, D)
    rc, ms, au = ON.fnft_nsep(orc, q, T, ps, kappa, o, K_cap=cap, M_cap=cap)
    assert rc == 0
    e_main, e_aux = NC.compare(ms, au, ms_x, au_x, o["bounding_box"], remove_box)
    assert e_main <= bounds[0] and e_aux <= bounds[1], (e_main, e_aux, bounds)


def _numeric(orc, name, points_per_spine=None, want_aux=True, cap=None):
    rec = NC.FIX["numeric"][name]
    q = NC.ARR[name + "/q"]
    D = q.size
    o = _opts(rec["opts"])
    if points_per_spine:
        o["points_per_spine"] = points_per_spine
    phase_shift = float(np.angle(q[D - 1] / q[0]))
    K = cap if cap is not None else D
    return ON.fnft_nsep(orc, q[: D - 1], rec["T"], phase_shift, rec["kappa"], o, K_cap=K, M_cap=D, want_aux=want_aux), rec


@pytest.mark.parametrize("name", ["numerical_focusing_1", "numerical_focusing_3", "numerical_defocusing_1"])
def test_numerical_main_and_aux(orc, name):
    (rc, ms, au), rec = _numeric(orc, name)
    assert rc == 0
    assert NC.hausdorff(NC.ARR[name + "/mainspec_exact"], ms) <= rec["dist_bounds"][0]
    assert NC.hausdorff(NC.ARR[name + "/auxspec_exact"], au) <= rec["dist_bounds"][1]


def test_numerical_focusing_2_main(orc):
    (rc, ms, au), rec = _numeric(orc, "numerical_focusing_2", want_aux=False)
    assert rc == 0
    assert NC.hausdorff(NC.ARR["numerical_focusing_2/mainspec_exact"], ms) <= rec["dist_bounds"][0]


@pytest.mark.parametrize("name", ["numerical_focusing_1", "numerical_focusing_2"])
def test_numerical_spines(orc, name):
    rec = NC.FIX["numeric"][name]
    D = NC.ARR[name + "/q"].size
    (rc, sp, _), _ = _numeric(orc, name, points_per_spine=rec["points_per_spine"], want_aux=False,
                              cap=D * rec["points_per_spine"])
    assert rc == 0
    on, flags = NC.spine_check(sp, rec["spine_tol"], rec["spine_real_eps"])
    assert on and all(flags), flags


def test_nonregression_1(orc):
    """494 spine points the reference itself printed (version 0.4.1), Hausdorff distance 1e-12 in the file; the oracle's
    LAPACK roots differ from eiscor's in the last digits before the refinement, hence a bound of 1e-10 here"""
    q, T = NC.nonregression_signal()
    o = _opts(NC.FIX["numeric"]["nonregression_1"]["opts"])
    o["points_per_spine"] = 100
    rc, sp, _ = ON.fnft_nsep(orc, q, T, 0.0, +1, o, K_cap=500, M_cap=2, want_aux=False)
    assert rc == 0
    assert NC.hausdorff(NC.ARR["nonregression_1/spines_exact"], sp) <= 1e-10


def test_gridsearch_fast_equals_pointwise(orc):
    """the array form of fnft__poly_roots_fftgridsearch in oracle/nsep.py against the point-by-point restatement"""
    rng = np.random.default_rng(3)
    r = np.exp(1j * np.array([0.3, 1.1, 2.0, 4.0, 5.5])) * np.array([1.0, 1.0, 0.9, 1.0, 1.2])
    p = np.poly(np.concatenate([r, 0.5 * rng.standard_normal(3) + 0.5j]))
    for PHI in ([0.0, 2 * math.pi], [0.2, 3.0]):
        a = ON.gridsearch_roots(orc, p, 400, PHI)
        b = poly_roots_fftgridsearch(p, 400, PHI)
        assert a.size == b.size and np.allclose(a, b, rtol=0, atol=1e-12)


def test_argument_checks(orc):
    q = np.ones(64, np.complex128)
    assert ON.fnft_nsep(orc, q[:48], [0, 1])[0] == 2          # D not a power of two
    assert ON.fnft_nsep(orc, q, [1, 0])[0] == 2
    assert ON.fnft_nsep(orc, q, [0, 1], kappa=0)[0] == 2
    assert ON.fnft_nsep(orc, q, [0, 1], want_main=False)[0] == 2   # filtering needs the main spectrum
