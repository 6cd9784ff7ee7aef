"""fnft_nsep through the C ABI on the GPU: EVERY harness call of all 15 files of the reference's test/fnft_nsep with the
files' own bounds, agreement with the oracle (oracle/nsep.py, pinned by the same files in tests/test_nsep_oracle.py) on
the same inputs, the seams it rides on for the CF4_2 scheme, and its argument checks."""
import copy
import math

import numpy as np
import pytest

import nsep_cases as NC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from fnft_amd import capi as c
    c.load()
    return c


@pytest.fixture(scope="module")
def orc():
    from oracle.oracle import load_oracle
    return load_oracle()


def _oracle_opts(d):
    from oracle import nsep as ON
    o = ON.default_opts()
    for k, v in d.items():
        if k == "bounding_box":
            o["bounding_box"] = [o["bounding_box"][i] if v[i] is None else v[i] for i in range(4)]
        else:
            o[k] = copy.deepcopy(v)
    return o


def test_every_reference_file_is_replayed():
    assert len(NC.FIX["analytic"]) == 10 and len(NC.FIX["numeric"]) == 5
    assert len(NC.analytic_stages()) == 30


@pytest.mark.parametrize("name,i,tc,D,bounds,opts", [pytest.param(*s, id="%s-%d" % (s[0], s[1])) for s in NC.analytic_stages()])
def test_analytic_files(capi, name, i, tc, D, bounds, opts):
    """nsep_testcases_test_fnft(tc, D, error_bounds, &opts) (src/private/fnft__nsep_testcases.c:283-400) for every
    harness call of the 10 analytic files, D up to 4096"""
    q, T, ps, ms_x, au_x, kappa, remove_box = NC.testcase(tc, D)
    cap = NC.capacity(opts, D)
    o = capi.nsep_opts(opts)
    rc, ms, au = capi.fnft_nsep(q, T, ps, kappa, o, K=cap, M=cap)
    assert rc == 0, capi.last_error()
    e_main, e_aux = NC.compare(ms, au, ms_x, au_x, list(o.bounding_box), remove_box)
    assert e_main <= bounds[0] and e_aux <= bounds[1], (e_main, e_aux, bounds)
    # the options object is handed back as it came (the manual bounding box is shifted and restored, :138-141, :210-213)
    assert [o.bounding_box[k] for k in range(4)] == [float(v) for v in opts["bounding_box"]]


def _numeric(capi, name, points_per_spine=None, want_aux=True, cap=None):
    rec = NC.FIX["numeric"][name]
    q = NC.ARR[name + "/q"]
    D = q.size
    d = dict(rec["opts"])
    if points_per_spine:
        d["points_per_spine"] = points_per_spine
    phase_shift = float(np.angle(q[D - 1] / q[0]))
    K = cap if cap is not None else D
    return capi.fnft_nsep(q[: D - 1], rec["T"], phase_shift, rec["kappa"], d, K=K, M=D, want_aux=want_aux), rec


@pytest.mark.parametrize("name", ["numerical_focusing_1", "numerical_focusing_3", "numerical_defocusing_1"])
def test_numerical_main_and_aux(capi, name):
    (rc, ms, au), rec = _numeric(capi, name)
    assert rc == 0, capi.last_error()
    assert NC.hausdorff(NC.ARR[name + "/mainspec_exact"], ms) <= rec["dist_bounds"][0]
    assert NC.hausdorff(NC.ARR[name + "/auxspec_exact"], au) <= rec["dist_bounds"][1]


def test_numerical_focusing_2_main(capi):
    (rc, ms, au), rec = _numeric(capi, "numerical_focusing_2", want_aux=False)
    assert rc == 0, capi.last_error()
    assert NC.hausdorff(NC.ARR["numerical_focusing_2/mainspec_exact"], ms) <= rec["dist_bounds"][0]


@pytest.mark.parametrize("name", ["numerical_focusing_1", "numerical_focusing_2"])
def test_numerical_spines(capi, name):
    rec = NC.FIX["numeric"][name]
    D = NC.ARR[name + "/q"].size
    (rc, sp, _), _ = _numeric(capi, name, points_per_spine=rec["points_per_spine"], want_aux=False,
                              cap=D * rec["points_per_spine"])
    assert rc == 0, capi.last_error()
    on, flags = NC.spine_check(sp, rec["spine_tol"], rec["spine_real_eps"])
    assert on and all(flags), flags


def test_nonregression_1(capi):
    """test/fnft_nsep/fnft_nsep_test_nonregression_1.c: 494 spine points printed by the reference (eiscor roots, then
    Newton refinement to sqrt(eps)); the Ehrlich-Aberth estimates differ from eiscor's in the last digits before the
    refinement, so the file's 1e-12 becomes 1e-10 (the oracle with LAPACK roots: the same)"""
    q, T = NC.nonregression_signal()
    d = dict(NC.FIX["numeric"]["nonregression_1"]["opts"])
    d["points_per_spine"] = 100
    rc, sp, _ = capi.fnft_nsep(q, T, 0.0, +1, d, K=500, M=2, want_aux=False)
    assert rc == 0, capi.last_error()
    assert sp.size == 494
    assert NC.hausdorff(NC.ARR["nonregression_1/spines_exact"], sp) <= 1e-10


@pytest.mark.parametrize("tc,D,disc,loc,tol", [("PLANE_WAVE_FOCUSING", 256, "2SPLIT2A", "MIXED", 2e-3),
                                               ("PLANE_WAVE_FOCUSING", 256, "2SPLIT4B", "SUBSAMPLE_AND_REFINE", 2e-3),
                                               ("PLANE_WAVE_FOCUSING", 128, "4SPLIT4B", "MIXED", 2e-3),
                                               ("CONSTANT_DEFOCUSING", 256, "2SPLIT4A", "GRIDSEARCH", 1e-6),
                                               ("CONSTANT_DEFOCUSING", 128, "4SPLIT4A", "MIXED", 1e-6),
                                               ("PLANE_WAVE_FOCUSING", 512, "2SPLIT2_MODAL", "GRIDSEARCH", 1e-6)])
def test_gpu_vs_oracle(capi, orc, tc, D, disc, loc, tol):
    """The same call on the GPU and through the oracle: the same SET of spectral points (Hausdorff distance; the root
    finders order their output differently).  Grid-search points agree to the round-off of the chirp z-transform of a
    degree-D polynomial times the slope of the linear fit (1e-6).  The plane wave's main spectrum consists of DOUBLE
    points: Newton's method for multiple roots stops at |f| < sqrt(eps), i.e. at a distance ~ eps^(1/4) from the root,
    wherever it started -- Ehrlich-Aberth and LAPACK estimates end on different points inside that ball, which is why the
    reference's own bounds for these files are 1e-4 .. 1e-5; two answers inside it may differ by its diameter."""
    from oracle import nsep as ON
    q, T, ps, ms_x, au_x, kappa, remove_box = NC.testcase(tc, D)
    d = {"discretization": disc, "localization": loc, "filtering": "MANUAL", "bounding_box": [-10, 10, -10, 10]}
    cap = NC.capacity(d, D)
    rc, ms, au = capi.fnft_nsep(q, T, ps, kappa, d, K=cap, M=cap)
    assert rc == 0, capi.last_error()
    rc2, ms_o, au_o = ON.fnft_nsep(orc, q, T, ps, kappa, _oracle_opts(d), K_cap=cap, M_cap=cap)
    assert rc2 == 0
    if tol < 1e-4:   # simple points: the same number of them
        assert ms.size == ms_o.size and au.size == au_o.size, (ms.size, ms_o.size, au.size, au_o.size)
    assert (ms.size == 0) == (ms_o.size == 0) and (au.size == 0) == (au_o.size == 0)
    if ms.size:
        assert NC.hausdorff(ms, ms_o) < tol
    if au.size:
        assert NC.hausdorff(au, au_o) < tol


def test_scatter_matrix_cf4_2_vs_oracle(capi, orc):
    """fnft__nse_scatter_matrix with CF4_2 (what fnft_nsep refines with for 4SPLIT4A/B): S and dS/dlambda against the
    oracle's restatement of src/private/fnft__akns_scatter_matrix.c:122-208"""
    from oracle import nsep as ON
    rng = np.random.default_rng(5)
    D = 200
    q = 0.7 * (rng.standard_normal(D) + 1j * rng.standard_normal(D))
    lam = np.array([0.3, -1.2 + 0.4j, 2.0j, 0.0])
    for kappa in (1, -1):
        ref = ON.scatter_matrix(orc, q, 0.05, kappa, lam, 2)
        rc, res = capi.nse_scatter_matrix(q, 0.05, kappa, lam, discretization="CF4_2")
        assert rc == 0, capi.last_error()
        assert np.max(np.abs(res - ref)) / np.max(np.abs(ref)) < 1e-13
    assert capi.nse_scatter_matrix(q[:199], 0.05, 1, lam, discretization="CF4_2")[0] == 8   # odd D: assertion, :123-126
    assert capi.nse_scatter_matrix(q, 0.05, 1, lam, discretization="CF4_3")[0] == 6


def test_argument_checks(capi):
    capi.silence_errors()
    q = np.ones(64, np.complex128)
    assert capi.fnft_nsep(q[:48], [0, 1])[0] == 2             # D not a power of two, src/fnft_nsep.c:100-101
    assert capi.fnft_nsep(q, [1, 0])[0] == 2
    assert capi.fnft_nsep(q, [0, 1], kappa=0)[0] == 2
    assert capi.fnft_nsep(q, [0, 1], sheet_indices=True)[0] == capi.FNFT_EC_NOT_YET_IMPLEMENTED
    assert capi.fnft_nsep(q, [0, 1], want_main=False)[0] == 2   # filtering the aux spectrum needs the main spectrum
    assert capi.fnft_nsep(q, [0, 1], opts={"discretization": "BO"})[0] == capi.FNFT_EC_NOT_YET_IMPLEMENTED
    o = capi.nsep_default_opts()
    assert (o.localization, o.filtering, o.max_evals, o.discretization, o.normalization_flag, o.points_per_spine, o.Dsub) \
        == (2, 2, 20, capi.NSE_DISC["2SPLIT2A"], 1, 2, 0)
    assert o.tol == -1 and list(o.floquet_range) == [-1, 1] and o.bounding_box[0] == -math.inf


def test_too_small_arrays_warn(capi):
    """more points than the caller's arrays hold: as many as fit, one warning per spectrum (:371-377, :622-628)"""
    msgs = capi.capture_messages() if hasattr(capi, "capture_messages") else None
    q, T, ps, ms_x, au_x, kappa, _ = NC.testcase("PLANE_WAVE_FOCUSING", 256)
    d = {"discretization": "2SPLIT2A", "filtering": "MANUAL", "bounding_box": [-10, 10, -10, 10]}
    rc, ms, au = capi.fnft_nsep(q, T, ps, kappa, d, K=3, M=2)
    assert rc == 0 and ms.size == 3 and au.size == 2
    if msgs is not None:
        text = "".join(msgs())
        assert "Found more than *K_ptr main spectrum points" in text and "Found more than *M_ptr aux" in text
