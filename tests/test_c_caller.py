"""A compiled C program against include/fnft_amd.h and libfnft_amd.so (the by-value entries -- the options
structs returned by value, fnft__poly_chirpz(..., const FNFT_COMPLEX A, const FNFT_COMPLEX W, ...) defined in a C++
translation unit -- had only been reached through ctypes).  CPU: it compiles, links and gets the defaults right by value,
and the compute entries fail loudly without a GPU.  GPU: its numbers are the ctypes numbers."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_caller", "c_caller.c")
EXE = os.path.join(ROOT, "tests", "c_caller", "c_caller")


def _build():
    cc = shutil.which("cc") or shutil.which("gcc")
    if cc is None:
        pytest.skip("no C compiler on this machine")
    from fnft_amd import build
    lib = build.build()
    libdir = os.path.dirname(lib)
    cmd = [cc, "-std=c11", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE, "-L", libdir, "-lfnft_amd",
           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-lm"]
    subprocess.check_call(cmd)
    return EXE


def _run():
    exe = _build()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    rec = {"cs": [], "cz": []}
    for line in out.stdout.splitlines():
        k, *v = line.split()
        if k in ("cs", "cz"):
            rec[k].append(complex(float(v[0]), float(v[1])))
        else:
            rec[k] = v
    return rec


def _check_defaults(rec):
    # src/fnft_nsev.c:26-36: FULL, SUBSAMPLE_AND_REFINE, niter 10, Dsub 0, NORMING_CONSTANTS, REFLECTION_COEFFICIENT,
    # normalization 1, 2SPLIT4B (ordinal 11), no Richardson
    assert rec["opts"] == ["2", "2", "10", "0", "0", "0", "1", "11", "0"]
    # src/fnft_kdvv.c:34-44: 2SPLIT8B (ordinal 17); src/fnft_nsep.c:26-39: MIXED, max_evals 20, tol -1
    assert rec["opts2"] == ["17", "2", "20", "-1"]


def test_c_caller_builds_and_gets_structs_by_value():
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU test")
    rec = _run()
    _check_defaults(rec)
    assert rec["nsev"] != ["0"] and rec["chirpz"] != ["0"]   # no GPU: loud failure, no CPU fallback


@pytest.mark.gpu
def test_c_caller_matches_ctypes():
    from fnft_amd import capi
    rec = _run()
    _check_defaults(rec)
    assert rec["nsev"] == ["0"] and rec["chirpz"] == ["0"]
    D, M = 256, 8
    T, XI = [-25.0, 25.0], [-1.4, 1.6]
    q = 3.2j / np.cosh(T[0] + np.arange(D) * (T[1] - T[0]) / (D - 1))
    rc, cs = capi.fnft_nsev(q, T, M, XI, kappa=1, discretization="2SPLIT2_MODAL", contspec_type="BOTH")
    # the same library; the inputs were formed by C's libm there and by numpy here (last-bit differences)
    assert rc == 0 and np.max(np.abs(np.array(rec["cs"]) - cs)) < 1e-12 * np.max(np.abs(cs))
    p = np.array([1 + 2j, -0.5, 0.25j, 3.0])
    A, W = 0.9 * np.exp(0.3j), np.exp(-0.2j)
    rc, cz = capi.poly_chirpz(p, A, W, 5)
    assert rc == 0 and np.max(np.abs(np.array(rec["cz"]) - cz)) < 1e-13 * np.max(np.abs(cz))
    ref = np.polyval(p, 1.0 / (A * W ** (-np.arange(5.0))))             # src/private/fnft__poly_chirpz.c:28-29
    assert np.max(np.abs(cz - ref)) < 100 * np.finfo(float).eps * np.max(np.abs(ref))
