#!/usr/bin/env python3
"""Extract the reference's own known-answer vectors for the fnft_nsev hot path into a fixture.

Run once in the build container (needs /root/reference); the output
tests/golden/reference_fixtures.json is committed and is all the tests read.  Only NUMBERS are
extracted (inputs, expected outputs, tolerances) -- no source text is kept.

Sources (relative to /root/reference):
  test/fnft__poly/fnft__poly_fmult2x2_test_n_is_power_of_2.c:61-81      (tree, n=4, deg 1)
  test/fnft__poly/fnft__poly_fmult2x2_test_n_is_no_power_of_2.c:68-92   (tree, n=5, deg 1)
  test/fnft__poly/fnft__poly_chirpz_test.c:28-40                         (chirp-z, deg 3)
  test/fnft__fft_wrapper/fnft__fft_wrapper_test.c                        (length-4 FFT)
  test/fnft__akns_fscatter/*.c                                           (per-scheme transfer matrices)
  src/private/fnft__nsev_testcases.c:142-287,463-567                     (analytic spectra)
  test/fnft_nsev/*.c                                                     (per-scheme error bounds)
  test/fnft__misc/fnft__misc_resample_test.c                             (band-limited resampler)
  test/fnft__poly/fnft__poly_roots_fasteigen_test.c                      (root finder, cubic)
  test/fnft__nse_scatter/fnft__nse_scatter_bound_states_test_bo.c        (slow scatterer BO: a, a', b)
  test/fnft__poly/fnft__poly_fmult_test_n_is_(no_)power_of_2.c            (scalar product tree)
  test/fnft__nse_finvscatter/*.c, ..._test.inc                            (inverse scattering round trip, bounds)
"""
import json
import math
import os
import re
import sys

REF = os.environ.get("FNFT_REFERENCE", "/root/reference")


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    s = re.sub(r"//[^\n]*", "", s)
    return s


def ceval(expr):
    """Evaluate a C complex literal expression such as '1.5 - I*2' or '3 + -2e-1*I'."""
    e = expr.replace("\\", " ").replace("\n", " ").strip()
    e = re.sub(r"\bI\b", "1j", e)
    if not re.fullmatch(r"[0-9eE+\-*/.()j\s]+", e):
        raise ValueError("unexpected token in %r" % expr)
    return complex(eval(e, {"__builtins__": {}}, {}))


def c2l(z):
    return [z.real, z.imag]


def array_init(src, name):
    m = re.search(re.escape(name) + r"\s*\[[^\]]*\]\s*=\s*\{(.*?)\}\s*;", src, flags=re.S)
    if not m:
        raise KeyError(name)
    body = m.group(1)
    return [ceval(x) for x in body.split(",") if x.strip()]


def walk_stages(src, arr_pat=r"error_bounds\w*", harness="nsev_testcases_test_fnft"):
    """Replays the body of one test/fnft_nsev/*.c file statement by statement -- D and DN updates,
    bound arrays and their rescaling loops, the Richardson flag -- and records every call of the
    harness nsev_testcases_test_fnft(tc, <D>, <bounds>, &opts) as
    {"D": .., "richardson": 0/1, "bounds": [rho, a, b]}."""
    env = {"D": None, "DN": None}
    arrays = {}
    rich = 0
    dsub = None    # opts.Dsub as last assigned (None: the default 0 = automatic)
    niter = None   # opts.niter as last assigned (None: the default 10)
    stages = []
    inf = float("inf")

    def num(expr):
        expr = expr.replace("(REAL)", "").replace("POW", "pow").replace("INFINITY", "inf").replace("FNFT_INF", "inf")
        return float(eval(expr, {"__builtins__": {}, "pow": pow, "inf": inf},
                          {k: v for k, v in env.items() if v is not None}))

    body = src[src.index("main"):] if "main" in src else src
    # a "for (i=0; i<6; i++) X;" loop is one statement applying X to every entry
    tok = re.compile(
        r"(?P<arr>REAL\s+(?P<an>" + arr_pat + r")\s*\[\s*6\s*\]\s*=\s*\{(?P<av>.*?)\}\s*;)"
        r"|(?P<decl>UINT\s+(?P<dn>D|DN)\s*=\s*(?P<dv>[^;]+);)"
        r"|(?P<loop>for\s*\([^)]*\)\s*(?P<ln>" + arr_pat + r")\s*\[\s*i\s*\]\s*(?P<lo>[*/])=\s*(?P<lv>[^;]+);)"
        r"|(?P<one>(?P<on>" + arr_pat + r")\s*\[\s*(?P<oi>\d+)\s*\]\s*(?P<oo>[*/])=\s*(?P<ov>[^;]+);)"
        r"|(?P<dop>\bD\s*(?P<do>[*/+-])=\s*(?P<dval>[^;]+);)"
        r"|(?P<dset>\bD\s*=\s*(?P<dsv>[^;=]+);)"
        r"|(?P<rich>opts\.richardson_extrapolation_flag\s*=\s*(?P<rv>\d)\s*;)"
        r"|(?P<dsub>opts\.Dsub\s*=\s*(?P<dsv2>[^;]+);)"
        r"|(?P<niter>opts\.niter\s*=\s*(?P<nv>[^;]+);)"
        r"|(?P<call>" + harness + r"\s*\(\s*tc\s*,\s*(?P<cd>[^,]+),\s*(?P<ca>\w+)\s*,)",
        flags=re.S)
    for m in tok.finditer(body):
        if m.group("arr"):
            arrays[m.group("an")] = [num(x.strip()) for x in m.group("av").split(",") if x.strip()]
        elif m.group("decl"):
            env[m.group("dn")] = int(num(m.group("dv")))
        elif m.group("loop"):
            f = num(m.group("lv"))
            a = arrays[m.group("ln")]
            arrays[m.group("ln")] = [x * f if m.group("lo") == "*" else x / f for x in a]
        elif m.group("one"):
            f = num(m.group("ov"))
            a = arrays[m.group("on")]
            i = int(m.group("oi"))
            a[i] = a[i] * f if m.group("oo") == "*" else a[i] / f
        elif m.group("dop"):
            v = num(m.group("dval"))
            op = m.group("do")
            env["D"] = int({"*": env["D"] * v, "/": env["D"] / v, "+": env["D"] + v, "-": env["D"] - v}[op])
        elif m.group("dset"):
            env["D"] = int(num(m.group("dsv")))
        elif m.group("rich"):
            rich = int(m.group("rv"))
        elif m.group("dsub"):
            dsub = int(num(m.group("dsv2")))
        elif m.group("niter"):
            niter = int(num(m.group("nv")))
        elif m.group("call"):
            stages.append({"D": int(num(m.group("cd"))), "richardson": rich,
                           "bounds": list(arrays[m.group("ca")][:3]),
                           "bounds_ds": list(arrays[m.group("ca")][3:6]),
                           "Dsub": dsub, "niter": niter})
    return stages


def main():
    out = {"_generated_by": "tests/golden/extract_reference_fixtures.py",
           "_reference": "IgorChekhovskoy/FNFT @ 2025-01-27 (FNFT 0.4.1)"}

    # ---- product tree -------------------------------------------------------------------
    for key, rel, n in (
        ("fmult2x2_pow2", "test/fnft__poly/fnft__poly_fmult2x2_test_n_is_power_of_2.c", 4),
        ("fmult2x2_nopow2", "test/fnft__poly/fnft__poly_fmult2x2_test_n_is_no_power_of_2.c", 5),
    ):
        src = strip_comments(read(rel))
        exact = array_init(src, "result_exact")
        # input rule stated in the test: p_e[i] = sqrt(i+1)*(cos(i+0.1e) + 1j*sin(-2i+0.1e))
        out[key] = {
            "deg": 1, "n": n,
            "input_rule": "p[e*n*(deg+1)+i] = sqrt(i+1)*(cos(i+0.1*e) + 1j*sin(-2*i+0.1*e)), e=0..3",
            "result_exact": [c2l(z) for z in exact],
            "tol_rel_l1": 100 * 2.220446049250313e-16,
        }

    # scalar tree, test/fnft__poly/fnft__poly_fmult_test_n_is_power_of_2.c:26-77 and ..._no_power_of_2.c:26-88
    for key, rel, n in (
        ("fmult_pow2", "test/fnft__poly/fnft__poly_fmult_test_n_is_power_of_2.c", 8),
        ("fmult_nopow2", "test/fnft__poly/fnft__poly_fmult_test_n_is_no_power_of_2.c", 11),
    ):
        src = strip_comments(read(rel))
        out[key] = {
            "deg": 2, "n": n,
            "input_rule": "p[i] = sqrt(i+1)*(cos(i) + 1j*sin(-2*i)), i < (deg+1)*n",
            "result_exact": [c2l(z) for z in array_init(src, "result_exact")],
            "tol_rel_l1": 100 * 2.220446049250313e-16,
        }

    # ---- chirp z ------------------------------------------------------------------------
    src = strip_comments(read("test/fnft__poly/fnft__poly_chirpz_test.c"))
    out["chirpz"] = {
        "deg": 3,
        "p": [c2l(z) for z in array_init(src, "p")],
        "A": [0.95, 0.0],
        "W_arg": 0.3,  # W = exp(0.3i)
        "result_M3": [c2l(z) for z in array_init(src, "result_exactM3")],
        "result_M6": [c2l(z) for z in array_init(src, "result_exactM6")],
        "tol_rel_l1": 100 * 2.220446049250313e-16,
    }

    # ---- fft wrapper ---------------------------------------------------------------------
    src = strip_comments(read("test/fnft__fft_wrapper/fnft__fft_wrapper_test.c"))
    fft = {}
    for nm in re.findall(r"COMPLEX\s+(\w+)\s*\[[^\]]*\]\s*=\s*\{", src):
        fft[nm] = [c2l(z) for z in array_init(src, nm)]
    out["fft_wrapper"] = fft

    # ---- akns_fscatter, one entry per splitting scheme ------------------------------------
    schemes = {}
    d = os.path.join(REF, "test/fnft__akns_fscatter")
    for fn in sorted(os.listdir(d)):
        src = strip_comments(read("test/fnft__akns_fscatter/" + fn))
        name = re.search(r"akns_discretization_(\w+)\s*;", src).group(1)
        eps = float(re.search(r"eps_t\s*=\s*([0-9.eE+-]+)\s*;", src).group(1))
        D = int(re.search(r"\bD\s*=\s*(\d+)", src).group(1))
        tol_eps = float(re.search(r"err_bnd\s*=\s*([0-9.]+)\s*\*\s*EPSILON", src).group(1))
        schemes[name] = {
            "D": D, "eps_t": eps, "tol_rel_l1": tol_eps * 2.220446049250313e-16,
            "result_exact": [c2l(z) for z in array_init(src, "result_exact")],
        }
    out["akns_fscatter"] = {
        "input_rule": {
            "q": "(0.41*cos(n) + 0.59j*sin(0.28*n))*50, n=1..D",
            "r": "(0.33*sin(n) + 0.85j*cos(0.43*n))*25, n=1..D",
            "z_args": [0.0, "pi/4", "9*pi/14", "4*pi/3", "-pi/5"],
            "layout": "result_exact[e*5+j] = S_e(z_j), e in (11,12,21,22)",
        },
        "tol_rel_l1": 100 * 2.220446049250313e-16,
        "schemes": schemes,
    }

    # ---- analytic spectra ------------------------------------------------------------------
    src = strip_comments(read("src/private/fnft__nsev_testcases.c"))

    def case_block(tag):
        # body of the LAST "case nsev_testcases_<tag>:" (the one that fills in the values)
        idx = [m.start() for m in re.finditer(r"case\s+nsev_testcases_" + tag + r"\s*:", src)][-1]
        end = src.find("break;", idx)
        return src[idx:end]

    def assigns(block, ptr):
        vals = {}
        for m in re.finditer(r"\(\*" + ptr + r"\)\[(\d+)\]\s*=\s*([^;]+);", block):
            vals[int(m.group(1))] = ceval(m.group(2))
        return [c2l(vals[i]) for i in range(len(vals))]

    blk = case_block("SECH_FOCUSING")
    out["nsev_sech_focusing"] = {
        "signal": "q[i] = 1j*3.2*sech(T0 + i*(T1-T0)/(D-1))",
        "T": [-25.0, 25.0], "XI": [-7.0 / 5.0, 8.0 / 5.0], "M": 16, "kappa": 1,
        "contspec": assigns(blk, "contspec_ptr"),
        "ab": assigns(blk, "ab_ptr"),
        # discrete spectrum, fnft__nsev_testcases.c:271-283: eigenvalues 0.7i, 1.7i, 2.7i, norming
        # constants i, -i, i, residues c_k * Gamma(2/5) / Gamma(1/5)^2 with c = -1428/25, -5236/15, -4284/11
        "bound_states": [[0.0, 0.7], [0.0, 1.7], [0.0, 2.7]],
        "normconsts": [[0.0, 1.0], [0.0, -1.0], [0.0, 1.0]],
        "residues": [[c * math.gamma(0.4) / math.gamma(0.2) ** 2, 0.0]
                     for c in (-1428.0 / 25.0, -5236.0 / 15.0, -4284.0 / 11.0)],
    }
    blk = case_block("SECH_DEFOCUSING")
    out["nsev_sech_defocusing"] = {
        "signal": "q[i] = -conj(Q/GAM*sech(t/GAM)**(1-2j*F)), Q=1, GAM=1/25, F=1.5",
        "T": [-2.0, 1.5], "XI": [-100.0, 80.0], "M": 16, "kappa": -1,
        "contspec": assigns(blk, "contspec_ptr"),
    }
    out["nsev_truncated_soliton"] = {
        "signal": "q[i] = -2*be*sech(2*be*t), q[0] *= 0.5, be=0.55",
        "T": [0.0, 15.0], "XI": [0.5, 3.0], "M": 16, "kappa": 1,
        "contspec_rule": "-1j*be/xi*(xi+1j*be)/(xi-1j*be)",
    }

    # ---- per-scheme error bounds of the integration tests ----------------------------------
    bounds = []
    d = os.path.join(REF, "test/fnft_nsev")
    for fn in sorted(os.listdir(d)):
        src = strip_comments(read("test/fnft_nsev/" + fn))
        mtc = re.search(r"nsev_testcases_(SECH_FOCUSING2|SECH_FOCUSING|SECH_DEFOCUSING|TRUNCATED_SOLITON)\b", src)
        mdisc = re.search(r"opts\.discretization\s*=\s*nse_discretization_(\w+)\s*;", src)
        mD = re.search(r"UINT\s+D\s*=\s*(\d+)\s*;", src)
        mb = re.search(r"error_bounds\s*\[\s*6\s*\]\s*=\s*\{(.*?)\}\s*;", src, flags=re.S)
        if not (mtc and mD and mb):
            continue
        vals = []
        for x in mb.group(1).split(","):
            x = x.strip()
            if not x:
                continue
            x = x.replace("INFINITY", "float('inf')").replace("FNFT_INF", "float('inf')")
            vals.append(float(eval(x, {"__builtins__": {}, "float": float}, {})))
        # second stage of each test: D doubles and the bounds are rescaled by the statements
        # between "D *= 2;" and the next harness call
        vals2 = None
        m2 = re.search(r"D\s*\*=\s*2\s*;(.*?)nsev_testcases_test_fnft", src, flags=re.S)
        if m2:
            vals2 = list(vals)
            for op in re.finditer(r"error_bounds\s*\[\s*(\w+)\s*\]\s*([*/])=\s*([0-9.eE+-]+)\s*;", m2.group(1)):
                idx, sign, fac = op.group(1), op.group(2), float(op.group(3))
                targets = range(len(vals2)) if not idx.isdigit() else [int(idx)]
                for t in targets:
                    vals2[t] = vals2[t] * fac if sign == "*" else vals2[t] / fac
        # Richardson stage (present in some files): error_bounds_RE at D, then /16 (4th order) at 2D
        vals_re = None
        mre = re.search(r"error_bounds_RE\s*\[\s*6\s*\]\s*=\s*\{(.*?)\}\s*;", src, flags=re.S)
        if mre:
            vals_re = [float(eval(x.strip(), {"__builtins__": {}, "float": float, "INFINITY": float("inf")}, {}))
                       for x in mre.group(1).split(",") if x.strip()]
        bounds.append({
            "stages": walk_stages(src),
            "error_bounds_RE": vals_re[:3] if vals_re else None,
            "file": fn, "testcase": mtc.group(1),
            "discretization": mdisc.group(1) if mdisc else "2SPLIT4B",
            "D": int(mD.group(1)), "error_bounds": vals[:3],
            "error_bounds_2D": vals2[:3] if vals2 else None,
        })
    out["nsev_error_bounds"] = bounds

    # ---- KdV: analytic spectra and the per-scheme integration tests ------------------------
    src = strip_comments(read("src/private/fnft__kdvv_testcases.c"))

    def kcase(name, nxt):
        i = src.index("case kdvv_testcases_" + name + ":", src.index("generate test case") if "generate test case" in src else 0)
        # the first occurrence of the label is in the M switch; take the one that assigns values
        cands = [m.start() for m in re.finditer(r"case kdvv_testcases_" + name + r"\s*:", src)]
        i = cands[-1]
        j = src.index(nxt, i) if nxt else len(src)
        return src[i:j]

    def kassign(block):
        vals = {}
        for m in re.finditer(r"\(\*contspec_ptr\)\[(\d+)\]\s*=\s*([^;]+);", block):
            vals[int(m.group(1))] = ceval(m.group(2))
        return [c2l(vals[i]) for i in range(len(vals))]

    out["kdvv_sech"] = {
        "signal": "u[i] = 3.2*sech(T0 + i*(T1-T0)/(D-1))**2",
        "T": [-16.0, 15.0], "XI": [-71.0 / 20.0, 79.0 / 20.0], "M": 16,
        "contspec": kassign(kcase("SECH", "case kdvv_testcases_RECT")),
    }
    out["kdvv_rect"] = {
        "signal": "t=T0+i*eps: |t|==0.5 -> 0.5, |t|<0.5 -> 1, else 0",
        "T": [-1.0, 2.0], "XI": [0.0, "15*pi/32"], "M": 16,
        "contspec": kassign(kcase("RECT", "case kdvv_testcases_NEGATIVE_RECT")),
    }
    out["kdvv_negative_rect"] = {
        "signal": "minus the rect signal",
        "T": [-1.0, 2.0], "XI": [0.0, "15*pi/32"], "M": 16,
        "contspec": kassign(kcase("NEGATIVE_RECT", "default:")),
    }
    kb = []
    d = os.path.join(REF, "test/fnft_kdvv")
    for fn in sorted(os.listdir(d)):
        s2 = strip_comments(read("test/fnft_kdvv/" + fn))
        mtc = re.search(r"kdvv_testcases_(NEGATIVE_RECT|RECT|SECH)\b", s2)
        mdisc = re.search(r"opts\.discretization\s*=\s*kdv_discretization_(\w+)\s*;", s2)
        if not mtc:
            continue
        kb.append({"file": fn, "testcase": mtc.group(1),
                   "discretization": mdisc.group(1) if mdisc else "2SPLIT8B",
                   "stages": walk_stages(s2, arr_pat=r"eb\w*|error_bounds\w*", harness="kdvv_testcases_test_fnft")})
    out["kdvv_error_bounds"] = kb

    # ---- pieces the discrete spectrum and the 4SPLIT4 front end rest on ----------------------------
    # test/fnft__misc/fnft__misc_resample_test.c:28-66: band-limited shift of a chirped sech
    src = strip_comments(read("test/fnft__misc/fnft__misc_resample_test.c"))
    mq = re.search(r"q\[i\]\s*=\s*([0-9.]+)\s*\*\s*misc_sech\(t\[i\]\)\s*\*\s*CEXP\(I\s*\*\s*([0-9.]+)\s*\*\s*t\[i\]\)", src)
    out["misc_resample"] = {
        "D": int(re.search(r"\bD\s*=\s*(\d+)", src).group(1)),
        "T0": -float(re.search(r"t\[i\]\s*=\s*-([0-9.]+)\s*\+", src).group(1)),
        "span": float(re.search(r"eps_t\s*=\s*([0-9.]+)\s*/\s*\(D-1\)", src).group(1)),
        "signal": "q[i] = amp*sech(t_i)*exp(1j*freq*t_i), exact shifted signal: the same formula at t_i + delta",
        "amp": float(mq.group(1)), "freq": float(mq.group(2)),
        "deltas": [float(x) for x in re.search(r"delta\[4\]\s*=\s*\{([^}]*)\}", src).group(1).split(",")],
        "tol_rel_l1": float(re.search(r"err\s*>\s*([0-9.eE+-]+)", src).group(1)),
    }
    # test/fnft__poly/fnft__poly_roots_fasteigen_test.c:27-44: cubic, Hausdorff distance <= 100 eps
    src = strip_comments(read("test/fnft__poly/fnft__poly_roots_fasteigen_test.c"))
    out["poly_roots_fasteigen"] = {
        "deg": 3,
        "p": [c2l(z) for z in array_init(src, "p")],
        "roots_exact": [c2l(z) for z in array_init(src, "roots_exact")],
        "tol_hausdorff": 100 * 2.220446049250313e-16,
    }
    # test/fnft__nse_scatter/fnft__nse_scatter_bound_states_test_bo.c:30-131: a, a', b of q = 3 sech(t) at 3 points
    # (the file computes its three errors against 100 eps but returns SUCCESS whatever they are, :133-140)
    src = strip_comments(read("test/fnft__nse_scatter/fnft__nse_scatter_bound_states_test_bo.c"))
    out["nse_scatter_bound_states_bo"] = {
        "D": 256, "T": [-16.0, 16.0], "signal": "q[i] = 3*sech(T0 + i*eps_t)",
        "bound_states": [c2l(z) for z in array_init(src, "bound_states")],
        "a_vals": [c2l(z) for z in array_init(src, "a_vals_exact")],
        "aprime_vals": [c2l(z) for z in array_init(src, "aprime_vals_exact")],
        "b_vals": [c2l(z) for z in array_init(src, "b_vals_exact")],
        "tol_rel_l1_stated": 100 * 2.220446049250313e-16,
        "note": "the reference test cannot fail on these bounds (it returns SUCCESS unconditionally)",
    }

    # test/fnft__nse_finvscatter/*.c + fnft__nse_finvscatter_test.inc:28-75: fscatter -> finvscatter round trip
    inc = strip_comments(read("test/fnft__nse_finvscatter/fnft__nse_finvscatter_test.inc"))
    cases = []
    d = os.path.join(REF, "test/fnft__nse_finvscatter")
    for fn in sorted(os.listdir(d)):
        if not fn.endswith(".c"):
            continue
        src = strip_comments(read("test/fnft__nse_finvscatter/" + fn))
        eps_mults = [float(x) for x in re.findall(r"=\s*([0-9.]+)\s*\*\s*FNFT_EPSILON", src)]
        cases.append({"file": fn, "kappa": int(re.search(r"kappa\s*=\s*([+-]?\d+)", src).group(1)),
                      "discretization": re.search(r"fnft_nse_discretization_(\w+)\s*;", src).group(1),
                      # with FFTW / with KissFFT where the file distinguishes; the looser one is the KissFFT build's
                      "bound_eps": max(eps_mults)})
    out["nse_finvscatter"] = {
        "D": int(re.search(r"\bD\s*=\s*(\d+)", inc).group(1)),
        "eps_t": float(re.search(r"eps_t\s*=\s*([0-9.]+)", inc).group(1)),
        "signal": "q[i] = ((i+1)/(D+1)/D)*exp(1j*i/D)",
        "cases": cases,
    }

    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_fixtures.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst, "with", len(schemes), "akns schemes,", len(bounds), "bound sets")


if __name__ == "__main__":
    sys.exit(main())
