#!/usr/bin/env python3
"""Extract the reference's own fnft_nsep tests (test/fnft_nsep/*.c, 15 files) into fixtures.

Run once in the build container (needs /root/reference); the outputs tests/golden/nsep_fixtures.json and
tests/golden/nsep_fixtures.npz are committed and are all the tests read.  Only NUMBERS and option values are extracted
-- no source text is kept:
  * the 10 analytic files (plane wave, focusing / constant signal, defocusing; src/private/fnft__nsep_testcases.c holds
    the closed forms, restated in tests/nsep_cases.py): per harness call nsep_testcases_test_fnft(tc, D, error_bounds,
    &opts) the values of D, the three bounds and the options at that point, replayed statement by statement;
  * the 5 numerical files: the sample arrays q, T, the expected main / auxiliary spectrum (or spine points), the
    options and the file's own distance bounds.
"""
import json
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from extract_reference_fixtures import REF, ceval, read, strip_comments   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DIR = "test/fnft_nsep"


def c_array(src, name):
    """complex values of  name[...] = { ... };"""
    m = re.search(re.escape(name) + r"\s*\[[^\]]*\]\s*=\s*\{(.*?)\}\s*;", src, flags=re.S)
    if not m:
        return None
    return np.array([ceval(x) for x in m.group(1).split(",") if x.strip()], np.complex128)


def opts_default():
    return {"localization": "MIXED", "filtering": "AUTO", "max_evals": 20, "bounding_box": [None, None, None, None],
            "normalization_flag": 1, "discretization": "2SPLIT2A", "floquet_range": [-1.0, 1.0], "points_per_spine": 2,
            "Dsub": 0, "tol": -1.0}


def num(expr, env):
    e = expr.replace("(REAL)", "").replace("PI", "3.14159265358979323846264338327950288").strip()
    return float(eval(e, {"__builtins__": {}}, dict(env)))


def walk_analytic(src):
    """{testcase, stages: [{D, bounds, opts}]} of one analytic file."""
    tc = re.search(r"nsep_testcases_(\w+)\s*;", src).group(1)
    env, bounds, opts, stages = {}, None, opts_default(), []
    tok = re.compile(
        r"(?P<arr>REAL\s+error_bounds\s*\[\s*3\s*\]\s*=\s*\{(?P<av>.*?)\}\s*;)"
        r"|(?P<decl>UINT\s+D\s*=\s*(?P<dv>[^;]+);)"
        r"|(?P<loop>for\s*\([^)]*\)\s*error_bounds\s*\[\s*i\s*\]\s*(?P<lo>[*/])=\s*(?P<lv>[^;]+);)"
        r"|(?P<one>error_bounds\s*\[\s*(?P<oi>\d)\s*\]\s*(?P<oo>[*/]?)=\s*(?P<ov>[^;]+);)"
        r"|(?P<dop>\bD\s*(?P<do>[*/])=\s*(?P<dval>[^;]+);)"
        r"|(?P<dset>\bD\s*=\s*(?P<dsv>[^;=]+);)"
        r"|(?P<bb>opts\.bounding_box\s*\[\s*(?P<bi>\d)\s*\]\s*=\s*(?P<bv>[^;]+);)"
        r"|(?P<disc>opts\.discretization\s*=\s*\w*?discretization_(?P<dn>\w+)\s*;)"
        r"|(?P<loc>opts\.localization\s*=\s*fnft_nsep_loc_(?P<ln>\w+)\s*;)"
        r"|(?P<filt>opts\.filtering\s*=\s*fnft_nsep_filt_(?P<fn>\w+)\s*;)"
        r"|(?P<oth>opts\.(?P<on>max_evals|Dsub|points_per_spine|tol|normalization_flag)\s*=\s*(?P<otv>[^;]+);)"
        r"|(?P<call>nsep_testcases_test_fnft\s*\(\s*tc\s*,\s*D\s*,\s*error_bounds\s*,)", flags=re.S)
    for m in tok.finditer(src[src.index("main"):]):
        if m.group("arr"):
            bounds = [num(x, env) for x in m.group("av").split(",") if x.strip()]
        elif m.group("decl"):
            env["D"] = int(num(m.group("dv"), env))
        elif m.group("loop"):
            f = num(m.group("lv"), env)
            bounds = [b * f if m.group("lo") == "*" else b / f for b in bounds]
        elif m.group("one"):
            i, f, op = int(m.group("oi")), num(m.group("ov"), env), m.group("oo")
            bounds[i] = f if op == "" else (bounds[i] * f if op == "*" else bounds[i] / f)
        elif m.group("dop"):
            v = num(m.group("dval"), env)
            env["D"] = int(env["D"] * v if m.group("do") == "*" else env["D"] / v)
        elif m.group("dset"):
            env["D"] = int(num(m.group("dsv"), env))
        elif m.group("bb"):
            opts["bounding_box"][int(m.group("bi"))] = num(m.group("bv"), env)
        elif m.group("disc"):
            opts["discretization"] = m.group("dn")
        elif m.group("loc"):
            opts["localization"] = m.group("ln")
        elif m.group("filt"):
            opts["filtering"] = m.group("fn")
        elif m.group("oth"):
            v = num(m.group("otv"), env)
            opts[m.group("on")] = v if m.group("on") == "tol" else int(v)
        elif m.group("call"):
            stages.append({"D": env["D"], "bounds": list(bounds), "opts": json.loads(json.dumps(opts))})
    return {"testcase": tc, "stages": stages}


def numeric_file(name, src, arrays):
    """parameters of one numerical file; its arrays go to the npz under '<name>/<array>'."""
    rec = {}
    for arr in ("q", "mainspec_exact", "auxspec_exact", "spines_exact"):
        a = c_array(src, arr)
        if a is not None:
            arrays["%s/%s" % (name, arr)] = a
    m = re.search(r"const\s+REAL\s+T\s*\[\s*2\s*\]\s*=\s*\{([^}]*)\}", src)
    rec["T"] = [num(x, {}) for x in m.group(1).split(",")]
    opts = opts_default()
    for mm in re.finditer(r"opts\.bounding_box\s*\[\s*(\d)\s*\]\s*=\s*([^;]+);", src):
        opts["bounding_box"][int(mm.group(1))] = num(mm.group(2), {})
    mm = re.search(r"opts\.filtering\s*=\s*fnft_nsep_filt_(\w+)", src)
    if mm:
        opts["filtering"] = mm.group(1)
    rec["opts"] = opts
    mm = re.search(r"UINT\s+points_per_spine\s*=\s*(\d+)", src)
    rec["points_per_spine"] = int(mm.group(1)) if mm else None
    rec["dist_bounds"] = [num(x, {}) for x in re.findall(r"dist\s*>\s*([0-9.eE+-]+)", src)]
    mm = re.search(r"const\s+REAL\s+tol\s*=\s*([0-9.eE+-]+)", src)
    rec["spine_tol"] = float(mm.group(1)) if mm else None
    mm = re.search(r"FABS\(CREAL\(lam\)\)\s*>\s*(\d+)\s*\*\s*EPSILON", src)
    rec["spine_real_eps"] = int(mm.group(1)) if mm else None
    mm = re.search(r"fnft_nsep\([^;]*?,\s*([+-]1)\s*,\s*&opts\)", src)
    rec["kappa"] = int(mm.group(1))
    return rec


def main():
    out = {"_generated_by": "tests/golden/extract_nsep_fixtures.py", "analytic": {}, "numeric": {}}
    arrays = {}
    for f in sorted(os.listdir(os.path.join(REF, DIR))):
        if not f.endswith(".c"):
            continue
        src = strip_comments(read(os.path.join(DIR, f)))
        name = f.replace("fnft_nsep_test_", "").replace(".c", "")
        if "nsep_testcases_test_fnft" in src:
            out["analytic"][name] = walk_analytic(src)
        else:
            out["numeric"][name] = numeric_file(name, src, arrays)
    # nonregression_1 builds its signal from a formula (test/fnft_nsep/fnft_nsep_test_nonregression_1.c:545-547):
    # q[i] = 1 + 0.22 exp(-0.822 i t_i), T = [0, 2 pi/0.822], D = 512, K = 5 * points_per_spine; restated in the tests
    with open(os.path.join(HERE, "nsep_fixtures.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "nsep_fixtures.npz"), **arrays)
    print(len(out["analytic"]), "analytic files,", len(out["numeric"]), "numerical files,", len(arrays), "arrays")
    for k, v in out["analytic"].items():
        print(" ", k, v["testcase"], [(s["D"], s["bounds"][:2]) for s in v["stages"]])
    for k, v in out["numeric"].items():
        print(" ", k, {kk: vv for kk, vv in v.items() if kk != "opts"})


if __name__ == "__main__":
    main()
