"""Extracts the DATA the reference's fnft_nsev_inverse tests hold (numbers only) into
tests/golden/inverse_sech_defocusing.npz, and their error bounds / sizes into tests/golden/inverse_fixtures.json.

    python tests/golden/extract_inverse_fixtures.py     (needs /root/reference; the outputs are committed)

Sources (all under test/fnft_nsev_inverse/):
  fnft_nsev_inverse_test_sech_defocusing/*_data_2048.inc, *_data_4096.inc  -- M, T, XI, q_exact[], contspec[]
  every *.c driver                                                          -- discretization, option values, D / M
                                                                               schedule and error bounds of main()
"""
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference/test/fnft_nsev_inverse"
HERE = os.path.dirname(os.path.abspath(__file__))
NUM = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|inf|nan)"


def cplx_array(src, name):
    m = re.search(r"(?:const\s+)?COMPLEX\s+%s\[\d+\]\s*=\s*\{(.*?)\};" % name, src, re.S)
    vals = re.findall(r"(%s)\s*\+\s*(%s)\s*\*\s*I" % (NUM, NUM), m.group(1))
    return np.array([complex(float(a), float(b)) for a, b in vals])


def real_array(src, name):
    m = re.search(r"const\s+REAL\s+%s\[\d+\]\s*=\s*\{(.*?)\};" % name, src, re.S)
    return np.array([float(x) for x in re.findall(NUM, m.group(1))])


def main():
    out = {}
    d = os.path.join(REF, "fnft_nsev_inverse_test_sech_defocusing")
    for n in (2048, 4096):
        src = open(os.path.join(d, "fnft_nsev_inverse_test_sech_defocusing_data_%d.inc" % n)).read()
        out["M_%d" % n] = np.array(int(re.search(r"const\s+UINT\s+M_%d\s*=\s*(\d+)" % n, src).group(1)))
        out["T_%d" % n] = real_array(src, "T_%d" % n)
        out["XI_%d" % n] = real_array(src, "XI_%d" % n)
        out["q_exact_%d" % n] = cplx_array(src, "q_exact_%d" % n)
        out["contspec_%d" % n] = cplx_array(src, "contspec_%d" % n)
        assert out["q_exact_%d" % n].size == n and out["contspec_%d" % n].size == int(out["M_%d" % n])
    np.savez_compressed(os.path.join(HERE, "inverse_sech_defocusing.npz"), **out)

    # drivers: the option assignments and the (D, M, error_bound) schedule of main(), as written in each file
    drivers = {}
    for root, _dirs, files in os.walk(REF):
        for f in sorted(files):
            if not f.endswith(".c"):
                continue
            src = open(os.path.join(root, f)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            src = re.sub(r"//[^\n]*", "", src)
            main_src = src[src.index("main("):]
            ent = {"opts": [], "bounds": []}
            for m in re.finditer(r"opts\w*\.(\w+)\s*=\s*(?:\w*?_)?([A-Za-z0-9_]+)\s*;", main_src):
                ent["opts"].append([m.start(), m.group(1), m.group(2)])
            for m in re.finditer(r"\b(D|M|error_bound)\s*=\s*([^;]+);", main_src):
                ent["bounds"].append([m.start(), m.group(1), m.group(2).strip()])
            drivers[f] = ent
    # fnft__nse_scatter_matrix known answers (test/fnft__nse_scatter/fnft__nse_scatter_matrix_test_*_bo.c): the 16 values
    # of result_exact (two lambdas x [S11 S12 S21 S22 S11' S12' S21' S22']); inputs are formulas, restated in the tests
    sm = {}
    for name, kappa in (("focusing", 1), ("defocusing", -1)):
        src = open("/root/reference/test/fnft__nse_scatter/fnft__nse_scatter_matrix_test_%s_bo.c" % name).read()
        body = re.search(r"result_exact\[16\]\s*=\s*\{(.*?)\};", src, re.S).group(1)
        vals = re.findall(r"([-+]?\s*\d[\d.]*(?:[eE][-+]?\d+)?)\s*([-+])\s*(\d[\d.]*(?:[eE][-+]?\d+)?)\s*\*\s*I", body)
        assert len(vals) == 16, len(vals)
        sm[name] = {"kappa": kappa, "result_exact": [[float(a.replace(" ", "")), float(sg + b)] for a, sg, b in vals]}
    drivers["__nse_scatter_matrix__"] = sm
    json.dump(drivers, open(os.path.join(HERE, "inverse_fixtures.json"), "w"), indent=0)
    print("wrote inverse_sech_defocusing.npz and inverse_fixtures.json (%d drivers)" % len(drivers))


if __name__ == "__main__":
    sys.exit(main())
