"""The reference's fnft_nsev_inverse tests (test/fnft_nsev_inverse/*) as data: every case is a dict with the inputs the
test's C code forms (restated from its formulas), the exact signal it compares with and its error bound (read from
tests/golden/inverse_fixtures.json, which tests/golden/extract_inverse_fixtures.py wrote from the files' main()).
Used by tests/test_inverse_oracle.py (CPU, oracle) and tests/test_gpu_inverse.py (GPU, C ABI)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "inverse_fixtures.json")))
DATA = np.load(os.path.join(HERE, "golden", "inverse_sech_defocusing.npz"))
DISCS = {"2split2A": "2SPLIT2A", "2split2_modal": "2SPLIT2_MODAL"}


def sech(x):
    return 1.0 / np.cosh(x)


def tgrid(T, D):
    return T[0] + np.arange(D) * (T[1] - T[0]) / (D - 1)


def first_bound(fname):
    return float([b for b in FIX[fname]["bounds"] if b[1] == "error_bound"][0][2])


def bounds(fname):
    return [float(b[2]) for b in FIX[fname]["bounds"] if b[1] == "error_bound" and "error_bound" not in b[2]]


def sech_defocusing(tag, n):
    f = "fnft_nsev_inverse_test_sech_defocusing_%s.c" % tag
    return dict(M=int(DATA["M_%d" % n]), contspec=DATA["contspec_%d" % n].copy(), XI=list(DATA["XI_%d" % n]), D=n,
                T=list(DATA["T_%d" % n]), kappa=-1, q_exact=DATA["q_exact_%d" % n],
                bound=first_bound(f) / (1 if n == 2048 else 4),
                opts=dict(discretization=DISCS[tag], contspec_inversion_method="TFMATRIX_CONTAINS_REFL_COEFF"))


def truncated_soliton(tag, D, XI_of):
    f = "fnft_nsev_inverse_test_truncated_soliton_%s.c" % tag
    M, T = 4 * D, [-2.0, 2.0]
    al, be = 2.0, 0.55
    gam = np.sqrt(al * al + be * be)
    XI = XI_of(D, T, M)
    xi = XI[0] + (XI[1] - XI[0]) / (M - 1) * np.arange(M)
    t = tgrid(T, D)
    q_exact = np.where(t <= 0, -(2j * gam * al / abs(al)) * sech(2 * gam * t + np.arctanh(be / gam)), 0.0)
    return dict(M=M, contspec=al / (xi - 1j * be), XI=XI, D=D, T=T, kappa=1, q_exact=q_exact,
                bound_states=np.array([1j * be]), normconsts=np.array([-1j * al / (gam + be)]),
                bound=first_bound(f) / (1 if D == 512 else 2),
                opts=dict(discretization=DISCS[tag], discspec_type="NORMING_CONSTANTS"))


def b_cases(kind, discrete, tag, step, XI_of):
    """B_of_tau / b_of_xi (with and without the discrete spectrum): D doubles four times, the bound falls by 4."""
    f = "fnft_nsev_inverse_test_%s%s_%s.c" % (kind, "_w_discrete" if discrete else "", tag)
    D = (512 if discrete else 256) << step
    T = [-25.0, 25.0]
    A, t0 = (3.45, 0.0) if discrete else (0.45, 1.2)
    t = tgrid(T, D)
    case = dict(M=D, D=D, T=T, kappa=1, q_exact=1j * A * sech(t - t0), bound=first_bound(f) / 4 ** step,
                opts=dict(discretization=DISCS[tag], contspec_type="B_OF_TAU" if kind == "B_of_tau" else "B_OF_XI"))
    if kind == "B_of_tau":
        case["XI"] = [-1.0, 1.0]
        case["contspec"] = 1j / (2 * np.pi) * np.sin(np.pi * A) * sech((2 * t - 2 * t0) / 2)
    else:
        XI = XI_of(D, T, D)
        xi = XI[0] + (XI[1] - XI[0]) / (D - 1) * np.arange(D)
        case["XI"] = XI
        case["contspec"] = 1j * np.exp(-2j * xi * t0) * np.sin(np.pi * A) / np.cosh(np.pi * xi)
    if discrete:
        K = int(np.floor(A + 0.5))
        i = np.arange(K)
        case["bound_states"] = 1j * ((A + 0.5) - (i + 1))
        case["normconsts"] = -1j * (-1.0) ** (i + 1)
    return case


def against_forward(sign, tag, idx):
    """D = 8; (M, bound, method) number idx of the file's main()."""
    f = "fnft_nsev_inverse_test_against_forward_%s_%s.c" % (sign, tag)
    Ms = [int(b[2]) for b in FIX[f]["bounds"] if b[1] == "M"]
    bs = bounds(f)
    method = "TFMATRIX_CONTAINS_REFL_COEFF" if idx < 2 else "TFMATRIX_CONTAINS_AB_FROM_ITER"
    q_exact = np.array([0.1, 0.1j, 0.2, -0.2, 0.0, 0.05 + 0.05j, -0.03j, 0.06], np.complex128)
    return dict(M=Ms[idx], D=8, T=[0.0, 7.0], kappa=1 if sign == "focusing" else -1, q_exact=q_exact, bound=bs[idx],
                forward="2SPLIT2A" if tag == "2split2A" else "2SPLIT2_MODAL",
                opts=dict(discretization=DISCS[tag], contspec_inversion_method=method))


def n_against_forward(sign, tag):
    f = "fnft_nsev_inverse_test_against_forward_%s_%s.c" % (sign, tag)
    return len([b for b in FIX[f]["bounds"] if b[1] == "M"])


def against_forward_w_discrete(tag, dstype, D):
    f = "fnft_nsev_inverse_test_against_forward_w_discrete_%s.c" % tag
    T = [-32.0, 32.0]
    t = tgrid(T, D)
    return dict(M=2 * D, D=D, T=T, kappa=1, q_exact=3.4 * sech(t) * np.exp(-4j * t),
                bound=first_bound(f) / (1 if D == 512 else 4), forward_ds=dstype,
                opts=dict(discretization=DISCS[tag], discspec_type=dstype))


def _nc_to_residues(bs, nc):
    out = np.array(nc, np.complex128)
    for i in range(len(bs)):
        tmp = 1.0 + 0j
        for j in range(len(bs)):
            if j != i:
                tmp = tmp * (bs[i] - bs[j]) / (bs[i] - np.conj(bs[j]))
        out[i] = nc[i] * (2j * bs[i].imag) / tmp
    return out


def addsoliton_cdt(D, dstype):
    T = [-20.0, 20.0]
    t = tgrid(T, D)
    bs = np.array([2.5 + 0.9j, 2.5 + 1.9j, 2.5 + 2.9j])
    nc = np.array([-1.0, 1.0, -1.0], np.complex128)
    return dict(M=0, contspec=None, XI=None, D=D, T=T, kappa=1, q_exact=3.4 * sech(t) * np.exp(-5j * t),
                q_seed=-0.4 * sech(t) * np.exp(-5j * t), bound_states=bs,
                normconsts=nc if dstype == "NORMING_CONSTANTS" else _nc_to_residues(bs, nc),
                bound=first_bound("fnft_nsev_inverse_test_addsoliton_cdt.c") / (1 if D == 512 else 4),
                opts=dict(discspec_type=dstype, contspec_inversion_method="USE_SEED_POTENTIAL_INSTEAD"))


def multisoliton_cdt(dstype, D=16384):
    T = [-15.0, 15.0]
    bs = 1j * np.array([0.5, 1.5, 2.5, 3.5, 4.5])
    nc = np.array([-1.0, 1.0, -1.0, 1.0, -1.0], np.complex128)
    return dict(M=0, contspec=None, XI=None, D=D, T=T, kappa=1, q_exact=5.0 * sech(tgrid(T, D)), bound_states=bs,
                normconsts=nc if dstype == "NORMING_CONSTANTS" else _nc_to_residues(bs, nc),
                bound=100 * np.finfo(float).eps, opts=dict(discspec_type=dstype))


# ---- fnft__poly_roots_fftgridsearch known answers (test/fnft__poly/fnft__poly_roots_fftgridsearch_test_*.c) --------------
# the polynomials and their exact unit-circle roots as the files give them; (M, bounds) schedules of the files' main()
GRID_P_EVEN = np.array([4 - 5j, 3 - 4j, 2 - 3j, 2, 2 + 3j, 3 + 4j, 4 + 5j])
GRID_ROOTS_EVEN = np.array([3.992603696776205e-01 + 9.168375849652399e-01j, -3.932716698145485e-01 + 9.194223152182454e-01j,
                            9.475398776668446e-01 - 3.196375763753407e-01j, -9.853253052543915e-01 + 1.706869732151292e-01j,
                            -7.486910771535749e-01 - 6.629190531208312e-01j, 4.384974884943283e-17 - 1.000000000000000e+00j])
GRID_P_ODD = np.array([1, -1.58378511059697 + 2.52620897565978j, 0.46009879991909 + 1.20456190643706j,
                       3.23268341994846 + 1.59679613801836j])
GRID_ROOTS_ODD = np.array([0.504846104599857 + 0.863209366648873j, -0.921060994002885 - 0.38941834230865j])
GRID_EVEN = [(128, 0.0014, 4.8e-6), (256, 0.0014 / 4, 4.8e-6 / 3.6)]       # deg_even: (M, eb1, eb2)
GRID_ODD = [(128, 0.0003), (256, 0.0003 / 4)]                               # deg_odd: (M, eb)
GRID_PH = [(128, 6.2e-4, 3.6e-6), (256, 6.2e-4 / 4, 3.6e-6 / 4)]              # paraherm: (M, eb1, eb2)
