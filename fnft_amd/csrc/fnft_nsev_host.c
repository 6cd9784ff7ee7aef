/*
 * fnft_nsev_host.c -- the C driver behind the drop-in symbols of include/fnft_amd.h section 1.
 *
 * Mirrors the argument handling of the reference's src/fnft_nsev.c:133-453 (validation order,
 * return codes, option defaults, message format of src/private/fnft__errwarn.c:28-46) and hands
 * the numerical work to the HIP shim (hip_backend.hip).  There is no CPU fallback: without a
 * usable GPU the call fails with FNFT_EC_OTHER and an error message.
 */
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/fnft_amd.h"

/* version of the interface this library stands in for (CMakeLists.txt:22-25 of the reference) */
#define FNFT_AMD_IFACE_MAJOR 0
#define FNFT_AMD_IFACE_MINOR 4
#define FNFT_AMD_IFACE_PATCH 1
#define FNFT_AMD_IFACE_SUFFIX "+amd"

/* HIP shim */
FNFT_INT fnft_amd__nsev_contspec_host(FNFT_UINT D, const FNFT_COMPLEX *q, const FNFT_REAL *T,
                                      FNFT_UINT M, FNFT_COMPLEX *contspec, const FNFT_REAL *XI,
                                      FNFT_INT kappa, int discretization, int contspec_type,
                                      FNFT_INT normalization_flag, FNFT_UINT nskip);

FNFT_INT fnft_amd__nsev_discspec_host(FNFT_UINT D, const FNFT_COMPLEX *q, const FNFT_REAL *T, int bsfilt,
                                      int bsloc, FNFT_UINT niter, FNFT_UINT Dsub, int dstype, int discretization,
                                      int richardson, FNFT_UINT *K_ptr, FNFT_COMPLEX *bound_states,
                                      FNFT_COMPLEX *normconsts_or_residues, int *warn);

/* ---- error / warning text, src/fnft_errwarn.c:28-60 ---------------------------------------- */
static FNFT_INT default_printf(const char *format, ...)
{
    va_list args;
    va_start(args, format);
    const int rc = vfprintf(stderr, format, args);
    va_end(args);
    return rc;
}

static _Thread_local fnft_printf_ptr_t printf_ptr = default_printf;

void fnft_errwarn_setprintf(fnft_printf_ptr_t p) { printf_ptr = p; }
/* WARN(msg) of src/private/fnft__errwarn.h:36-48 for the C++ side of the library (internal) */
void fnft_amd__warn(const char *msg, const char *func, int line)
{
    fnft_printf_ptr_t p = printf_ptr;
    if (p != NULL)
        p("FNFT Warning: %s\n in %s(%i)-%d.%d.%d%s\n", msg, func, line, FNFT_AMD_IFACE_MAJOR, FNFT_AMD_IFACE_MINOR,
          FNFT_AMD_IFACE_PATCH, FNFT_AMD_IFACE_SUFFIX);
}
fnft_printf_ptr_t fnft_errwarn_getprintf(void) { return printf_ptr; }

static FNFT_INT raise(FNFT_INT ec, const char *func, int line, const char *msg)
{
    fnft_printf_ptr_t p = fnft_errwarn_getprintf();
    if (p != NULL)
        p("FNFT Error: %s\n in %s(%i)-%d.%d.%d%s\n", msg, func, line, FNFT_AMD_IFACE_MAJOR,
          FNFT_AMD_IFACE_MINOR, FNFT_AMD_IFACE_PATCH, FNFT_AMD_IFACE_SUFFIX);
    return ec;
}
/* shared with fnft_kdvv_host.c */
FNFT_INT fnft_amd__raise(FNFT_INT ec, const char *func, int line, const char *msg)
{
    return raise(ec, func, line, msg);
}
#define E_INVALID_ARGUMENT(name) raise(FNFT_EC_INVALID_ARGUMENT, __func__, __LINE__, "Invalid argument " #name ".")
#define E_NOT_YET_IMPLEMENTED(name, msg) \
    raise(FNFT_EC_NOT_YET_IMPLEMENTED, __func__, __LINE__, "Not yet implemented (" #name "). " msg)
#define E_SUBROUTINE(ec) raise(-abs(ec), __func__, __LINE__, "Subroutine failure.")

/* ---- defaults, src/fnft_nsev.c:26-45 --------------------------------------------------------- */
static fnft_nsev_opts_t default_opts = {
    .bound_state_filtering = fnft_nsev_bsfilt_FULL,
    .bound_state_localization = fnft_nsev_bsloc_SUBSAMPLE_AND_REFINE,
    .niter = 10,
    .Dsub = 0,
    .discspec_type = fnft_nsev_dstype_NORMING_CONSTANTS,
    .contspec_type = fnft_nsev_cstype_REFLECTION_COEFFICIENT,
    .normalization_flag = 1,
    .discretization = fnft_nse_discretization_2SPLIT4B,
    .richardson_extrapolation_flag = 0};

fnft_nsev_opts_t fnft_nsev_default_opts(void) { return default_opts; }

/* degree of one step, src/private/fnft__akns_discretization.c:29-67 via nse->akns mapping;
 * 0 for the slow (non-polynomial) discretizations */
static FNFT_UINT nse_degree(fnft_nse_discretization_t d)
{
    switch (d) {
    case fnft_nse_discretization_2SPLIT2_MODAL:
    case fnft_nse_discretization_2SPLIT1A:
    case fnft_nse_discretization_2SPLIT1B:
    case fnft_nse_discretization_2SPLIT2A:
    case fnft_nse_discretization_2SPLIT2B:
    case fnft_nse_discretization_2SPLIT2S: return 1;
    case fnft_nse_discretization_2SPLIT3S:
    case fnft_nse_discretization_2SPLIT4B:
    case fnft_nse_discretization_4SPLIT4B: return 2;
    case fnft_nse_discretization_2SPLIT3A:
    case fnft_nse_discretization_2SPLIT3B: return 3;
    case fnft_nse_discretization_2SPLIT4A:
    case fnft_nse_discretization_4SPLIT4A: return 4;
    case fnft_nse_discretization_2SPLIT6B: return 6;
    case fnft_nse_discretization_2SPLIT6A:
    case fnft_nse_discretization_2SPLIT8B: return 12;
    case fnft_nse_discretization_2SPLIT5A:
    case fnft_nse_discretization_2SPLIT5B: return 15;
    case fnft_nse_discretization_2SPLIT8A: return 24;
    case fnft_nse_discretization_2SPLIT7A:
    case fnft_nse_discretization_2SPLIT7B: return 105;
    default: return 0;
    }
}

/* shared with fnft_nsev_inverse_host.c */
FNFT_UINT fnft_amd__nse_degree(fnft_nse_discretization_t d) { return nse_degree(d); }

/* src/fnft_nsev.c:51-57 */
FNFT_UINT fnft_nsev_max_K(const FNFT_UINT D, fnft_nsev_opts_t const *const opts)
{
    if (opts != NULL) return nse_degree(opts->discretization) * D;
    return nse_degree(default_opts.discretization) * D;
}

/* include/fnft_nsev.h:371-376, src/fnft_nsev.c:133-453 */
FNFT_INT fnft_nsev(const FNFT_UINT D, FNFT_COMPLEX *const q, FNFT_REAL const *const T,
                   const FNFT_UINT M, FNFT_COMPLEX *const contspec, FNFT_REAL const *const XI,
                   FNFT_UINT *const K_ptr, FNFT_COMPLEX *const bound_states,
                   FNFT_COMPLEX *const normconsts_or_residues, const FNFT_INT kappa,
                   fnft_nsev_opts_t *opts)
{
    FNFT_INT ret_code = FNFT_SUCCESS;

    /* same checks, same order as src/fnft_nsev.c:163-180 */
    if (D < 2) return E_INVALID_ARGUMENT(D);
    if (q == NULL) return E_INVALID_ARGUMENT(q);
    if (T == NULL || T[0] >= T[1]) return E_INVALID_ARGUMENT(T);
    if (contspec != NULL) {
        if (XI == NULL || XI[0] >= XI[1]) return E_INVALID_ARGUMENT(XI);
    }
    if (abs(kappa) != 1) return E_INVALID_ARGUMENT(kappa);
    if (bound_states != NULL) {
        if (K_ptr == NULL) return E_INVALID_ARGUMENT(K_ptr);
    }
    if (opts == NULL) opts = &default_opts;

    /* src/fnft_nsev.c:183-220 */
    switch (opts->discretization) {
    case fnft_nse_discretization_2SPLIT2_MODAL:
    case fnft_nse_discretization_2SPLIT1A:
    case fnft_nse_discretization_2SPLIT1B:
    case fnft_nse_discretization_2SPLIT2A:
    case fnft_nse_discretization_2SPLIT2B:
    case fnft_nse_discretization_2SPLIT2S:
    case fnft_nse_discretization_2SPLIT3S:
    case fnft_nse_discretization_2SPLIT4B:
    case fnft_nse_discretization_2SPLIT3A:
    case fnft_nse_discretization_2SPLIT3B:
    case fnft_nse_discretization_2SPLIT4A:
    case fnft_nse_discretization_2SPLIT6B:
    case fnft_nse_discretization_2SPLIT6A:
    case fnft_nse_discretization_2SPLIT8B:
    case fnft_nse_discretization_2SPLIT5A:
    case fnft_nse_discretization_2SPLIT5B:
    case fnft_nse_discretization_2SPLIT8A:
    case fnft_nse_discretization_2SPLIT7A:
    case fnft_nse_discretization_2SPLIT7B:
    case fnft_nse_discretization_4SPLIT4A:
    case fnft_nse_discretization_4SPLIT4B:
        break;
    case fnft_nse_discretization_BO:
    case fnft_nse_discretization_CF4_2:
    case fnft_nse_discretization_CF4_3:
    case fnft_nse_discretization_CF5_3:
    case fnft_nse_discretization_CF6_4:
    case fnft_nse_discretization_ES4:
    case fnft_nse_discretization_TES4:
        if (opts->bound_state_localization != fnft_nsev_bsloc_NEWTON && kappa == +1)
            return E_INVALID_ARGUMENT(opts->bound_state_localization);
        break;
    default:
        return E_INVALID_ARGUMENT(opts->discretization);
    }

    /* what this build does not cover yet is reported, never silently approximated */
    switch (opts->discretization) {
    case fnft_nse_discretization_2SPLIT2_MODAL:
    case fnft_nse_discretization_2SPLIT1A:
    case fnft_nse_discretization_2SPLIT1B:
    case fnft_nse_discretization_2SPLIT2A:
    case fnft_nse_discretization_2SPLIT2B:
    case fnft_nse_discretization_2SPLIT2S:
    case fnft_nse_discretization_2SPLIT3A:
    case fnft_nse_discretization_2SPLIT3B:
    case fnft_nse_discretization_2SPLIT3S:
    case fnft_nse_discretization_2SPLIT4A:
    case fnft_nse_discretization_2SPLIT4B:
    case fnft_nse_discretization_2SPLIT5A:
    case fnft_nse_discretization_2SPLIT5B:
    case fnft_nse_discretization_2SPLIT6A:
    case fnft_nse_discretization_2SPLIT6B:
    case fnft_nse_discretization_2SPLIT7A:
    case fnft_nse_discretization_2SPLIT7B:
    case fnft_nse_discretization_2SPLIT8A:
    case fnft_nse_discretization_2SPLIT8B:
    case fnft_nse_discretization_4SPLIT4A:
    case fnft_nse_discretization_4SPLIT4B:
        break;
    default:
        return E_NOT_YET_IMPLEMENTED(discretization,
                                     "GPU path covers the fast (polynomial) discretizations.");
    }
    if (kappa == +1 && bound_states != NULL) {
        /* option values that nsev_compute_boundstates / _normconsts_or_residues reject,
         * src/fnft_nsev.c:717-719, :957-958 (wrapped once by CHECK_RETCODE in fnft_nsev_base, once in fnft_nsev) */
        const int loc = (int)opts->bound_state_localization, dst = (int)opts->discspec_type;
        if (loc < 0 || loc > 2) return E_INVALID_ARGUMENT(opts->bound_state_localization);
        if (normconsts_or_residues != NULL && (dst < 0 || dst > 2)) {
            ret_code = E_INVALID_ARGUMENT(opts->discspec_type);
            ret_code = E_SUBROUTINE(ret_code);
            return E_SUBROUTINE(ret_code);
        }
    }
    if (contspec != NULL && M > 0) {
        const int cst = (int)opts->contspec_type;
        if (cst < 0 || cst > 2) { /* src/fnft_nsev.c:880-883, raised inside nsev_compute_contspec */
            ret_code = E_INVALID_ARGUMENT(opts->contspec_type);
            ret_code = E_SUBROUTINE(ret_code);
            return E_SUBROUTINE(ret_code);
        }
    }

    ret_code = fnft_amd__nsev_contspec_host(D, q, T, (contspec != NULL) ? M : 0, contspec, XI, kappa,
                                            (int)opts->discretization, (int)opts->contspec_type,
                                            opts->normalization_flag, 1);
    if (ret_code != FNFT_SUCCESS) {
        if (ret_code == FNFT_EC_OTHER || ret_code == FNFT_EC_NOMEM)
            return raise(ret_code, __func__, __LINE__, "GPU runtime failure (see fnft_amd_last_error()).");
        return E_SUBROUTINE(ret_code);
    }

    /* Richardson extrapolation of the continuous spectrum, src/fnft_nsev.c:316-406: second
     * transform of every other sample (subsampling rule of
     * src/private/fnft__nse_discretization.c:426-473), method order 2 for the 2SPLIT schemes and
     * 4 for 4SPLIT4A/B (src/private/fnft__akns_discretization.c:157-192).  For 4SPLIT4A/B the coarse
     * transform still resamples the full signal (:474-503), so the device plan gets all D samples
     * and the skip count. */
    if (opts->richardson_extrapolation_flag == 1 && contspec != NULL && M > 0) {
        const FNFT_REAL eps_t = (T[1] - T[0]) / (D - 1);
        const int cst = (int)opts->contspec_type;
        const FNFT_UINT cs_len = M * (cst == 0 ? 1 : (cst == 1 ? 2 : 3));
        FNFT_UINT Dsub = D / 2; /* CEIL(D/2) on integers, :376 */
        if (Dsub < 2) Dsub = 2;
        if (Dsub > D) Dsub = D;
        const FNFT_UINT nskip = (FNFT_UINT)round((FNFT_REAL)D / Dsub);
        Dsub = (FNFT_UINT)round((FNFT_REAL)D / nskip);
        FNFT_COMPLEX *qsub = malloc(Dsub * sizeof(FNFT_COMPLEX));
        FNFT_COMPLEX *csub = malloc(cs_len * sizeof(FNFT_COMPLEX));
        if (qsub == NULL || csub == NULL) {
            free(qsub);
            free(csub);
            return raise(FNFT_EC_NOMEM, __func__, __LINE__, "Out of memory.");
        }
        for (FNFT_UINT i = 0; i < Dsub; i++) qsub[i] = q[i * nskip];
        const FNFT_REAL Tsub[2] = {T[0], T[0] + ((Dsub - 1) * nskip) * eps_t};
        const FNFT_REAL eps_t_sub = (Tsub[1] - Tsub[0]) / (Dsub - 1);
        const int four = opts->discretization == fnft_nse_discretization_4SPLIT4A
                         || opts->discretization == fnft_nse_discretization_4SPLIT4B;
        if (four)
            ret_code = fnft_amd__nsev_contspec_host(D, q, T, M, csub, XI, kappa, (int)opts->discretization,
                                                    cst, opts->normalization_flag, nskip);
        else
            ret_code = fnft_amd__nsev_contspec_host(Dsub, qsub, Tsub, M, csub, XI, kappa,
                                                    (int)opts->discretization, cst,
                                                    opts->normalization_flag, 1);
        if (ret_code == FNFT_SUCCESS) {
            const FNFT_REAL scl_num = pow(eps_t_sub / eps_t, four ? 4.0 : 2.0);
            const FNFT_REAL scl_den = scl_num - 1.0;
            const FNFT_REAL dxi = (XI[1] - XI[0]) / (M - 1);
            const FNFT_REAL pi = acos(-1.0);
            for (FNFT_UINT i = 0; i < M; i++)
                if (fabs(XI[0] + dxi * i) < 0.9 * pi / (2.0 * eps_t_sub))
                    for (FNFT_UINT j = 0; j < cs_len; j += M)
                        contspec[i + j] = (scl_num * contspec[i + j] - csub[i + j]) / scl_den;
        }
        free(qsub);
        free(csub);
        if (ret_code != FNFT_SUCCESS) {
            if (ret_code == FNFT_EC_OTHER || ret_code == FNFT_EC_NOMEM)
                return raise(ret_code, __func__, __LINE__, "GPU runtime failure (see fnft_amd_last_error()).");
            return E_SUBROUTINE(ret_code);
        }
    }
    /* discrete spectrum, src/fnft_nsev.c:276-309 (+ :406-441 with Richardson), :545-560 */
    if (kappa == +1 && bound_states != NULL) {
        int warn = 0;
        ret_code = fnft_amd__nsev_discspec_host(D, q, T, (int)opts->bound_state_filtering,
                                                (int)opts->bound_state_localization, opts->niter, opts->Dsub,
                                                (int)opts->discspec_type, (int)opts->discretization,
                                                opts->richardson_extrapolation_flag == 1, K_ptr, bound_states,
                                                normconsts_or_residues, &warn);
        if (warn & 1) {
            fnft_printf_ptr_t p = fnft_errwarn_getprintf();
            if (p != NULL)
                p("FNFT Warning: %s\n in %s(%i)-%d.%d.%d%s\n",
                  "Found more than *K_ptr bound states. Returning as many as possible.", __func__, __LINE__,
                  FNFT_AMD_IFACE_MAJOR, FNFT_AMD_IFACE_MINOR, FNFT_AMD_IFACE_PATCH, FNFT_AMD_IFACE_SUFFIX);
        }
        if (warn & 2) {
            fnft_printf_ptr_t p = fnft_errwarn_getprintf();
            if (p != NULL)
                p("FNFT Warning: %s\n in %s(%i)-%d.%d.%d%s\n",
                  "Root finder stopped at its iteration limit; bound states may be inaccurate (clustered roots?).",
                  __func__, __LINE__,
                  FNFT_AMD_IFACE_MAJOR, FNFT_AMD_IFACE_MINOR, FNFT_AMD_IFACE_PATCH, FNFT_AMD_IFACE_SUFFIX);
        }
        if (ret_code != FNFT_SUCCESS) {
            if (ret_code == FNFT_EC_OTHER || ret_code == FNFT_EC_NOMEM)
                return raise(ret_code, __func__, __LINE__, "GPU runtime failure (see fnft_amd_last_error()).");
            return E_SUBROUTINE(ret_code);
        }
    } else if (K_ptr != NULL) {
        *K_ptr = 0; /* src/fnft_nsev.c:558-560 */
    }
    return FNFT_SUCCESS;
}
