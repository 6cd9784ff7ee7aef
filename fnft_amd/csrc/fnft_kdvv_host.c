/*
 * fnft_kdvv_host.c -- C driver behind fnft_kdvv (include/fnft_amd.h section 4).
 *
 * Mirrors the argument handling of the reference's src/fnft_kdvv.c:59-123 (validation order,
 * return codes, default options) and hands the numerical work to the HIP shim.  No CPU fallback.
 */
#include <stdlib.h>

#include "../../include/fnft_amd.h"

FNFT_INT fnft_amd__kdvv_contspec_host(FNFT_UINT D, const FNFT_COMPLEX *u, const FNFT_REAL *T, FNFT_UINT M,
                                      FNFT_COMPLEX *contspec, const FNFT_REAL *XI, int discretization);
FNFT_INT fnft_amd__raise(FNFT_INT ec, const char *func, int line, const char *msg);

#define E_INVALID_ARGUMENT(name) fnft_amd__raise(FNFT_EC_INVALID_ARGUMENT, __func__, __LINE__, "Invalid argument " #name ".")
#define E_NOT_YET_IMPLEMENTED(name, msg) \
    fnft_amd__raise(FNFT_EC_NOT_YET_IMPLEMENTED, __func__, __LINE__, "Not yet implemented (" #name "). " msg)
#define E_SUBROUTINE(ec) fnft_amd__raise(-abs(ec), __func__, __LINE__, "Subroutine failure.")

/* src/fnft_kdvv.c:34-44 */
static fnft_kdvv_opts_t default_opts = {.discretization = fnft_kdv_discretization_2SPLIT8B};

fnft_kdvv_opts_t fnft_kdvv_default_opts(void) { return default_opts; }

FNFT_INT fnft_kdvv(const FNFT_UINT D, FNFT_COMPLEX *const u, FNFT_REAL const *const T, const FNFT_UINT M,
                   FNFT_COMPLEX *const contspec, FNFT_REAL const *const XI, FNFT_UINT *const K_ptr,
                   FNFT_COMPLEX *const bound_states, FNFT_COMPLEX *const normconsts_or_residues,
                   fnft_kdvv_opts_t *opts_ptr)
{
    /* same checks, same order as src/fnft_kdvv.c:77-94 */
    if (D < 2) return E_INVALID_ARGUMENT(D);
    if (u == NULL) return E_INVALID_ARGUMENT(u);
    if (T == NULL || T[0] >= T[1]) return E_INVALID_ARGUMENT(T);
    if (contspec == NULL) return E_INVALID_ARGUMENT(contspec);
    if (XI == NULL || XI[0] >= XI[1]) return E_INVALID_ARGUMENT(XI);
    if (K_ptr != NULL) return E_NOT_YET_IMPLEMENTED(K_ptr, "Please pass \"NULL\".");
    if (bound_states != NULL) return E_NOT_YET_IMPLEMENTED(bound_states, "Please pass \"NULL\".");
    if (normconsts_or_residues != NULL)
        return E_NOT_YET_IMPLEMENTED(normconsts_or_residues, "Please pass \"NULL\".");
    if (opts_ptr == NULL) opts_ptr = &default_opts;

    int kd = (int)opts_ptr->discretization;
    /* 4SPLIT4A / 4SPLIT4B: the reference's fnft_kdvv hands the RAW samples to kdv_fscatter (no resampling, :108-113),
     * where these two use the per-sample formulas and degrees of 2SPLIT4A / 2SPLIT4B
     * (src/private/fnft__kdv_discretization.c:139-143, fnft__akns_fscatter.c:362-363,402-403): the same transform */
    if (kd == (int)fnft_kdv_discretization_4SPLIT4A) kd = (int)fnft_kdv_discretization_2SPLIT4A;
    if (kd == (int)fnft_kdv_discretization_4SPLIT4B) kd = (int)fnft_kdv_discretization_2SPLIT4B;
    if ((int)opts_ptr->discretization < 0 || (int)opts_ptr->discretization > (int)fnft_kdv_discretization_CF6_4) {
        /* kdv_fscatter_numel returns 0 and kdv_fscatter raises, src/fnft_kdvv.c:100-113 */
        FNFT_INT rc = E_INVALID_ARGUMENT(discretization);
        return E_SUBROUTINE(rc);
    }
    if (kd > (int)fnft_kdv_discretization_2SPLIT8B)
        return E_NOT_YET_IMPLEMENTED(discretization, "GPU path covers the 2SPLIT schemes.");
    if (M < 2) {
        /* eps_xi = (XI1-XI0)/(M-1): the reference divides by zero for M = 1 and returns NaNs;
         * an empty grid is an argument error here */
        return E_INVALID_ARGUMENT(M);
    }
    const FNFT_INT rc = fnft_amd__kdvv_contspec_host(D, u, T, M, contspec, XI, kd);
    if (rc != FNFT_SUCCESS) {
        if (rc == FNFT_EC_OTHER || rc == FNFT_EC_NOMEM)
            return fnft_amd__raise(rc, __func__, __LINE__, "GPU runtime failure (see fnft_amd_last_error()).");
        return E_SUBROUTINE(rc);
    }
    return FNFT_SUCCESS;
}
