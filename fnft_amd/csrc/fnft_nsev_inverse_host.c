/*
 * fnft_nsev_inverse_host.c -- the C driver behind fnft_nsev_inverse / fnft_nsev_inverse_XI /
 * fnft_nsev_inverse_default_opts of include/fnft_amd.h.
 *
 * Mirrors the argument handling and control flow of the reference's src/fnft_nsev_inverse.c:26-248 (validation
 * order, return codes, option defaults, warning texts) and hands the numerical work to the HIP shim
 * (hip_backend.hip: fnft_amd__inverse_*).  There is no CPU fallback.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>

#include "../../include/fnft_amd.h"

FNFT_INT fnft_amd__raise(FNFT_INT ec, const char *func, int line, const char *msg);   /* fnft_nsev_host.c */
FNFT_UINT fnft_amd__nse_degree(fnft_nse_discretization_t d);
void fnft_amd__warn(const char *msg, const char *func, int line);
#define E_INVALID_ARGUMENT(name) fnft_amd__raise(FNFT_EC_INVALID_ARGUMENT, __func__, __LINE__, "Invalid argument " #name ".")
#define E_SANITY_CHECK_FAILED(msg) fnft_amd__raise(FNFT_EC_SANITY_CHECK_FAILED, __func__, __LINE__, "Sanity check failed (" #msg ").")
#define E_SUBROUTINE(ec) fnft_amd__raise(-abs(ec), __func__, __LINE__, "Subroutine failure.")

/* HIP shim */
FNFT_INT fnft_amd__inverse_transfer_matrix(FNFT_UINT M, FNFT_COMPLEX *contspec, const FNFT_REAL *XI, FNFT_UINT K,
                                           const FNFT_COMPLEX *bound_states, FNFT_UINT D, const FNFT_REAL *T,
                                           FNFT_UINT deg, FNFT_COMPLEX *tm, FNFT_INT kappa, int cstype, int method,
                                           FNFT_UINT max_iter, FNFT_UINT oversampling, FNFT_REAL phase_factor,
                                           int *warn_specfact, int *warn_maxiter);
FNFT_INT fnft_amd__inverse_add_discrete(FNFT_UINT K, const FNFT_COMPLEX *bound_states,
                                        const FNFT_COMPLEX *normconsts_or_residues, FNFT_UINT D, FNFT_COMPLEX *q,
                                        const FNFT_REAL *T, int contspec_flag, int residues, int seed_method);

/* src/fnft_nsev_inverse.c:26-33 */
static fnft_nsev_inverse_opts_t default_opts = {
    .discretization = fnft_nse_discretization_2SPLIT2A,
    .contspec_type = fnft_nsev_inverse_cstype_REFLECTION_COEFFICIENT,
    .contspec_inversion_method = fnft_nsev_inverse_csmethod_DEFAULT,
    .discspec_type = fnft_nsev_inverse_dstype_NORMING_CONSTANTS,
    .max_iter = 100,
    .oversampling_factor = 8};

fnft_nsev_inverse_opts_t fnft_nsev_inverse_default_opts(void) { return default_opts; }

/* src/fnft_nsev_inverse.c:40-65; z -> lambda of fnft__akns_discretization.c:225-240 with degree 1, no upsampling */
FNFT_INT fnft_nsev_inverse_XI(const FNFT_UINT D, FNFT_REAL const *const T, const FNFT_UINT M, FNFT_REAL *const XI,
                              const fnft_nse_discretization_t discretization)
{
    if (D < 2)
        return E_INVALID_ARGUMENT(D);
    if (M == 0)
        return E_INVALID_ARGUMENT(M);
    if (XI == NULL)
        return E_INVALID_ARGUMENT(XI);
    if (T == NULL || !(T[0] < T[1]))
        return E_INVALID_ARGUMENT(T);
    /* degree of one step times the upsampling factor, fnft__akns_discretization.c:225-240; unknown (slow)
     * discretizations have degree 0 there and fail the same way */
    FNFT_REAL degree1step = (FNFT_REAL)fnft_amd__nse_degree(discretization);
    if (degree1step == 0)
        return E_SUBROUTINE(FNFT_EC_INVALID_ARGUMENT);
    if (discretization == fnft_nse_discretization_4SPLIT4A || discretization == fnft_nse_discretization_4SPLIT4B)
        degree1step *= 2.0;
    const FNFT_REAL eps_t = (T[1] - T[0]) / (D - 1);
    const double complex z0 = cexp(2.0 * 3.14159265358979323846 * I * (double)(M / 2 + 1) / (double)M);
    const double complex z1 = -1.0;
    XI[0] = creal(clog(z0) / (2.0 * I * eps_t / degree1step));
    XI[1] = creal(clog(z1) / (2.0 * I * eps_t / degree1step));
    return FNFT_SUCCESS;
}

/* src/fnft_nsev_inverse.c:121-248 */
/* contspec[i] *= prod_k (xi_i - lambda_k) / (xi_i - conj(lambda_k)): what the successful path does on the device
 * (body_inv_op); only the validation failures of the reflection-coefficient branch need it on the host */
static void blaschke_on_host(FNFT_UINT M, FNFT_COMPLEX *contspec, const FNFT_REAL *XI, FNFT_UINT K,
                             const FNFT_COMPLEX *bound_states)
{
    if (K == 0 || contspec == NULL || XI == NULL || M < 2) return;
    const FNFT_REAL step = (XI[1] - XI[0]) / (FNFT_REAL)(M - 1);
    for (FNFT_UINT i = 0; i < M; i++) {
        const FNFT_REAL xi = XI[0] + (FNFT_REAL)i * step;
        for (FNFT_UINT k = 0; k < K; k++) contspec[i] *= (xi - bound_states[k]) / (xi - conj(bound_states[k]));
    }
}

FNFT_INT fnft_nsev_inverse(const FNFT_UINT M, FNFT_COMPLEX *const contspec, FNFT_REAL const *const XI,
                           FNFT_UINT const K, FNFT_COMPLEX const *const bound_states,
                           FNFT_COMPLEX const *const normconsts_or_residues, const FNFT_UINT D, FNFT_COMPLEX *const q,
                           FNFT_REAL const *const T, const FNFT_INT kappa, fnft_nsev_inverse_opts_t *opts_ptr)
{
    if (M > 0 && contspec == NULL)
        return E_INVALID_ARGUMENT(contspec);
    if (contspec != NULL && M % 2 != 0)
        return E_INVALID_ARGUMENT(M);
    if (contspec != NULL && M < D)
        return E_INVALID_ARGUMENT(M);
    if (D < 2 || (D & (D - 1)) != 0)
        return E_INVALID_ARGUMENT(D);
    if (q == NULL)
        return E_INVALID_ARGUMENT(q);
    if (T == NULL || !(T[0] < T[1]))
        return E_INVALID_ARGUMENT(T);
    if (kappa != +1 && kappa != -1)
        return E_INVALID_ARGUMENT(kappa);
    if (K > 0 && kappa != +1)
        return E_SANITY_CHECK_FAILED(Discrete spectrum is present only in the focussing case(kappa=1).);
    if (K > 0 && bound_states == NULL)
        return E_INVALID_ARGUMENT(bound_states);
    FNFT_UINT i;
    for (i = 0; i < K; i++) {
        if (cimag(bound_states[i]) <= 0)
            return E_SANITY_CHECK_FAILED(bound_states should be stricly in the upper-half complex-plane.);
    }
    if (K > 0 && normconsts_or_residues == NULL)
        return E_INVALID_ARGUMENT(normconsts_or_residues);
    if (opts_ptr == NULL)
        opts_ptr = &default_opts;
    if (opts_ptr->discretization != fnft_nse_discretization_2SPLIT2A
        && opts_ptr->discretization != fnft_nse_discretization_2SPLIT2_MODAL)
        return E_INVALID_ARGUMENT(opts_ptr->discretization);
    if (contspec == NULL && K == 0)
        return E_SANITY_CHECK_FAILED(Neither contspec nor discspec provided.);
    if (XI == NULL && contspec != NULL && opts_ptr->contspec_type != fnft_nsev_inverse_cstype_B_OF_TAU)
        return E_INVALID_ARGUMENT(XI);

    FNFT_INT ret_code = FNFT_SUCCESS;
    int contspec_flag = 0;
    FNFT_COMPLEX *transfer_matrix = NULL;

    if (contspec != NULL) {
        contspec_flag = 1;
        const FNFT_UINT deg = D;   /* degree 1 per step for both admissible discretizations */
        transfer_matrix = malloc(4 * (deg + 1) * sizeof(FNFT_COMPLEX));
        if (transfer_matrix == NULL)
            return fnft_amd__raise(FNFT_EC_NOMEM, __func__, __LINE__, "Out of memory.");
        const FNFT_REAL eps_t = (T[1] - T[0]) / (D - 1);
        /* boundary coefficient 0.5, degree1step 1: fnft__nse_discretization.c:240-258 and :319-379 */
        const FNFT_REAL pf_rho = -2.0 * (T[1] + eps_t * 0.5) + eps_t;
        const FNFT_REAL pf_b = -eps_t * D - (T[1] + eps_t * 0.5) - (T[0] - eps_t * 0.5) + eps_t;
        int method = 0, cstype, w_sf = 0, w_it = 0;
        FNFT_REAL pf = 0.0;

        /* Step 1: the transfer matrix from the given representation of the continuous spectrum, :182-214 */
        switch (opts_ptr->contspec_type) {
        case fnft_nsev_inverse_cstype_REFLECTION_COEFFICIENT:
            cstype = 0;
            pf = pf_rho;
            switch (opts_ptr->contspec_inversion_method) {   /* :512-556 */
            case fnft_nsev_inverse_csmethod_DEFAULT:
            case fnft_nsev_inverse_csmethod_TFMATRIX_CONTAINS_REFL_COEFF:
                method = 1;
                break;
            case fnft_nsev_inverse_csmethod_TFMATRIX_CONTAINS_AB_FROM_ITER:
                method = 2;
                /* :393-400.  The reference has multiplied the Blaschke factors into the caller's contspec (:198-199,
                 * :1013-1033) before these checks fail; a caller that inspects contspec afterwards sees the same here */
                if (M != D) {
                    blaschke_on_host(M, contspec, XI, K, bound_states);
                    ret_code = E_SUBROUTINE(E_INVALID_ARGUMENT(M));
                    goto leave_fun;
                }
                if (kappa != -1) {
                    blaschke_on_host(M, contspec, XI, K, bound_states);
                    ret_code = E_SUBROUTINE(E_INVALID_ARGUMENT(kappa));
                    goto leave_fun;
                }
                break;
            default:
                blaschke_on_host(M, contspec, XI, K, bound_states);
                ret_code = E_SUBROUTINE(E_INVALID_ARGUMENT(opts_ptr->contspec_inversion_method));
                goto leave_fun;
            }
            break;
        case fnft_nsev_inverse_cstype_B_OF_XI:
            cstype = 1;
            pf = pf_b;
            break;
        case fnft_nsev_inverse_cstype_B_OF_TAU:   /* :643-650 */
            cstype = 2;
            if (M != D) { ret_code = E_SUBROUTINE(E_INVALID_ARGUMENT(M)); goto leave_fun; }
            if (T[0] != -T[1]) { ret_code = E_SUBROUTINE(E_INVALID_ARGUMENT(T)); goto leave_fun; }
            if (opts_ptr->contspec_inversion_method != fnft_nsev_inverse_csmethod_DEFAULT) {
                ret_code = E_SUBROUTINE(E_INVALID_ARGUMENT(opts_ptr->contspec_inversion_method));
                goto leave_fun;
            }
            break;
        default:
            ret_code = E_INVALID_ARGUMENT(opts_ptr->contspec_type);
            goto leave_fun;
        }
        if (cstype != 0 && opts_ptr->oversampling_factor == 0) {   /* fnft__poly_specfact.c:37-38 */
            ret_code = E_SUBROUTINE(FNFT_EC_INVALID_ARGUMENT);
            goto leave_fun;
        }
        ret_code = fnft_amd__inverse_transfer_matrix(M, contspec, XI, K, bound_states, D, T, deg, transfer_matrix, kappa,
                                                     cstype, method, opts_ptr->max_iter, opts_ptr->oversampling_factor,
                                                     pf, &w_sf, &w_it);
        if (ret_code != FNFT_SUCCESS) { ret_code = E_SUBROUTINE(ret_code); goto leave_fun; }
        if (w_sf) fnft_amd__warn("Ill-posed spectral factorization problem.", "fnft__poly_specfact", __LINE__);
        if (w_it)
            fnft_amd__warn("Maximum number of iterations reached when constructing transfer matrix.", __func__, __LINE__);

        /* Step 2: the time-domain signal from the transfer matrix, :218-222 */
        ret_code = fnft__nse_finvscatter(deg, transfer_matrix, q, eps_t, kappa, opts_ptr->discretization);
        if (ret_code != FNFT_SUCCESS) { ret_code = E_SUBROUTINE(ret_code); goto leave_fun; }
    }

    if (K > 0) {   /* :227-232 */
        const int seed = opts_ptr->contspec_inversion_method == fnft_nsev_inverse_csmethod_USE_SEED_POTENTIAL_INSTEAD;
        ret_code = fnft_amd__inverse_add_discrete(K, bound_states, normconsts_or_residues, D, q, T, contspec_flag,
                                                  opts_ptr->discspec_type == fnft_nsev_inverse_dstype_RESIDUES, seed);
        if (ret_code != FNFT_SUCCESS) { ret_code = E_SUBROUTINE(ret_code); goto leave_fun; }
    }

leave_fun:
    free(transfer_matrix);
    return ret_code;
}
