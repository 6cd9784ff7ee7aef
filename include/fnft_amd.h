/*
 * fnft_amd.h -- C ABI of libfnft_amd.so: the MI355X-native fast nonlinear Fourier transform
 * (continuous spectrum of the NSE with vanishing boundaries).
 *
 * Section 1 is the DROP-IN boundary: the same symbols, argument meaning, ownership rules and
 * return codes as the reference's libfnft.so, so a caller of the reference relinks against
 * libfnft_amd.so unchanged.  Every declaration cites the reference interface it replaces
 * (file:line relative to the FNFT source tree).
 * Section 2 is the optional second seam (the reference's exported private layer).
 * Section 3 is the device-resident extension used when inputs already live in HBM.
 *
 * Plain pointers and sizes only; no framework types.
 */
#ifndef FNFT_AMD_H
#define FNFT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#include <complex>
typedef std::complex<double> FNFT_COMPLEX; /* include/fnft_numtypes.h:47-50 */
extern "C" {
#else
#include <complex.h>
typedef double complex FNFT_COMPLEX; /* include/fnft_numtypes.h:47-50 */
#endif

typedef double FNFT_REAL; /* include/fnft_numtypes.h:40 */
typedef int32_t FNFT_INT; /* include/fnft_numtypes.h:56 */
typedef size_t FNFT_UINT; /* include/fnft_numtypes.h:62 */

/* ---- return codes, include/fnft_errwarn.h:44-94 ------------------------------------------ */
#define FNFT_SUCCESS 0
#define FNFT_EC_NOMEM 1
#define FNFT_EC_INVALID_ARGUMENT 2
#define FNFT_EC_DIV_BY_ZERO 3
#define FNFT_EC_TEST_FAILED 4
#define FNFT_EC_OTHER 5
#define FNFT_EC_NOT_YET_IMPLEMENTED 6
#define FNFT_EC_SANITY_CHECK_FAILED 7
#define FNFT_EC_ASSERTION_FAILED 8

/* ---- error/warning text hook, include/fnft_errwarn.h:36,101,108 --------------------------- */
typedef FNFT_INT (*fnft_printf_ptr_t)(const char *, ...);
void fnft_errwarn_setprintf(fnft_printf_ptr_t printf_ptr);
fnft_printf_ptr_t fnft_errwarn_getprintf(void);

/* ---- discretizations, include/fnft_nse_discretization_t.h:104-133 (same ordinals) --------- */
typedef enum {
    fnft_nse_discretization_2SPLIT2_MODAL,
    fnft_nse_discretization_BO,
    fnft_nse_discretization_2SPLIT1A,
    fnft_nse_discretization_2SPLIT1B,
    fnft_nse_discretization_2SPLIT2A,
    fnft_nse_discretization_2SPLIT2B,
    fnft_nse_discretization_2SPLIT2S,
    fnft_nse_discretization_2SPLIT3A,
    fnft_nse_discretization_2SPLIT3B,
    fnft_nse_discretization_2SPLIT3S,
    fnft_nse_discretization_2SPLIT4A,
    fnft_nse_discretization_2SPLIT4B,
    fnft_nse_discretization_2SPLIT5A,
    fnft_nse_discretization_2SPLIT5B,
    fnft_nse_discretization_2SPLIT6A,
    fnft_nse_discretization_2SPLIT6B,
    fnft_nse_discretization_2SPLIT7A,
    fnft_nse_discretization_2SPLIT7B,
    fnft_nse_discretization_2SPLIT8A,
    fnft_nse_discretization_2SPLIT8B,
    fnft_nse_discretization_4SPLIT4A,
    fnft_nse_discretization_4SPLIT4B,
    fnft_nse_discretization_CF4_2,
    fnft_nse_discretization_CF4_3,
    fnft_nse_discretization_CF5_3,
    fnft_nse_discretization_CF6_4,
    fnft_nse_discretization_ES4,
    fnft_nse_discretization_TES4
} fnft_nse_discretization_t;

/* include/private/fnft__akns_discretization_t.h:104-134 (same ordinals) */
typedef enum {
    fnft__akns_discretization_2SPLIT2_MODAL,
    fnft__akns_discretization_2SPLIT1A,
    fnft__akns_discretization_2SPLIT1B,
    fnft__akns_discretization_2SPLIT2A,
    fnft__akns_discretization_2SPLIT2B,
    fnft__akns_discretization_2SPLIT2S,
    fnft__akns_discretization_2SPLIT3A,
    fnft__akns_discretization_2SPLIT3B,
    fnft__akns_discretization_2SPLIT3S,
    fnft__akns_discretization_2SPLIT4A,
    fnft__akns_discretization_2SPLIT4B,
    fnft__akns_discretization_2SPLIT5A,
    fnft__akns_discretization_2SPLIT5B,
    fnft__akns_discretization_2SPLIT6A,
    fnft__akns_discretization_2SPLIT6B,
    fnft__akns_discretization_2SPLIT7A,
    fnft__akns_discretization_2SPLIT7B,
    fnft__akns_discretization_2SPLIT8A,
    fnft__akns_discretization_2SPLIT8B,
    fnft__akns_discretization_BO,
    fnft__akns_discretization_4SPLIT4A,
    fnft__akns_discretization_4SPLIT4B,
    fnft__akns_discretization_CF4_2,
    fnft__akns_discretization_CF4_3,
    fnft__akns_discretization_CF5_3,
    fnft__akns_discretization_CF6_4,
    fnft__akns_discretization_ES4,
    fnft__akns_discretization_TES4
} fnft__akns_discretization_t;

/* ---- fnft_nsev options, include/fnft_nsev.h:51-55,91-95,108-112,130-134,198-208 ----------- */
typedef enum {
    fnft_nsev_bsfilt_NONE,
    fnft_nsev_bsfilt_BASIC,
    fnft_nsev_bsfilt_FULL
} fnft_nsev_bsfilt_t;

typedef enum {
    fnft_nsev_bsloc_FAST_EIGENVALUE,
    fnft_nsev_bsloc_NEWTON,
    fnft_nsev_bsloc_SUBSAMPLE_AND_REFINE
} fnft_nsev_bsloc_t;

typedef enum {
    fnft_nsev_dstype_NORMING_CONSTANTS,
    fnft_nsev_dstype_RESIDUES,
    fnft_nsev_dstype_BOTH
} fnft_nsev_dstype_t;

typedef enum {
    fnft_nsev_cstype_REFLECTION_COEFFICIENT,
    fnft_nsev_cstype_AB,
    fnft_nsev_cstype_BOTH
} fnft_nsev_cstype_t;

typedef struct {
    fnft_nsev_bsfilt_t bound_state_filtering;
    fnft_nsev_bsloc_t bound_state_localization;
    FNFT_UINT niter;
    FNFT_UINT Dsub;
    fnft_nsev_dstype_t discspec_type;
    fnft_nsev_cstype_t contspec_type;
    FNFT_INT normalization_flag;
    fnft_nse_discretization_t discretization;
    FNFT_UINT richardson_extrapolation_flag;
} fnft_nsev_opts_t;

/* ======================================================================================== */
/* 1. Drop-in boundary                                                                      */
/* ======================================================================================== */

/* include/fnft_nsev.h:226 -- defaults: FULL filtering, SUBSAMPLE_AND_REFINE, niter 10, Dsub 0,
 * NORMING_CONSTANTS, REFLECTION_COEFFICIENT, normalization on, 2SPLIT4B, no Richardson
 * (src/fnft_nsev.c:26-36). */
fnft_nsev_opts_t fnft_nsev_default_opts(void);

/* include/fnft_nsev.h:241-242 -- degree(discretization) * D (src/fnft_nsev.c:51-57). */
FNFT_UINT fnft_nsev_max_K(const FNFT_UINT D, fnft_nsev_opts_t const *const opts);

/* include/fnft_nsev.h:371-376.  All buffers are HOST memory owned by the caller:
 *   q[D]; T[2]; XI[2]; contspec[M], [2M] or [3M] by opts->contspec_type (NULL: skip);
 *   K_ptr / bound_states / normconsts_or_residues: discrete spectrum.
 * Continuous spectrum (all 21 fast discretizations, Richardson extrapolation) and discrete
 * spectrum (kappa == +1 and bound_states != NULL: bound states by FAST_EIGENVALUE, NEWTON or
 * SUBSAMPLE_AND_REFINE, filtering, norming constants / residues / both, *K_ptr = capacity in,
 * count out) are computed on the GPU.  The slow discretizations (BO, CF*, ES4, TES4) return
 * FNFT_EC_NOT_YET_IMPLEMENTED; argument errors return the codes the reference returns, checked
 * in the same order (src/fnft_nsev.c:163-220). */
FNFT_INT fnft_nsev(const FNFT_UINT D, FNFT_COMPLEX *const q, FNFT_REAL const *const T,
                   const FNFT_UINT M, FNFT_COMPLEX *const contspec, FNFT_REAL const *const XI,
                   FNFT_UINT *const K_ptr, FNFT_COMPLEX *const bound_states,
                   FNFT_COMPLEX *const normconsts_or_residues, const FNFT_INT kappa,
                   fnft_nsev_opts_t *opts);

/* ---------------------------------------------------------------------------------------- */
/* 1b. fnft_nsev_inverse (round 2; SURVEY 8f rank 4: the caller below the forward path)      */
/* ---------------------------------------------------------------------------------------- */

/* include/fnft_nsev_inverse.h:58-62 */
typedef enum {
    fnft_nsev_inverse_cstype_REFLECTION_COEFFICIENT,
    fnft_nsev_inverse_cstype_B_OF_XI,
    fnft_nsev_inverse_cstype_B_OF_TAU
} fnft_nsev_inverse_cstype_t;
/* include/fnft_nsev_inverse.h:76-79 */
typedef enum {
    fnft_nsev_inverse_dstype_NORMING_CONSTANTS,
    fnft_nsev_inverse_dstype_RESIDUES
} fnft_nsev_inverse_dstype_t;
/* include/fnft_nsev_inverse.h:110-115 */
typedef enum {
    fnft_nsev_inverse_csmethod_DEFAULT,
    fnft_nsev_inverse_csmethod_TFMATRIX_CONTAINS_REFL_COEFF,
    fnft_nsev_inverse_csmethod_TFMATRIX_CONTAINS_AB_FROM_ITER,
    fnft_nsev_inverse_csmethod_USE_SEED_POTENTIAL_INSTEAD
} fnft_nsev_inverse_csmethod_t;
/* include/fnft_nsev_inverse.h:155-162 (same fields, same order) */
typedef struct {
    fnft_nse_discretization_t discretization;
    fnft_nsev_inverse_cstype_t contspec_type;
    fnft_nsev_inverse_csmethod_t contspec_inversion_method;
    fnft_nsev_inverse_dstype_t discspec_type;
    FNFT_UINT max_iter;
    FNFT_UINT oversampling_factor;
} fnft_nsev_inverse_opts_t;

/* include/fnft_nsev_inverse.h:176 (src/fnft_nsev_inverse.c:26-38): 2SPLIT2A, reflection coefficient, DEFAULT method,
 * norming constants, max_iter 100, oversampling_factor 8 */
fnft_nsev_inverse_opts_t fnft_nsev_inverse_default_opts(void);
/* include/fnft_nsev_inverse.h:199-204 (src/fnft_nsev_inverse.c:40-65): the xi-grid XI[0..1] that makes the M samples
 * of the continuous spectrum an M-point FFT grid for D samples on T */
FNFT_INT fnft_nsev_inverse_XI(const FNFT_UINT D, FNFT_REAL const *const T, const FNFT_UINT M, FNFT_REAL *const XI,
                              const fnft_nse_discretization_t discretization);
/* include/fnft_nsev_inverse.h:275-286 (src/fnft_nsev_inverse.c:121-248).  HOST buffers owned by the caller; like the
 * reference, contspec[M] is modified (boundary phase factors removed, Blaschke factors applied).  D a power of two,
 * M even and >= D, discretization 2SPLIT2A or 2SPLIT2_MODAL.  Argument checks in the reference's order with its return
 * codes.  On the GPU: the M-point and spectral-factorization DFTs (any length), the element-wise stages, the layer
 * peeling's products (fnft__nse_finvscatter), the multi-soliton recursion, the seed's eigenfunctions and the Darboux
 * steps.  Limit: DFT lengths up to 2^24 (chirp transforms of 2^25 points), i.e. with the default oversampling factor 8
 * of the spectral factorization (contspec_type B_OF_XI; B_OF_TAU and REFLECTION_COEFFICIENT factorize at degree D - 1
 * resp. not at all) D <= 2^20; beyond it the call fails with FNFT_EC_NOT_YET_IMPLEMENTED wrapped as a subroutine
 * failure.  fnft__nse_finvscatter itself: D * degree <= 2^24 as for the forward tree.  A call that fails one of the
 * checks of the reflection-coefficient branch (src/fnft_nsev_inverse.c:393-400, 512-556) leaves contspec multiplied by
 * the Blaschke factors of the bound states, as the reference does (:198-199). */
FNFT_INT fnft_nsev_inverse(const FNFT_UINT M, FNFT_COMPLEX *const contspec, FNFT_REAL const *const XI,
                           FNFT_UINT const K, FNFT_COMPLEX const *const bound_states,
                           FNFT_COMPLEX const *const normconsts_or_residues, const FNFT_UINT D, FNFT_COMPLEX *const q,
                           FNFT_REAL const *const T, const FNFT_INT kappa, fnft_nsev_inverse_opts_t *opts_ptr);

/* ======================================================================================== */
/* 2. Private-layer seam (symbols the reference's libfnft.so also exports)                  */
/* ======================================================================================== */
/* Argument errors of fnft__akns_fscatter, fnft__kdv_fscatter, fnft__nse_finvscatter, fnft__poly_specfact,
 * fnft__poly_chirpz (pointer twin), fnft__misc_resample, fnft__poly_roots_fasteigen, the grid searches and
 * fnft__nse_scatter_bound_states are raised like the reference raises them: return code FNFT_EC_INVALID_ARGUMENT plus
 * "Invalid argument <name>." through the fnft_errwarn_setprintf hook. */

/* include/private/fnft__poly_fmult.h:199 -- 4*(deg+1)*nextpow2(n) */
FNFT_UINT fnft__poly_fmult2x2_numel(const FNFT_UINT deg, const FNFT_UINT n);

/* include/private/fnft__poly_fmult.h:223-224 (src/private/fnft__poly_fmult.c:381-546).
 * Host buffers; p = [p11|p12|p21|p22], each n*(deg+1), is overwritten; result receives
 * [r11|r12|r21|r22], each (*d)+1 on return; *W_ptr (if non-NULL) receives the power-of-two
 * exponent with  true product = result * 2^W. */
FNFT_INT fnft__poly_fmult2x2(FNFT_UINT *const d, FNFT_UINT n, FNFT_COMPLEX *const p,
                             FNFT_COMPLEX *const result, FNFT_INT *const W_ptr);

/* include/private/fnft__poly_fmult.h:162,183-184 (src/private/fnft__poly_fmult.c:35-38,152-237): product of n scalar
 * polynomials of degree *d, in place (p holds n*(deg+1) coefficients on entry, the (*d)+1 coefficients of the
 * product on return; sized by fnft__poly_fmult_numel); *W_ptr as in fnft__poly_fmult2x2. */
FNFT_UINT fnft__poly_fmult_numel(const FNFT_UINT deg, const FNFT_UINT n);
FNFT_INT fnft__poly_fmult(FNFT_UINT *const d, FNFT_UINT n, FNFT_COMPLEX *const p, FNFT_INT *const W_ptr);

/* include/private/fnft__poly_fmult.h:39,82-92,136-148 (src/private/fnft__poly_fmult.c:45-121,239-328): one product of
 * two polynomials / of two 2x2 polynomial matrices of degree deg -- the calls fnft__nse_finvscatter.c:128,155 makes.
 * plan_fwd / plan_inv (the reference's FFT-wrapper handles) and the scratch buffers are accepted for call
 * compatibility and not used by the GPU transforms, except that buf1 / buf2 carry the factors from one call of
 * fnft__poly_fmult_two_polys to the next (a NULL factor = the previous call's) and `result` the partial sum between a
 * mode-2 and a mode-3 call -- as coefficients, where the reference keeps spectra.  fnft__poly_fmult_two_polys_len is the
 * reference's buffer length (kiss_fft_next_fast_size(2*deg+1)). */
FNFT_INT fnft__poly_fmult_two_polys_len(const FNFT_UINT deg);
FNFT_INT fnft__poly_fmult_two_polys(const FNFT_UINT deg, FNFT_COMPLEX const *const p1, FNFT_COMPLEX const *const p2,
                                    FNFT_COMPLEX *const result, void *plan_fwd, void *plan_inv, FNFT_COMPLEX *const buf0,
                                    FNFT_COMPLEX *const buf1, FNFT_COMPLEX *const buf2, const FNFT_UINT mode);
FNFT_INT fnft__poly_fmult_two_polys2x2(const FNFT_UINT deg, FNFT_COMPLEX const *const p1_11, const FNFT_UINT p1_stride,
                                       FNFT_COMPLEX const *const p2_11, const FNFT_UINT p2_stride,
                                       FNFT_COMPLEX *const result_11, const FNFT_UINT result_stride, void *plan_fwd,
                                       void *plan_inv, FNFT_COMPLEX *const buf0, FNFT_COMPLEX *const buf1,
                                       FNFT_COMPLEX *const buf2, const FNFT_UINT mode_offset);

/* include/private/fnft__nse_finvscatter.h (src/private/fnft__nse_finvscatter.c:234-366): fast inverse scattering by
 * layer peeling -- the D = deg samples q behind a transfer matrix [T11|T12|T21|T22] (each deg+1, highest power first,
 * as fnft__nse_fscatter returns it with W_ptr = NULL), D a power of two, 2SPLIT2_MODAL or 2SPLIT2A.  Host buffers in
 * and out; in between everything is on the GPU: blocks of 256 samples by a leaf kernel, the 2x2 polynomial products
 * above that by the tree's pair kernels, no synchronisation inside the recursion.  FNFT_EC_OTHER if a reconstructed
 * sample violates |eps_t q| < 1 (defocusing) or D is not a power of two. */
FNFT_INT fnft__nse_finvscatter(const FNFT_UINT deg, FNFT_COMPLEX *const transfer_matrix, FNFT_COMPLEX *const q,
                               const FNFT_REAL eps_t, const FNFT_INT kappa, const fnft_nse_discretization_t discretization);

/* include/private/fnft__poly_specfact.h (src/private/fnft__poly_specfact.c:25-140): spectral factor of
 * |P|^2 (kappa = 0), 1 + |P|^2 (kappa = -1) or 1 - |P|^2 (kappa = +1) on the unit circle by the cepstral method on a
 * grid of next_fast_size((deg+1)*oversampling_factor) points (the reference's 2-3-5-smooth FFT length, as a DFT of
 * that length on the GPU).  poly, result: deg+1 coefficients, host.  Prints the reference's "Ill-posed spectral
 * factorization problem." warning through the printf hook. */
FNFT_INT fnft__poly_specfact(const FNFT_UINT deg, FNFT_COMPLEX const *const poly, FNFT_COMPLEX *const result,
                             const FNFT_UINT oversampling_factor, const FNFT_INT kappa);

/* include/private/fnft__poly_chirpz.h:61 (src/private/fnft__poly_chirpz.c:33-105).
 * A and W are passed as pointers to {re, im} here (complex-by-value does not cross every FFI);
 * fnft__poly_chirpz below is the by-value form with the reference's exact signature. */
FNFT_INT fnft_amd_poly_chirpz(const FNFT_UINT deg, FNFT_COMPLEX const *const p,
                              const double *A_reim, const double *W_reim, const FNFT_UINT M,
                              FNFT_COMPLEX *const result);
FNFT_INT fnft__poly_chirpz(const FNFT_UINT deg, FNFT_COMPLEX const *const p, const FNFT_COMPLEX A,
                           const FNFT_COMPLEX W, const FNFT_UINT M, FNFT_COMPLEX *const result);

/* include/private/fnft__misc.h:241-242 (src/private/fnft__misc.c:326-407): q_new = q shifted by delta on the
 * periodically continued, band-limited grid (the 4SPLIT4A/B front end).  Host buffers of D samples, D > 2. */
FNFT_INT fnft__misc_resample(const FNFT_UINT D, const FNFT_REAL eps_t, FNFT_COMPLEX const *const q,
                             const FNFT_REAL delta, FNFT_COMPLEX *const q_new);

/* include/private/fnft__poly_roots_fasteigen.h (src/private/fnft__poly_roots_fasteigen.c:29-48): the deg roots of
 * p[0] z^deg + ... + p[deg] (host buffers), in no particular order.  The reference's Fortran QR (eiscor) is
 * replaced by Ehrlich-Aberth sweeps on the GPU; failure to converge returns -FNFT_EC_OTHER as there. */
FNFT_INT fnft__poly_roots_fasteigen(const FNFT_UINT deg, FNFT_COMPLEX const *const p, FNFT_COMPLEX *const roots);

/* include/private/fnft__nse_scatter.h:76-80 (src/private/fnft__nse_scatter_bound_states.c:29-667), BO and CF4_2
 * schemes (the two fnft_nsev refines with, src/fnft_nsev.c:669-676): a(lambda_k), a'(lambda_k) and, unless
 * skip_b_flag, b(lambda_k) for K points; q: D samples on [T0, T1] (CF4_2: the D preprocessed samples, two per grid
 * point, D even, :188-197), r is not read (r = -conj(q)).  Other schemes: FNFT_EC_NOT_YET_IMPLEMENTED. */
FNFT_INT fnft__nse_scatter_bound_states(const FNFT_UINT D, FNFT_COMPLEX const *const q, FNFT_COMPLEX *r,
                                        FNFT_REAL const *const T, FNFT_UINT K, FNFT_COMPLEX *bound_states,
                                        FNFT_COMPLEX *a_vals, FNFT_COMPLEX *aprime_vals, FNFT_COMPLEX *b,
                                        fnft_nse_discretization_t discretization, FNFT_UINT skip_b_flag);

/* include/private/fnft__poly_roots_fftgridsearch.h (src/private/fnft__poly_roots_fftgridsearch.c:35-151, :159-217):
 * estimates of the roots of p (deg+1 coefficients, highest power first) on the arc exp(i phi), PHI[0] <= phi <= PHI[1],
 * of the unit circle from a grid of *M_ptr points -- p evaluated on three rings by chirp z-transforms, local minima of
 * |p|, one step of a least-squares linear fit; the para-Hermitian form (even degree) from sign changes of one ring.
 * *M_ptr returns the number of estimates, roots (*M_ptr entries on entry) holds them in grid order.  Host buffers. */
FNFT_INT fnft__poly_roots_fftgridsearch(const FNFT_UINT deg, FNFT_COMPLEX const *const p, FNFT_UINT *const M_ptr,
                                        FNFT_REAL const *const PHI, FNFT_COMPLEX *const roots);
FNFT_INT fnft__poly_roots_fftgridsearch_paraherm(const FNFT_UINT deg, FNFT_COMPLEX const *const p,
                                                 FNFT_UINT *const M_ptr, FNFT_REAL const *const PHI,
                                                 FNFT_COMPLEX *const roots);

/* include/private/fnft__nse_scatter.h:119-123 (src/private/fnft__nse_scatter_matrix.c:33-86, BO and CF4_2 schemes
 * of fnft__akns_scatter_matrix.c:116-130): scattering matrix S(lambda) = U_{D-1} ... U_0 of the D samples q
 * (r = -kappa conj(q) when r == NULL) with step eps_t for K values of lambda; result holds [S11 S12 S21 S22] per
 * lambda, followed by the derivatives with respect to lambda [S11' S12' S21' S22'] when derivative_flag != 0 (4K or
 * 8K values).  CF4_2: q holds two preprocessed samples per step (D even, FNFT_EC_ASSERTION_FAILED otherwise).  Host
 * buffers; chunk-parallel on the GPU.  Other discretizations: FNFT_EC_NOT_YET_IMPLEMENTED. */
FNFT_INT fnft__nse_scatter_matrix(const FNFT_UINT D, FNFT_COMPLEX const *const q, FNFT_COMPLEX *r, const FNFT_REAL eps_t,
                                  const FNFT_INT kappa, const FNFT_UINT K, FNFT_COMPLEX const *const lambda,
                                  FNFT_COMPLEX *const result, fnft_nse_discretization_t discretization,
                                  const FNFT_UINT derivative_flag);

/* include/private/fnft__akns_fscatter.h:57,89-90 (src/private/fnft__akns_fscatter.c:34-42,64-925) */
FNFT_UINT fnft__akns_fscatter_numel(FNFT_UINT D, fnft__akns_discretization_t discretization);
FNFT_INT fnft__akns_fscatter(const FNFT_UINT D, FNFT_COMPLEX const *const q,
                             FNFT_COMPLEX const *const r, const FNFT_REAL eps_t,
                             FNFT_COMPLEX *const result, FNFT_UINT *const deg_ptr,
                             FNFT_INT *const W_ptr, fnft__akns_discretization_t discretization);

/* include/private/fnft__nse_fscatter.h:49,81-84 (src/private/fnft__nse_fscatter.c:34-91) */
FNFT_UINT fnft__nse_fscatter_numel(FNFT_UINT D, fnft_nse_discretization_t discretization);
FNFT_INT fnft__nse_fscatter(const FNFT_UINT D, FNFT_COMPLEX const *const q, const FNFT_REAL eps_t,
                            const FNFT_INT kappa, FNFT_COMPLEX *const result,
                            FNFT_UINT *const deg_ptr, FNFT_INT *const W_ptr,
                            fnft_nse_discretization_t discretization);

/* ======================================================================================== */
/* 3. Device-resident extension (no counterpart in the reference: it has no device path)    */
/* ======================================================================================== */

typedef struct fnft_amd_plan fnft_amd_plan_t;

/* Device rule.  Every host-pointer entry point of sections 1, 2 and 4 (fnft_nsev, fnft_kdvv,
 * fnft__*_fscatter, fnft__poly_fmult*, fnft__poly_chirpz) computes on the calling thread's CURRENT
 * HIP device -- the one hipSetDevice() / torch.cuda.set_device() selected, device 0 if nothing was
 * selected -- and leaves the current device unchanged; cached plans are keyed by that device.  In a
 * one-process-per-GPU job each rank therefore selects its GPU once (LOCAL_RANK) and calls the
 * drop-in as usual.  A device-resident plan lives on the `device` given at creation; plan calls
 * may be made with any current device and restore it before they return. */

/* Number of HIP devices visible; <0 on HIP failure. */
int fnft_amd_device_count(void);
/* The calling thread's current HIP device (what the host-pointer entry points will use); <0: none. */
int fnft_amd_current_device(void);
/* Last HIP error text of the calling thread ("" if none). */
const char *fnft_amd_last_error(void);
/* What the host-pointer entry points keep in HBM between calls -- per device: up to four cached plans per drop-in
 * (fnft_nsev, fnft_kdvv), the resident layer-peeling state of fnft__nse_finvscatter / fnft_nsev_inverse (plans per
 * degree and work arrays, about 100 MB after one D = 2^18 call), and a cache of released work arrays (at most 6 GiB per
 * process; a repeated call then performs no hipMalloc / hipFree) -- is given back to the driver.  device < 0: all
 * devices.  Plans the caller created with fnft_amd_plan_create are not touched. */
void fnft_amd_release_cached(int device);

/* Plan for `batch` independent signals of D samples each, M spectral points, one
 * discretization.  Allocates every workspace the call needs in HBM once. */
FNFT_INT fnft_amd_plan_create(fnft_amd_plan_t **plan, FNFT_UINT D, FNFT_UINT M, FNFT_UINT batch,
                              fnft_nse_discretization_t discretization, int device);
/* fnft__poly_fmult2x2 (src/private/fnft__poly_fmult.c:381-546) on DEVICE buffers: d_p holds n matrices of degree deg in
 * the reference's input layout [4][n*(deg+1)], d_result (4*(n*deg+1) complex128) receives [r11|r12|r21|r22] normalised,
 * *W_out the exponent (true product = result * 2^W).  On `stream` of the current device; waits for it once.  This is the
 * root's step of a sample-axis split: the block matrices arrive by RCCL and never visit the host. */
FNFT_INT fnft_amd_poly_fmult2x2_device(FNFT_UINT deg, FNFT_UINT n, const void *d_p, void *d_result, FNFT_UINT *deg_out,
                                       FNFT_INT *W_out, void *stream);
/* same, for the second (coarse) transform of Richardson extrapolation with 4SPLIT4A/B: the
 * transform uses every nskip-th step of the D samples; the band-limited resampling of
 * fnft__nse_discretization_preprocess_signal (src/private/fnft__nse_discretization.c:474-503)
 * still sees all D samples.  nskip = 1 is fnft_amd_plan_create; other schemes accept only 1. */
FNFT_INT fnft_amd_plan_create_sub(fnft_amd_plan_t **plan, FNFT_UINT D, FNFT_UINT M, FNFT_UINT batch,
                                  fnft_nse_discretization_t discretization, int device, FNFT_UINT nskip);
void fnft_amd_plan_destroy(fnft_amd_plan_t *plan);
/* Bytes of HBM the plan holds. */
FNFT_UINT fnft_amd_plan_workspace_bytes(const fnft_amd_plan_t *plan);
/* Warnings of the last finished call (after fnft_amd_plan_finish): bit 0 = the band-limited resampler of the
 * 4SPLIT4A/B front end found the signal's spectrum not decayed (the reference's "Signal does not appear to be
 * bandlimited" warning, src/private/fnft__misc.c:371-381; the drop-in fnft_nsev prints it through the printf hook). */
int fnft_amd_plan_last_warnings(const fnft_amd_plan_t *plan);
/* Measurement hook: stage i of the calling thread's last discrete-spectrum computation (the bound-state part of
 * fnft_nsev) as (name, host wall-clock ms); -1 past the last stage. */
double fnft_amd_discspec_stage_ms(FNFT_UINT i, char *name, FNFT_UINT name_cap);
/* Device the plan's workspace lives on (-1 for NULL). */
int fnft_amd_plan_device(const fnft_amd_plan_t *plan);

/* Continuous spectrum of batch signals, everything resident in HBM:
 *   d_q:        batch*D complex128 (device), signal b at d_q + b*D
 *   d_contspec: batch*cs_len complex128 (device), cs_len = M*{1,2,3} by contspec_type, layout per
 *               signal as fnft_nsev writes it ([rho], [a b] or [rho a b]).
 * stream is a hipStream_t passed as void* (NULL = default stream).  Asynchronous: returns after
 * enqueueing; fnft_amd_plan_finish() waits and returns the device-side status
 * (FNFT_SUCCESS, -FNFT_EC_DIV_BY_ZERO, -FNFT_EC_OTHER for the MODAL step-size check). */
FNFT_INT fnft_amd_nsev_contspec_device(fnft_amd_plan_t *plan, const void *d_q, void *d_contspec,
                                       const FNFT_REAL *T, const FNFT_REAL *XI, FNFT_INT kappa,
                                       fnft_nsev_cstype_t contspec_type,
                                       FNFT_INT normalization_flag, void *stream);
FNFT_INT fnft_amd_plan_finish(fnft_amd_plan_t *plan, void *stream);

/* nsev_compute_contspec (src/fnft_nsev.c:744-891) alone: continuous spectrum from a transfer matrix the caller
 * holds in DEVICE memory -- per signal [r11|r12|r21|r22], each D*deg0+1 complex128, highest power first (the
 * layout of fnft__poly_fmult2x2 / fnft_amd_plan_get_transfer_matrix_device), true matrix = stored * 2^W --
 * on the plan's M-point grid (chirp-z of entries 11 and 21, phase factors, rho / a / b).  T is the interval of
 * the plan's D samples.  Status (FNFT_EC_DIV_BY_ZERO where a(xi) = 0 exactly, src/fnft_nsev.c:850-853) through
 * fnft_amd_plan_finish.  Used by the root of the sample-axis split (fnft_amd/sharding.py). */
FNFT_INT fnft_amd_nsev_contspec_from_tm_device(fnft_amd_plan_t *plan, const void *d_tm, FNFT_INT W, void *d_contspec,
                                               const FNFT_REAL *T, const FNFT_REAL *XI,
                                               fnft_nsev_cstype_t contspec_type, void *stream);

/* Time (ms) the product tree / the chirp-z+epilogue stage took in the last finished call,
 * measured with HIP events on the stream the kernels ran on.  which: 0 = coefficients + tree,
 * 1 = chirp-z + epilogue, 2 = whole call. */
double fnft_amd_plan_last_ms(const fnft_amd_plan_t *plan, int which);
/* Enable/disable the per-stage event timers (they add two event records per stage). */
void fnft_amd_plan_set_timing(fnft_amd_plan_t *plan, int enabled);
/* Per-launch timers (measurement only; no counterpart in the reference): while enabled every kernel
 * launch of the plan is bracketed by a HIP event pair on the launch stream.  Enabling resets the
 * list; after the stream has been synchronised, launch i of the calls since then is read back as
 * (kernel name, duration in ms).  launch_ms returns -1 for an index out of range. */
void fnft_amd_plan_set_launch_timing(fnft_amd_plan_t *plan, int enabled);
FNFT_UINT fnft_amd_plan_launch_count(const fnft_amd_plan_t *plan);
double fnft_amd_plan_launch_ms(const fnft_amd_plan_t *plan, FNFT_UINT i, char *name, FNFT_UINT name_cap);

/* Transfer matrix of the last call (device pointers into the plan's workspace, signal b):
 * coefficient arrays in the reference's result layout [r11|r12|r21|r22], each deg+1, highest
 * power first, exponent W with true = stored * 2^W.  Copies to HOST buffers. */
FNFT_INT fnft_amd_plan_get_transfer_matrix(fnft_amd_plan_t *plan, FNFT_UINT b,
                                           FNFT_COMPLEX *result_host, FNFT_UINT *deg,
                                           FNFT_INT *W);
/* Same, into a DEVICE buffer of 4*(D*deg0+1) complex128 (device-to-device, no host staging of the
 * coefficients; the multi-GPU sample-axis split gathers these buffers over RCCL).  *deg, *W come back to
 * the host, so the call waits for `stream`.  A plan made with M = 0 and a call of
 * fnft_amd_nsev_contspec_device with d_contspec = NULL computes the transfer matrix only. */
FNFT_INT fnft_amd_plan_get_transfer_matrix_device(fnft_amd_plan_t *plan, FNFT_UINT b, void *d_result,
                                                  FNFT_UINT *deg, FNFT_INT *W, void *stream);

/* ======================================================================================== */
/* 4. Korteweg-de Vries equation, vanishing boundaries (include/fnft_kdvv.h)                  */
/* ======================================================================================== */

/* include/fnft_kdv_discretization_t.h:96-122 (same ordinals) */
typedef enum {
    fnft_kdv_discretization_2SPLIT1A,
    fnft_kdv_discretization_2SPLIT1B,
    fnft_kdv_discretization_2SPLIT2A,
    fnft_kdv_discretization_2SPLIT2B,
    fnft_kdv_discretization_2SPLIT2S,
    fnft_kdv_discretization_2SPLIT3A,
    fnft_kdv_discretization_2SPLIT3B,
    fnft_kdv_discretization_2SPLIT3S,
    fnft_kdv_discretization_2SPLIT4A,
    fnft_kdv_discretization_2SPLIT4B,
    fnft_kdv_discretization_2SPLIT5A,
    fnft_kdv_discretization_2SPLIT5B,
    fnft_kdv_discretization_2SPLIT6A,
    fnft_kdv_discretization_2SPLIT6B,
    fnft_kdv_discretization_2SPLIT7A,
    fnft_kdv_discretization_2SPLIT7B,
    fnft_kdv_discretization_2SPLIT8A,
    fnft_kdv_discretization_2SPLIT8B,
    fnft_kdv_discretization_4SPLIT4A,
    fnft_kdv_discretization_4SPLIT4B,
    fnft_kdv_discretization_BO,
    fnft_kdv_discretization_CF4_2,
    fnft_kdv_discretization_CF4_3,
    fnft_kdv_discretization_CF5_3,
    fnft_kdv_discretization_CF6_4
} fnft_kdv_discretization_t;

/* include/fnft_kdvv.h:60-62 */
typedef struct {
    fnft_kdv_discretization_t discretization;
} fnft_kdvv_opts_t;

/* include/fnft_kdvv.h:73, src/fnft_kdvv.c:34-44: discretization = 2SPLIT8B */
fnft_kdvv_opts_t fnft_kdvv_default_opts(void);

/* Drop-in for include/fnft_kdvv.h:100-105 / src/fnft_kdvv.c:59-123: reflection coefficient of
 * the KdV scattering problem on M points of XI.  Same argument checks in the same order; like the
 * reference, K_ptr, bound_states and normconsts_or_residues must be NULL
 * (FNFT_EC_NOT_YET_IMPLEMENTED otherwise).  The 18 2SPLIT schemes are covered. */
FNFT_INT fnft_kdvv(const FNFT_UINT D, FNFT_COMPLEX *const u, FNFT_REAL const *const T, const FNFT_UINT M,
                   FNFT_COMPLEX *const contspec, FNFT_REAL const *const XI, FNFT_UINT *const K_ptr,
                   FNFT_COMPLEX *const bound_states, FNFT_COMPLEX *const normconsts_or_residues,
                   fnft_kdvv_opts_t *opts_ptr);

/* private seam, include/private/fnft__kdv_fscatter.h:50,84-86 */
FNFT_UINT fnft__kdv_fscatter_numel(FNFT_UINT D, fnft_kdv_discretization_t discretization);
FNFT_INT fnft__kdv_fscatter(const FNFT_UINT D, FNFT_COMPLEX const *const u, const FNFT_REAL eps_t,
                            FNFT_COMPLEX *const result, FNFT_UINT *const deg_ptr, FNFT_INT *const W_ptr,
                            fnft_kdv_discretization_t discretization);

/* device-resident KdV transform: plan as in section 3 (destroy / finish / last_ms / set_timing are
 * shared); d_u: batch*D complex128, d_contspec: batch*M complex128 */
FNFT_INT fnft_amd_kdvv_plan_create(fnft_amd_plan_t **plan, FNFT_UINT D, FNFT_UINT M, FNFT_UINT batch,
                                   fnft_kdv_discretization_t discretization, int device);
FNFT_INT fnft_amd_kdvv_contspec_device(fnft_amd_plan_t *plan, const void *d_u, void *d_contspec,
                                       const FNFT_REAL *T, const FNFT_REAL *XI, void *stream);
/* A real potential (the KdV case; with r = -1 every step matrix of src/private/fnft__akns_fscatter.c:116-917 is then a
 * real polynomial matrix) takes the real-coefficient product tree: folded transforms of half the length, half the
 * bytes.  mode -1 (default): fnft_amd_kdvv_contspec_device asks the device whether u is real (one small kernel and a
 * 4-byte read-back: the call waits for the stream once before it enqueues the transform); 1: the caller guarantees a
 * real u (no check; a non-real sample is reported by fnft_amd_plan_finish as FNFT_EC_INVALID_ARGUMENT); 0: always the
 * complex path.  The host-pointer entries fnft_kdvv / fnft__kdv_fscatter look at u on the host. */
FNFT_INT fnft_amd_kdvv_plan_set_real_mode(fnft_amd_plan_t *plan, int mode);

/* ======================================================================================== */
/* 5. fnft_nsep: NSE with (quasi-)periodic boundary conditions           */
/* ======================================================================================== */

/* include/fnft_nsep.h:53-57 */
typedef enum {
    fnft_nsep_loc_SUBSAMPLE_AND_REFINE,
    fnft_nsep_loc_GRIDSEARCH,
    fnft_nsep_loc_MIXED
} fnft_nsep_loc_t;

/* include/fnft_nsep.h:73-77 */
typedef enum {
    fnft_nsep_filt_NONE,
    fnft_nsep_filt_MANUAL,
    fnft_nsep_filt_AUTO
} fnft_nsep_filt_t;

/* include/fnft_nsep.h:140-151 (same field order) */
typedef struct {
    fnft_nsep_loc_t localization;
    fnft_nsep_filt_t filtering;
    FNFT_REAL bounding_box[4];
    FNFT_UINT max_evals;
    fnft_nse_discretization_t discretization;
    FNFT_INT normalization_flag;
    FNFT_REAL floquet_range[2];
    FNFT_UINT points_per_spine;
    FNFT_UINT Dsub;
    FNFT_REAL tol;
} fnft_nsep_opts_t;

/* include/fnft_nsep.h:172, src/fnft_nsep.c:26-39: MIXED, AUTO filtering, max_evals 20, bounding box (-inf, inf)^2,
 * normalization on, 2SPLIT2A, floquet_range {-1, 1}, points_per_spine 2, Dsub 0 (automatic), tol -1 (automatic). */
fnft_nsep_opts_t fnft_nsep_default_opts(void);

/* Drop-in for include/fnft_nsep.h:263-268 / src/fnft_nsep.c:82-220: main and auxiliary spectrum of one period
 * q[0..D) (D a power of two) of a quasi-periodic signal, q(t + T1 - T0) = q(t) exp(i phase_shift).  Same argument
 * checks in the same order (sheet_indices must be NULL: FNFT_EC_NOT_YET_IMPLEMENTED otherwise), same options, same
 * localization methods: the grid search (:222-439) builds the Floquet polynomials from fnft__nse_fscatter and finds
 * their roots on the unit circle with fnft__poly_roots_fftgridsearch; subsample-and-refine (:441-706) takes all roots
 * of the polynomials of a subsampled signal (fnft__poly_roots_fasteigen) and refines them by Newton's method on
 * fnft__nse_scatter_matrix (:708-860; BO for the 2SPLIT schemes, CF4_2 for 4SPLIT4A/B) -- here all estimates advance
 * together, one scattering-matrix call per Newton stage, each with the reference's own stopping rule.  *K_ptr / *M_ptr:
 * capacity of main_spec / aux_spec in, number of points out.  The 21 fast discretizations are covered. */
FNFT_INT fnft_nsep(const FNFT_UINT D, FNFT_COMPLEX const *const q, FNFT_REAL const *const T, FNFT_REAL const phase_shift,
                   FNFT_UINT *const K_ptr, FNFT_COMPLEX *const main_spec, FNFT_UINT *const M_ptr,
                   FNFT_COMPLEX *const aux_spec, FNFT_REAL *const sheet_indices, const FNFT_INT kappa,
                   fnft_nsep_opts_t *opts);

#ifdef __cplusplus
}
#endif
#endif /* FNFT_AMD_H */
