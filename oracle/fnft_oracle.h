/*
 * fnft_oracle.h -- CPU restatement of the FNFT fnft_nsev continuous-spectrum path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (fnft_amd/, include/) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * Parity status: PINNED by fixtures.  The restatement is checked against the reference's own
 * known-answer vectors (tests/golden/reference_fixtures.json, extracted from
 * /root/reference/test and /root/reference/src/private/fnft__nsev_testcases.c by
 * tests/golden/extract_reference_fixtures.py).  The reference itself is unbuildable under this
 * project's rules (its sources need the cmake-generated fnft_config.h), so there is no
 * oracle/_ref.
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 */
#ifndef FNFT_ORACLE_H
#define FNFT_ORACLE_H

#include <complex.h>
#include <stddef.h>
#include <stdint.h>

typedef double complex orc_cplx;

/* Return codes: same numbering as include/fnft_errwarn.h:44-94 of the reference. */
#define ORC_SUCCESS 0
#define ORC_EC_NOMEM 1
#define ORC_EC_INVALID_ARGUMENT 2
#define ORC_EC_DIV_BY_ZERO 3
#define ORC_EC_OTHER 5
#define ORC_EC_NOT_YET_IMPLEMENTED 6

/* Same ordinal values as fnft_nse_discretization_t (include/fnft_nse_discretization_t.h:104-133). */
enum orc_nse_disc {
    ORC_NSE_2SPLIT2_MODAL = 0, ORC_NSE_BO, ORC_NSE_2SPLIT1A, ORC_NSE_2SPLIT1B, ORC_NSE_2SPLIT2A,
    ORC_NSE_2SPLIT2B, ORC_NSE_2SPLIT2S, ORC_NSE_2SPLIT3A, ORC_NSE_2SPLIT3B, ORC_NSE_2SPLIT3S,
    ORC_NSE_2SPLIT4A, ORC_NSE_2SPLIT4B, ORC_NSE_2SPLIT5A, ORC_NSE_2SPLIT5B, ORC_NSE_2SPLIT6A,
    ORC_NSE_2SPLIT6B, ORC_NSE_2SPLIT7A, ORC_NSE_2SPLIT7B, ORC_NSE_2SPLIT8A, ORC_NSE_2SPLIT8B,
    ORC_NSE_4SPLIT4A, ORC_NSE_4SPLIT4B
};

/* Same ordinal values as fnft__akns_discretization_t
 * (include/private/fnft__akns_discretization_t.h:104-134). */
enum orc_akns_disc {
    ORC_AKNS_2SPLIT2_MODAL = 0, ORC_AKNS_2SPLIT1A, ORC_AKNS_2SPLIT1B, ORC_AKNS_2SPLIT2A,
    ORC_AKNS_2SPLIT2B, ORC_AKNS_2SPLIT2S, ORC_AKNS_2SPLIT3A, ORC_AKNS_2SPLIT3B, ORC_AKNS_2SPLIT3S,
    ORC_AKNS_2SPLIT4A, ORC_AKNS_2SPLIT4B, ORC_AKNS_2SPLIT5A, ORC_AKNS_2SPLIT5B, ORC_AKNS_2SPLIT6A,
    ORC_AKNS_2SPLIT6B, ORC_AKNS_2SPLIT7A, ORC_AKNS_2SPLIT7B, ORC_AKNS_2SPLIT8A, ORC_AKNS_2SPLIT8B,
    ORC_AKNS_BO, ORC_AKNS_4SPLIT4A, ORC_AKNS_4SPLIT4B
};

/* contspec_type values of fnft_nsev_cstype_t (include/fnft_nsev.h:130-134). */
enum orc_cstype { ORC_CS_RHO = 0, ORC_CS_AB = 1, ORC_CS_BOTH = 2 };

/* --- FFT (kiss_fft.c:396-408 length policy; fnft__fft_wrapper.h:124-137 semantics) --- */
size_t orc_next_fast_size(size_t n);
/* out[k] = sum_n in[n] exp(sign*2*pi*i*n*k/len), sign = -1 forward, +1 inverse, no 1/len. */
int orc_fft(size_t len, const orc_cplx *in, orc_cplx *out, int sign);
/* any length (Bluestein when len has prime factors other than 2, 3, 5) */
int orc_dft(size_t len, const orc_cplx *in, orc_cplx *out, int sign);
/* fnft__misc.c:326-407 */
int orc_misc_resample(size_t D, double eps_t, const orc_cplx *q, double delta, orc_cplx *q_new);
/* fnft__nse_discretization.c:386-656 for the splitting schemes: q_pre has Dsub*upsampling entries */
int orc_nse_preprocess(size_t D, const orc_cplx *q, double eps_t, size_t *Dsub_ptr, orc_cplx **q_pre,
                       size_t *first_last, int nse_disc);
size_t orc_nse_upsampling(int nse_disc);

/* --- polynomial helpers --- */
/* fnft__poly_eval.c:25-53: z[i] <- p(z[i]), coefficients highest power first. */
int orc_poly_eval(size_t deg, const orc_cplx *p, size_t nz, orc_cplx *z);
/* fnft__misc.c:41-51 */
double orc_rel_err(size_t len, const orc_cplx *numer, const orc_cplx *exact);
/* fnft__misc.c:316-324 */
size_t orc_nextpowerof2(size_t n);

/* fnft__poly_fmult.c:40-43 */
size_t orc_poly_fmult2x2_numel(size_t deg, size_t n);
/* fnft__poly_fmult.c:381-546 */
int orc_poly_fmult2x2(size_t *d, size_t n, orc_cplx *p, orc_cplx *result, int32_t *W_ptr);

/* fnft__poly_chirpz.c:33-105 */
int orc_poly_chirpz(size_t deg, const orc_cplx *p, orc_cplx A, orc_cplx W, size_t M,
                    orc_cplx *result);
/* same, with A and W passed as {re, im} pointers (for FFI callers without complex-by-value) */
int orc_poly_chirpz_p(size_t deg, const orc_cplx *p, const double *A, const double *W, size_t M,
                      orc_cplx *result);

/* fnft__akns_discretization.c:29-67 ; 0 = unsupported by this restatement */
size_t orc_akns_degree(int akns_disc);
/* fnft__nse_discretization.c:108-... mapping; returns -1 if unsupported */
int orc_nse_to_akns(int nse_disc);

/* fnft__akns_fscatter.c:34-42 */
size_t orc_akns_fscatter_numel(size_t D, int akns_disc);
/* fnft__akns_fscatter.c:64-925 (schemes of degree <= 4) */
int orc_akns_fscatter(size_t D, const orc_cplx *q, const orc_cplx *r, double eps_t,
                      orc_cplx *result, size_t *deg_ptr, int32_t *W_ptr, int akns_disc);
/* per-sample coefficient stage only (the loop before the tree): p has 4*D*(deg+1) entries */
int orc_akns_coeffs(size_t D, const orc_cplx *q, const orc_cplx *r, double eps_t,
                    orc_cplx *p, int akns_disc);

/* fnft__nse_fscatter.c:34-91 */
size_t orc_nse_fscatter_numel(size_t D, int nse_disc);
int orc_nse_fscatter(size_t D, const orc_cplx *q, double eps_t, int kappa, orc_cplx *result,
                     size_t *deg_ptr, int32_t *W_ptr, int nse_disc);

/* fnft_nsev.c:744-891 (fast discretizations only) */
int orc_nsev_contspec(size_t deg, int32_t W, const orc_cplx *transfer_matrix, const double *T,
                      size_t D, const double *XI, size_t M, orc_cplx *result, int nse_disc,
                      int contspec_type);

/* fnft_nsev.c:133-453 restricted to: contspec only (bound_states == NULL),
 * no Richardson extrapolation, discretizations with upsampling factor 1. */
int orc_fnft_nsev(size_t D, const orc_cplx *q, const double *T, size_t M, orc_cplx *contspec,
                  const double *XI, int kappa, int nse_disc, int contspec_type,
                  int normalization_flag);

/* same with Richardson extrapolation of the continuous spectrum (fnft_nsev.c:316-406) */
int orc_fnft_nsev_ex(size_t D, const orc_cplx *q, const double *T, size_t M, orc_cplx *contspec,
                     const double *XI, int kappa, int nse_disc, int contspec_type,
                     int normalization_flag, int richardson_flag);

/* wall-clock seconds spent in the last orc_fnft_nsev call: [0] fscatter, [1] contspec */
void orc_last_timings(double out[2]);

/* ---- fnft_kdvv (src/fnft_kdvv.c), include/fnft_kdv_discretization_t.h:96-122 ordinals ---- */
enum {
    ORC_KDV_2SPLIT1A = 0, ORC_KDV_2SPLIT1B, ORC_KDV_2SPLIT2A, ORC_KDV_2SPLIT2B, ORC_KDV_2SPLIT2S,
    ORC_KDV_2SPLIT3A, ORC_KDV_2SPLIT3B, ORC_KDV_2SPLIT3S, ORC_KDV_2SPLIT4A, ORC_KDV_2SPLIT4B,
    ORC_KDV_2SPLIT5A, ORC_KDV_2SPLIT5B, ORC_KDV_2SPLIT6A, ORC_KDV_2SPLIT6B, ORC_KDV_2SPLIT7A,
    ORC_KDV_2SPLIT7B, ORC_KDV_2SPLIT8A, ORC_KDV_2SPLIT8B
};
int orc_kdv_to_akns(int kdv_disc);
size_t orc_kdv_fscatter_numel(size_t D, int kdv_disc);
int orc_kdv_fscatter(size_t D, const orc_cplx *u, double eps_t, orc_cplx *result, size_t *deg_ptr,
                     int32_t *W_ptr, int kdv_disc);
int orc_fnft_kdvv(size_t D, const orc_cplx *u, const double *T, size_t M, orc_cplx *contspec,
                  const double *XI, int kdv_disc);

/* ---- discrete spectrum helpers (outer logic lives in oracle/oracle.py) ---- */
int orc_nse_scatter_bound_states(size_t D, const orc_cplx *q, const double *T, size_t K,
                                 const orc_cplx *lam, orc_cplx *a_vals, orc_cplx *aprime_vals,
                                 orc_cplx *b_vals, int ups, int skip_b);
int orc_nse_scatter_matrix(size_t D, const orc_cplx *q, double eps_t, int kappa, size_t K, const orc_cplx *lam,
                           orc_cplx *result, int ups, int derivative);
double orc_l2norm2(size_t N, const orc_cplx *Z, double a, double b);

#endif
