/*
 * fnft_nsep_host.c -- C driver behind fnft_nsep (include/fnft_amd.h section 5): the nonlinear Fourier transform of the
 * NSE with (quasi-)periodic boundary conditions, main and auxiliary spectrum.
 *
 * Mirrors the reference's src/fnft_nsep.c (argument checks and their order :100-119, phase-shift removal :121-141,
 * localization switch :142-202, grid search :222-439, subsample-and-refine :441-706, Newton refinements :708-860,
 * automatic bounding box :862-) on the library's own GPU seams:
 *     fnft__nse_fscatter                  transfer matrix of the (sub)sampled period       (product tree on the GPU)
 *     fnft__poly_roots_fftgridsearch      roots of the Floquet polynomials on the unit circle (chirp z-transforms)
 *     fnft__poly_roots_fasteigen          all roots of the subsampled polynomials           (Ehrlich-Aberth kernels)
 *     fnft__nse_scatter_matrix            monodromy matrix and its lambda-derivative        (chunk-parallel scatterer)
 *     fnft__misc_resample                 band-limited resampling of the 4SPLIT4A/B schemes
 * One structural difference: the reference refines one spectral point after the other, each with up to max_evals
 * sequential scattering-matrix evaluations; here all points of a set advance together -- one batched
 * fnft__nse_scatter_matrix call per Newton stage -- while every point keeps the reference's own evaluation count and
 * stopping rule, so the numbers are the ones the sequential loops produce.  No CPU fallback.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fnft_amd.h"

FNFT_INT fnft_amd__raise(FNFT_INT ec, const char *func, int line, const char *msg);
void fnft_amd__warn(const char *msg, const char *func, int line);

#define E_INVALID_ARGUMENT(name) fnft_amd__raise(FNFT_EC_INVALID_ARGUMENT, __func__, __LINE__, "Invalid argument " #name ".")
#define E_NOT_YET_IMPLEMENTED(name, msg) \
    fnft_amd__raise(FNFT_EC_NOT_YET_IMPLEMENTED, __func__, __LINE__, "Not yet implemented (" #name "). " msg)
#define E_SUBROUTINE(ec) fnft_amd__raise(-abs(ec), __func__, __LINE__, "Subroutine failure.")
#define E_NOMEM fnft_amd__raise(FNFT_EC_NOMEM, __func__, __LINE__, "Out of memory.")
#define E_DIV_BY_ZERO fnft_amd__raise(FNFT_EC_DIV_BY_ZERO, __func__, __LINE__, "Division by zero.")
#define E_ASSERTION_FAILED fnft_amd__raise(FNFT_EC_ASSERTION_FAILED, __func__, __LINE__, "Assertion failed.")
#define E_OTHER(msg) fnft_amd__raise(FNFT_EC_OTHER, __func__, __LINE__, msg)
#define CHECK(rc, label)                 \
    do {                                 \
        if ((rc) != FNFT_SUCCESS) {      \
            (rc) = E_SUBROUTINE(rc);     \
            goto label;                  \
        }                                \
    } while (0)

static const FNFT_UINT kOversampling = 32;   /* src/fnft_nsep.c:41 */
static const char *const kTooManyMain = "Found more than *K_ptr main spectrum points. Returning as many as possible.";
static const char *const kTooManyAux = "Found more than *M_ptr aux spectrum points. Returning as many as possible.";

/* src/fnft_nsep.c:26-39 */
static fnft_nsep_opts_t default_opts = {
    .localization = fnft_nsep_loc_MIXED,
    .filtering = fnft_nsep_filt_AUTO,
    .bounding_box = {-INFINITY, INFINITY, -INFINITY, INFINITY},
    .max_evals = 20,
    .discretization = fnft_nse_discretization_2SPLIT2A,
    .normalization_flag = 1,
    .floquet_range = {-1, 1},
    .points_per_spine = 2,
    .Dsub = 0,
    .tol = -1};

fnft_nsep_opts_t fnft_nsep_default_opts(void) { return default_opts; }

/* degree per step and samples per step of the fast discretizations (src/private/fnft__akns_discretization.c:29-67,
 * 114-152 through fnft__nse_discretization.c:29-40); 0: not a discretization this library transforms */
static FNFT_UINT disc_degree(fnft_nse_discretization_t d)
{
    switch (d) {
    case fnft_nse_discretization_2SPLIT2_MODAL: case fnft_nse_discretization_2SPLIT1A: case fnft_nse_discretization_2SPLIT1B:
    case fnft_nse_discretization_2SPLIT2A: case fnft_nse_discretization_2SPLIT2B: case fnft_nse_discretization_2SPLIT2S:
        return 1;
    case fnft_nse_discretization_2SPLIT3S: case fnft_nse_discretization_2SPLIT4B: case fnft_nse_discretization_4SPLIT4B:
        return 2;
    case fnft_nse_discretization_2SPLIT3A: case fnft_nse_discretization_2SPLIT3B: return 3;
    case fnft_nse_discretization_2SPLIT4A: case fnft_nse_discretization_4SPLIT4A: return 4;
    case fnft_nse_discretization_2SPLIT6B: return 6;
    case fnft_nse_discretization_2SPLIT6A: case fnft_nse_discretization_2SPLIT8B: return 12;
    case fnft_nse_discretization_2SPLIT5A: case fnft_nse_discretization_2SPLIT5B: return 15;
    case fnft_nse_discretization_2SPLIT8A: return 24;
    case fnft_nse_discretization_2SPLIT7A: case fnft_nse_discretization_2SPLIT7B: return 105;
    default: return 0;
    }
}
static FNFT_UINT disc_upsampling(fnft_nse_discretization_t d)
{
    if (d == fnft_nse_discretization_4SPLIT4A || d == fnft_nse_discretization_4SPLIT4B) return 2;
    return disc_degree(d) ? 1 : 0;
}

/* fnft__nse_discretization_preprocess_signal (src/private/fnft__nse_discretization.c:386-656) for the fast schemes:
 * every nskip-th sample (:466-473), or, 4SPLIT4A/B, two band-limited resamplings at -/+ sqrt(3)/6 of the kept step
 * combined with the CF4_2 weights (:474-503).  *Dsub_ptr: wish in, actual out; out: ups * Dsub samples (malloc'd). */
static FNFT_INT preprocess(FNFT_UINT D, const FNFT_COMPLEX *q, FNFT_REAL eps_t, FNFT_UINT *Dsub_ptr, FNFT_COMPLEX **out,
                           FNFT_UINT first_last[2], fnft_nse_discretization_t disc)
{
    FNFT_UINT Dsub = *Dsub_ptr;
    if (Dsub < 2) Dsub = 2;
    if (Dsub > D) Dsub = D;
    const FNFT_UINT nskip = (FNFT_UINT)llround((double)D / (double)Dsub);
    Dsub = (FNFT_UINT)llround((double)D / (double)nskip);
    const FNFT_UINT ups = disc_upsampling(disc);
    if (ups == 0) return E_INVALID_ARGUMENT(discretization);
    FNFT_COMPLEX *qp = malloc(Dsub * ups * sizeof(FNFT_COMPLEX));
    if (qp == NULL) return E_NOMEM;
    FNFT_INT rc = FNFT_SUCCESS;
    if (ups == 1) {
        for (FNFT_UINT i = 0; i < Dsub; i++) qp[i] = q[i * nskip];
    } else {
        FNFT_COMPLEX *q1 = malloc(D * sizeof(FNFT_COMPLEX)), *q2 = malloc(D * sizeof(FNFT_COMPLEX));
        if (q1 == NULL || q2 == NULL) rc = E_NOMEM;
        const FNFT_REAL delta = eps_t * (sqrt(3.0) / 6.0) * (FNFT_REAL)nskip;
        if (rc == FNFT_SUCCESS) rc = fnft__misc_resample(D, eps_t, q, -delta, q1);
        if (rc == FNFT_SUCCESS) rc = fnft__misc_resample(D, eps_t, q, delta, q2);
        if (rc == FNFT_SUCCESS) {
            const FNFT_REAL w0 = 0.25 + sqrt(3.0) / 6.0, w1 = 0.25 - sqrt(3.0) / 6.0;   /* CF4_2 weights */
            for (FNFT_UINT is = 0, i = 0; is < Dsub; is++, i += nskip) {
                qp[2 * is] = w0 * q1[i] + w1 * q2[i];
                qp[2 * is + 1] = w1 * q1[i] + w0 * q2[i];
            }
        }
        free(q1);
        free(q2);
        if (rc != FNFT_SUCCESS) {
            free(qp);
            return rc;
        }
    }
    first_last[0] = 0;
    first_last[1] = (Dsub - 1) * nskip;
    *Dsub_ptr = Dsub;
    *out = qp;
    return FNFT_SUCCESS;
}

/* fnft__nse_discretization_z_to_lambda (src/private/fnft__akns_discretization.c:224-239) */
static void z_to_lambda(FNFT_UINT n, FNFT_REAL eps_t, FNFT_COMPLEX *vals, fnft_nse_discretization_t disc)
{
    const FNFT_REAL deg1 = (FNFT_REAL)(disc_degree(disc) * disc_upsampling(disc));
    for (FNFT_UINT i = 0; i < n; i++) vals[i] = clog(vals[i]) / (2.0 * I * eps_t / deg1);
}

/* misc_filter (src/private/fnft__misc.c:114-157): keep the values inside the box (NaNs are outside) */
static FNFT_INT filter_box(FNFT_UINT *n_ptr, FNFT_COMPLEX *vals, const FNFT_REAL box[4])
{
    if (!(box[0] <= box[1]) || !(box[2] <= box[3])) return E_INVALID_ARGUMENT(bounding_box);
    FNFT_UINT kept = 0;
    for (FNFT_UINT i = 0; i < *n_ptr; i++) {
        const FNFT_REAL re = creal(vals[i]), im = cimag(vals[i]);
        if (!(re >= box[0]) || !(re <= box[1]) || !(im >= box[2]) || !(im <= box[3])) continue;
        vals[kept++] = vals[i];
    }
    *n_ptr = kept;
    return FNFT_SUCCESS;
}
/* misc_filter_nonreal (:205-226): keep the values with |Im| > tol_im */
static void filter_nonreal(FNFT_UINT *n_ptr, FNFT_COMPLEX *vals, FNFT_REAL tol_im)
{
    FNFT_UINT kept = 0;
    for (FNFT_UINT i = 0; i < *n_ptr; i++)
        if (fabs(cimag(vals[i])) > tol_im) vals[kept++] = vals[i];
    *n_ptr = kept;
}

/* src/fnft_nsep.c:862-: |Re lambda| <= 0.9 pi/(|map_coeff| eps_t), |Im lambda| <= -log(0.1)/(|map_coeff| eps_t) */
static void update_bounding_box_if_auto(FNFT_REAL eps_t, FNFT_REAL map_coeff, fnft_nsep_opts_t *o)
{
    if (o->filtering != fnft_nsep_filt_AUTO) return;
    o->bounding_box[1] = 0.9 * 3.14159265358979323846 / (fabs(map_coeff) * eps_t);
    o->bounding_box[0] = -o->bounding_box[1];
    o->bounding_box[3] = -log(0.1) / (fabs(map_coeff) * eps_t);
    o->bounding_box[2] = -o->bounding_box[3];
}

/* ---- Newton refinements, all points together ------------------------------------------------------------------------ */
/* src/fnft_nsep.c:708-792: main spectrum, f = a + a~ + rhs; Newton's method for roots of order m = 1 or 2, the better
 * of the two trial points is taken.  Per point the reference's loop
 *     evaluate at lam; for (nevals = 1; nevals <= max_evals;) { two trial evaluations (the second is skipped when the
 *     first is below tol); step; if below tol: one last first-order step and stop }
 * is run as a state machine; a stage evaluates the pending point of every unfinished estimate in one call. */
static FNFT_INT refine_mainspec(FNFT_UINT D, const FNFT_COMPLEX *q, FNFT_REAL eps_t, FNFT_UINT K, FNFT_COMPLEX *ms,
                                FNFT_UINT max_evals, FNFT_REAL rhs, FNFT_REAL tol, FNFT_INT kappa,
                                fnft_nse_discretization_t disc)
{
    if (max_evals == 0 || K == 0) return FNFT_SUCCESS;
    FNFT_INT rc = FNFT_SUCCESS;
    FNFT_COMPLEX *lam = malloc(K * sizeof(FNFT_COMPLEX)), *S = malloc(8 * K * sizeof(FNFT_COMPLEX));
    FNFT_COMPLEX *next_f = malloc(K * sizeof(FNFT_COMPLEX)), *next_fp = malloc(K * sizeof(FNFT_COMPLEX));
    FNFT_COMPLEX *incr = malloc(K * sizeof(FNFT_COMPLEX));
    FNFT_REAL *min_abs = malloc(K * sizeof(FNFT_REAL));
    FNFT_UINT *nevals = malloc(K * sizeof(FNFT_UINT)), *idx = malloc(K * sizeof(FNFT_UINT)), *best_m = malloc(K * sizeof(FNFT_UINT));
    char *state = malloc(K);   /* 0: iterating, 1: finished */
    if (!lam || !S || !next_f || !next_fp || !incr || !min_abs || !nevals || !idx || !best_m || !state) {
        rc = E_NOMEM;
        goto done;
    }
    rc = fnft__nse_scatter_matrix(D, q, NULL, eps_t, kappa, K, ms, S, disc, 1);
    CHECK(rc, done);
    for (FNFT_UINT k = 0; k < K; k++) {
        next_f[k] = S[8 * k] + S[8 * k + 3] + rhs;      /* f = a(lam) + atil(lam) + rhs */
        next_fp[k] = S[8 * k + 4] + S[8 * k + 7];       /* f' */
        nevals[k] = 1;
        state[k] = 0;
    }
    for (;;) {
        /* estimates whose loop condition nevals <= max_evals still holds start an iteration */
        FNFT_UINT na = 0;
        for (FNFT_UINT k = 0; k < K; k++) {
            if (state[k]) continue;
            if (nevals[k] > max_evals) { state[k] = 1; continue; }
            if (next_fp[k] == 0.0) { rc = E_DIV_BY_ZERO; goto done; }
            incr[k] = next_f[k] / next_fp[k];
            min_abs[k] = INFINITY;
            best_m[k] = 1;
            idx[na] = k;
            lam[na++] = ms[k] - incr[k];                /* m = 1 */
        }
        if (na == 0) break;
        for (FNFT_UINT m = 1; m <= 2 && na > 0; m++) {
            rc = fnft__nse_scatter_matrix(D, q, NULL, eps_t, kappa, na, lam, S, disc, 1);
            CHECK(rc, done);
            FNFT_UINT nb = 0;
            for (FNFT_UINT t = 0; t < na; t++) {
                const FNFT_UINT k = idx[t];
                nevals[k]++;
                const FNFT_COMPLEX tmp = S[8 * t] + S[8 * t + 3] + rhs;
                const FNFT_REAL cur = cabs(tmp);
                int below = 0;
                if (cur < min_abs[k]) {
                    min_abs[k] = cur;
                    best_m[k] = m;
                    next_f[k] = tmp;
                    next_fp[k] = S[8 * t + 4] + S[8 * t + 7];
                    below = cur < tol;
                }
                if (m == 1 && !below) {                 /* the second trial point, m = 2 */
                    idx[nb] = k;
                    lam[nb++] = ms[k] - 2.0 * incr[k];
                }
            }
            na = (m == 1) ? nb : 0;
        }
        for (FNFT_UINT k = 0; k < K; k++) {
            if (state[k] || isinf(min_abs[k])) continue;
            ms[k] -= (FNFT_REAL)best_m[k] * incr[k];    /* Newton step */
            if (min_abs[k] < tol) {                      /* one last first-order step with the values already known */
                if (next_fp[k] == 0.0) { rc = E_DIV_BY_ZERO; goto done; }
                ms[k] -= next_f[k] / next_fp[k];
                state[k] = 1;
            }
            min_abs[k] = INFINITY;
        }
    }
done:
    free(lam); free(S); free(next_f); free(next_fp); free(incr); free(min_abs); free(nevals); free(idx); free(best_m);
    free(state);
    return rc;
}

/* src/fnft_nsep.c:794-836: auxiliary spectrum, plain Newton on b(lambda) = S12; the step is taken before the test, so a
 * point below tol still gets its last step */
static FNFT_INT refine_auxspec(FNFT_UINT D, const FNFT_COMPLEX *q, FNFT_REAL eps_t, FNFT_UINT K, FNFT_COMPLEX *aux,
                               FNFT_UINT max_evals, FNFT_REAL tol, FNFT_INT kappa, fnft_nse_discretization_t disc)
{
    if (max_evals == 0 || K == 0) return FNFT_SUCCESS;
    FNFT_INT rc = FNFT_SUCCESS;
    FNFT_COMPLEX *lam = malloc(K * sizeof(FNFT_COMPLEX)), *S = malloc(8 * K * sizeof(FNFT_COMPLEX));
    FNFT_UINT *idx = malloc(K * sizeof(FNFT_UINT));
    char *done_flag = calloc(K, 1);
    if (!lam || !S || !idx || !done_flag) {
        rc = E_NOMEM;
        goto done;
    }
    for (FNFT_UINT nevals = 0; nevals < max_evals; nevals++) {
        FNFT_UINT na = 0;
        for (FNFT_UINT k = 0; k < K; k++)
            if (!done_flag[k]) { idx[na] = k; lam[na++] = aux[k]; }
        if (na == 0) break;
        rc = fnft__nse_scatter_matrix(D, q, NULL, eps_t, kappa, na, lam, S, disc, 1);
        CHECK(rc, done);
        for (FNFT_UINT t = 0; t < na; t++) {
            const FNFT_UINT k = idx[t];
            const FNFT_COMPLEX f = S[8 * t + 1], fp = S[8 * t + 5];   /* b, b' */
            if (fp == 0.0) { rc = E_DIV_BY_ZERO; goto done; }
            aux[k] -= f / fp;
            if (cabs(f) < tol) done_flag[k] = 1;
        }
    }
done:
    free(lam); free(S); free(idx); free(done_flag);
    return rc;
}

/* ---- src/fnft_nsep.c:222-439 ---------------------------------------------------------------------------------------- */
static FNFT_INT gridsearch(FNFT_UINT D, const FNFT_COMPLEX *q, const FNFT_REAL *T, FNFT_UINT *K_ptr, FNFT_COMPLEX *main_spec,
                           FNFT_UINT *M_ptr, FNFT_COMPLEX *aux_spec, FNFT_INT kappa, fnft_nsep_opts_t *o, FNFT_INT warn_flags[2])
{
    FNFT_COMPLEX *tm = NULL, *p = NULL, *roots = NULL, *qp = NULL;
    FNFT_REAL PHI[2];
    FNFT_UINT deg = 0, K = 0, M = 0, first_last[2] = {0, 0};
    FNFT_INT W = 0, rc = FNFT_SUCCESS;
    const FNFT_UINT ups = disc_upsampling(o->discretization);
    if (ups == 0) return E_INVALID_ARGUMENT(opts->discretization);
    const FNFT_UINT Deff = D * ups;
    const FNFT_REAL eps_t = (T[1] - T[0]) / (FNFT_REAL)D;
    FNFT_UINT Dsub = D;
    rc = preprocess(D, q, eps_t, &Dsub, &qp, first_last, o->discretization);
    CHECK(rc, release);
    const FNFT_UINT numel = fnft__nse_fscatter_numel(Deff, o->discretization);
    if (numel == 0) { rc = E_INVALID_ARGUMENT(opts_ptr->discretization); goto release; }
    tm = malloc(numel * sizeof(FNFT_COMPLEX));
    if (tm == NULL) { rc = E_NOMEM; goto release; }
    rc = fnft__nse_fscatter(Deff, qp, eps_t, kappa, tm, &deg, o->normalization_flag ? &W : NULL, o->discretization);
    CHECK(rc, release);
    /* arc of the unit circle that belongs to the bounding box, :287-298 */
    const FNFT_REAL map_coeff = 2.0 / (FNFT_REAL)disc_degree(o->discretization);
    update_bounding_box_if_auto(eps_t, map_coeff, o);
    PHI[0] = map_coeff * eps_t * o->bounding_box[0];
    PHI[1] = map_coeff * eps_t * o->bounding_box[1];
    if (PHI[0] > PHI[1]) { const FNFT_REAL t = PHI[0]; PHI[0] = PHI[1]; PHI[1] = t; }
    roots = malloc(kOversampling * deg * sizeof(FNFT_COMPLEX));
    if (roots == NULL) { rc = E_NOMEM; goto release; }
    if (main_spec != NULL) {
        /* p(z) ~ z^(deg/2) (Delta(z) -/+ 2), Delta = trace of the monodromy matrix, :310-325 */
        p = malloc((deg + 1) * sizeof(FNFT_COMPLEX));
        if (p == NULL) { rc = E_NOMEM; goto release; }
        for (FNFT_UINT i = 0; i <= deg; i++) p[i] = tm[i] + conj(tm[deg - i]);
        const FNFT_REAL unit = pow(2.0, -(FNFT_REAL)W);   /* nse_fscatter rescales */
        FNFT_UINT Kfound = 0;
        for (int sign = 0; sign < 2; sign++) {           /* +2, then -2 */
            p[deg / 2] += (sign == 0 ? 2.0 : -4.0) * unit;
            FNFT_UINT Kn = kOversampling * deg;
            rc = fnft__poly_roots_fftgridsearch(deg, p, &Kn, PHI, roots);
            CHECK(rc, release);
            if (Kn > deg) { rc = E_OTHER("Found more roots than memory is available."); goto release; }
            z_to_lambda(Kn, eps_t, roots, o->discretization);
            if (o->filtering != fnft_nsep_filt_NONE) {
                rc = filter_box(&Kn, roots, o->bounding_box);
                CHECK(rc, release);
            }
            if (Kfound + Kn > *K_ptr) {
                if (warn_flags[0] == 0) { fnft_amd__warn(kTooManyMain, __func__, __LINE__); warn_flags[0] = 1; }
                Kn = (Kfound < *K_ptr) ? *K_ptr - Kfound : 0;
            }
            memcpy(main_spec + Kfound, roots, Kn * sizeof(FNFT_COMPLEX));
            Kfound += Kn;
        }
        K = Kfound;
    }
    if (aux_spec != NULL) {   /* roots of b(z), the 12 entry, on the real line, :400-425 */
        M = kOversampling * deg;
        rc = fnft__poly_roots_fftgridsearch(deg, tm + (deg + 1), &M, PHI, roots);
        CHECK(rc, release);
        z_to_lambda(M, eps_t, roots, o->discretization);
        if (o->filtering != fnft_nsep_filt_NONE) {
            rc = filter_box(&M, roots, o->bounding_box);
            CHECK(rc, release);
        }
        if (M > *M_ptr) {
            if (warn_flags[1] == 0) { fnft_amd__warn(kTooManyAux, __func__, __LINE__); warn_flags[1] = 1; }
            M = *M_ptr;
        }
        memcpy(aux_spec, roots, M * sizeof(FNFT_COMPLEX));
    }
    *K_ptr = K;
    *M_ptr = M;
release:
    free(tm); free(p); free(roots); free(qp);
    return rc;
}

/* ---- src/fnft_nsep.c:441-706 ---------------------------------------------------------------------------------------- */
static FNFT_INT subsample_and_refine(FNFT_UINT D, const FNFT_COMPLEX *q, const FNFT_REAL *T, FNFT_UINT *K_ptr,
                                     FNFT_COMPLEX *main_spec, FNFT_UINT *M_ptr, FNFT_COMPLEX *aux_spec, FNFT_INT kappa,
                                     fnft_nsep_opts_t *o, FNFT_INT skip_real_flag, FNFT_INT warn_flags[2])
{
    FNFT_COMPLEX *tm = NULL, *p = NULL, *roots = NULL, *qfull = NULL, *qsub = NULL;
    FNFT_UINT deg = 0, K = 0, M = 0, first_last[2] = {0, 0};
    FNFT_INT W = 0, rc = FNFT_SUCCESS;
    const FNFT_UINT ups = disc_upsampling(o->discretization);
    if (ups == 0) return E_INVALID_ARGUMENT(opts->discretization);
    const FNFT_UINT Deff = D * ups;
    const FNFT_REAL eps_t = (T[1] - T[0]) / (FNFT_REAL)D;
    /* the signal the estimates are refined on, :479-483 */
    FNFT_UINT Dsub = D;
    rc = preprocess(D, q, eps_t, &Dsub, &qfull, first_last, o->discretization);
    CHECK(rc, release);
    /* the subsampled signal the estimates come from, :485-494: a power of two */
    Dsub = o->Dsub;
    if (Dsub == 0) Dsub = (FNFT_UINT)pow(2.0, ceil(0.5 * log2((double)D * log2((double)D) * log2((double)D))));
    else Dsub = (FNFT_UINT)pow(2.0, round(log2((double)Dsub)));
    rc = preprocess(D, q, eps_t, &Dsub, &qsub, first_last, o->discretization);
    CHECK(rc, release);
    const FNFT_UINT nskip = D / Dsub;
    if (first_last[0] != 0 || first_last[1] + nskip != D) { rc = E_ASSERTION_FAILED; goto release; }
    const fnft_nse_discretization_t slow = (ups == 2) ? fnft_nse_discretization_CF4_2 : fnft_nse_discretization_BO;
    const FNFT_REAL refine_tol = (o->tol < 0) ? sqrt(2.220446049250313e-16) : o->tol;
    const FNFT_UINT numel = fnft__nse_fscatter_numel(Dsub * ups, o->discretization);
    if (numel == 0) { rc = E_INVALID_ARGUMENT(opts_ptr->discretization); goto release; }
    tm = malloc(numel * sizeof(FNFT_COMPLEX));
    if (tm == NULL) { rc = E_NOMEM; goto release; }
    const FNFT_REAL eps_t_sub = (FNFT_REAL)nskip * eps_t;
    rc = fnft__nse_fscatter(Dsub * ups, qsub, eps_t_sub, kappa, tm, &deg, o->normalization_flag ? &W : NULL,
                            o->discretization);
    CHECK(rc, release);
    const FNFT_REAL map_coeff = 2.0 / (FNFT_REAL)disc_degree(o->discretization);
    update_bounding_box_if_auto(eps_t_sub, map_coeff, o);
    const FNFT_REAL tol_im = (o->bounding_box[1] - o->bounding_box[0]) / (FNFT_REAL)(kOversampling * (D - 1));
    roots = malloc((deg + 1) * sizeof(FNFT_COMPLEX));
    if (roots == NULL) { rc = E_NOMEM; goto release; }
    if (main_spec != NULL) {
        p = malloc((deg + 1) * sizeof(FNFT_COMPLEX));
        if (p == NULL) { rc = E_NOMEM; goto release; }
        for (FNFT_UINT i = 0; i <= deg; i++) p[i] = tm[i] + conj(tm[deg - i]);
        /* Delta(z) = rhs for a grid of rhs between 2 floquet_range[0] and 2 floquet_range[1]: the end points give the main
         * spectrum, more points per spine trace the spines, :571-581 */
        const FNFT_REAL rhs_0 = o->floquet_range[0], rhs_1 = o->floquet_range[1];
        const FNFT_UINT nvals = o->points_per_spine;
        FNFT_REAL rhs_step = rhs_1 - rhs_0;
        if (nvals > 1) rhs_step /= (FNFT_REAL)(nvals - 1);
        const FNFT_COMPLEX center = p[deg / 2];
        const FNFT_REAL unit = pow(2.0, -(FNFT_REAL)W);
        for (FNFT_UINT nval = 0; nval < nvals; nval++) {
            const FNFT_REAL rhs = 2.0 * (rhs_0 + (FNFT_REAL)nval * rhs_step);
            p[deg / 2] = center - rhs * unit;
            rc = fnft__poly_roots_fasteigen(deg, p, roots);
            CHECK(rc, release);
            z_to_lambda(deg, eps_t_sub, roots, o->discretization);
            FNFT_UINT Kn = deg;
            if (o->filtering != fnft_nsep_filt_NONE) {
                rc = filter_box(&Kn, roots, o->bounding_box);
                CHECK(rc, release);
            }
            if (skip_real_flag != 0) filter_nonreal(&Kn, roots, tol_im);
            rc = refine_mainspec(Deff, qfull, eps_t, Kn, roots, o->max_evals, -rhs, refine_tol, kappa, slow);
            CHECK(rc, release);
            if (o->filtering != fnft_nsep_filt_NONE) {
                rc = filter_box(&Kn, roots, o->bounding_box);
                CHECK(rc, release);
            }
            if (skip_real_flag != 0) filter_nonreal(&Kn, roots, tol_im);
            if (K + Kn > *K_ptr) {
                if (warn_flags[0] == 0) { fnft_amd__warn(kTooManyMain, __func__, __LINE__); warn_flags[0] = 1; }
                Kn = *K_ptr - K;
            }
            memcpy(main_spec + K, roots, Kn * sizeof(FNFT_COMPLEX));
            K += Kn;
            if (warn_flags[0] == 1) break;   /* the caller's array is full */
        }
    }
    if (aux_spec != NULL) {
        rc = fnft__poly_roots_fasteigen(deg, tm + (deg + 1), roots);
        CHECK(rc, release);
        M = deg;
        z_to_lambda(M, eps_t_sub, roots, o->discretization);
        if (o->filtering != fnft_nsep_filt_NONE) {
            rc = filter_box(&M, roots, o->bounding_box);
            CHECK(rc, release);
        }
        rc = refine_auxspec(Deff, qfull, eps_t, M, roots, o->max_evals, refine_tol, kappa, slow);
        CHECK(rc, release);
        if (o->filtering != fnft_nsep_filt_NONE) {
            rc = filter_box(&M, roots, o->bounding_box);
            CHECK(rc, release);
        }
        if (skip_real_flag != 0) filter_nonreal(&M, roots, tol_im);
        if (M > *M_ptr) {
            if (warn_flags[1] == 0) { fnft_amd__warn(kTooManyAux, __func__, __LINE__); warn_flags[1] = 1; }
            M = *M_ptr;
        }
        memcpy(aux_spec, roots, M * sizeof(FNFT_COMPLEX));
    }
    *K_ptr = K;
    *M_ptr = M;
release:
    free(tm); free(p); free(roots); free(qfull); free(qsub);
    return rc;
}

/* ---- src/fnft_nsep.c:82-220 ----------------------------------------------------------------------------------------- */
FNFT_INT fnft_nsep(const FNFT_UINT D, FNFT_COMPLEX const *const q, FNFT_REAL const *const T, FNFT_REAL const phase_shift,
                   FNFT_UINT *const K_ptr, FNFT_COMPLEX *const main_spec, FNFT_UINT *const M_ptr,
                   FNFT_COMPLEX *const aux_spec, FNFT_REAL *const sheet_indices, const FNFT_INT kappa,
                   fnft_nsep_opts_t *opts_ptr)
{
    FNFT_INT rc = FNFT_SUCCESS, warn_flags[2] = {0, 0};
    FNFT_UINT K1, K2, M1, M2;
    /* same checks, same order as :100-119 (D a power of two: the subsampled signal has to stay periodic) */
    if (D < 2 || (D & (D - 1)) != 0) return E_INVALID_ARGUMENT(D);
    if (q == NULL) return E_INVALID_ARGUMENT(q);
    if (T == NULL || T[0] >= T[1]) return E_INVALID_ARGUMENT(T);
    if (abs(kappa) != 1) return E_INVALID_ARGUMENT(kappa);
    if (K_ptr == NULL) return E_INVALID_ARGUMENT(K_ptr);
    if (M_ptr == NULL) return E_INVALID_ARGUMENT(M_ptr);
    if (sheet_indices != NULL) return E_NOT_YET_IMPLEMENTED(sheet_indices, "Pass sheet_indices=\"NULL\".");
    if (opts_ptr == NULL) opts_ptr = &default_opts;
    if (opts_ptr->filtering != fnft_nsep_filt_NONE && main_spec == NULL && aux_spec != NULL)
        return E_INVALID_ARGUMENT(main_spec. Filtering of the auxiliary spectrum is not possible if the main spectrum is not computed.);
    if (disc_degree(opts_ptr->discretization) == 0)
        return E_NOT_YET_IMPLEMENTED(discretization, "GPU path covers the fast (polynomial) discretizations.");

    /* quasi-periodic signals: remove the phase rotation along q, :121-134 */
    const FNFT_REAL Lam_shift = phase_shift / (-2.0 * (T[1] - T[0]));
    const FNFT_REAL eps_t = (T[1] - T[0]) / (FNFT_REAL)D;
    FNFT_COMPLEX *qp = malloc(D * sizeof(FNFT_COMPLEX));
    if (qp == NULL) return E_NOMEM;
    for (FNFT_UINT i = 0; i < D; i++) qp[i] = q[i] * cexp(2.0 * I * Lam_shift * (T[0] + eps_t * (FNFT_REAL)i));
    if (opts_ptr->filtering == fnft_nsep_filt_MANUAL) {   /* the box moves with the spectrum, :138-141 */
        opts_ptr->bounding_box[0] -= Lam_shift;
        opts_ptr->bounding_box[1] -= Lam_shift;
    }
    switch (opts_ptr->localization) {
    case fnft_nsep_loc_MIXED:
        /* non-real points by subsample-and-refine, real points by the grid search, :144-186 */
        K1 = *K_ptr;
        M1 = *M_ptr;
        if (kappa == +1) {
            rc = subsample_and_refine(D, qp, T, &K1, main_spec, &M1, aux_spec, kappa, opts_ptr, 1, warn_flags);
        } else {   /* no non-real main spectrum in the defocusing case */
            K1 = 0;
            rc = subsample_and_refine(D, qp, T, &K1, NULL, &M1, aux_spec, kappa, opts_ptr, 1, warn_flags);
        }
        CHECK(rc, leave);
        if (K1 > *K_ptr || M1 > *M_ptr) { rc = E_ASSERTION_FAILED; goto leave; }
        K2 = *K_ptr - K1;
        M2 = *M_ptr - M1;
        rc = gridsearch(D, qp, T, &K2, main_spec ? main_spec + K1 : NULL, &M2, aux_spec ? aux_spec + M1 : NULL, kappa,
                        opts_ptr, warn_flags);
        CHECK(rc, leave);
        *K_ptr = K1 + K2;
        *M_ptr = M1 + M2;
        break;
    case fnft_nsep_loc_SUBSAMPLE_AND_REFINE:
        rc = subsample_and_refine(D, qp, T, K_ptr, main_spec, M_ptr, aux_spec, kappa, opts_ptr, 0, warn_flags);
        CHECK(rc, leave);
        break;
    case fnft_nsep_loc_GRIDSEARCH:
        rc = gridsearch(D, qp, T, K_ptr, main_spec, M_ptr, aux_spec, kappa, opts_ptr, warn_flags);
        CHECK(rc, leave);
        break;
    default:
        rc = E_INVALID_ARGUMENT(opts_ptr->discretization);
        goto leave;
    }
    if (main_spec != NULL)
        for (FNFT_UINT i = 0; i < *K_ptr; i++) main_spec[i] += Lam_shift;
    if (aux_spec != NULL)
        for (FNFT_UINT i = 0; i < *M_ptr; i++) aux_spec[i] += Lam_shift;
leave:
    if (opts_ptr->filtering == fnft_nsep_filt_MANUAL) {   /* :210-213 */
        opts_ptr->bounding_box[0] += Lam_shift;
        opts_ptr->bounding_box[1] += Lam_shift;
    }
    free(qp);
    return rc;
}
