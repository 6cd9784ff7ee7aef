/*
 * fnft_oracle.c -- CPU restatement (plain C99) of FNFT's fnft_nsev continuous-spectrum path.
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP path, never the thing shipped or measured
 * (except as bench.py's "cpu_baseline" of kind "port").  Parity status: PINNED by the
 * reference's own known-answer vectors (tests/golden/reference_fixtures.json); see fnft_oracle.h.
 *
 * file:line citations are relative to /root/reference.
 */
#define _POSIX_C_SOURCE 200809L
#include "fnft_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static double g_timings[2];
void orc_last_timings(double out[2]) { out[0] = g_timings[0]; out[1] = g_timings[1]; }

/* ------------------------------------------------------------------------------------------ */
/* FFT: Stockham autosort, mixed radix 4/2/3/5.                                                */
/* Length policy follows src/3rd_party/kiss_fft/kiss_fft.c:396-408 (smallest 2^a 3^b 5^c >= n); */
/* transform semantics follow include/private/fnft__fft_wrapper.h:124-137 (out-of-place,      */
/* sign -1 forward / +1 inverse, no 1/len).  The butterfly schedule is this file's own.        */
/* ------------------------------------------------------------------------------------------ */

size_t orc_next_fast_size(size_t n)
{
    if (n == 0) return 1;
    for (;; n++) {
        size_t m = n;
        while (m % 2 == 0) m /= 2;
        while (m % 3 == 0) m /= 3;
        while (m % 5 == 0) m /= 5;
        if (m <= 1) return n;
    }
}

typedef struct {
    size_t len;
    int sign;
    int nfac;
    int fac[64];
    orc_cplx *tw;   /* tw[k] = exp(sign*2*pi*i*k/len), k < len */
    orc_cplx *work; /* len */
} orc_plan;

static int plan_init(orc_plan *pl, size_t len, int sign)
{
    memset(pl, 0, sizeof(*pl));
    pl->len = len;
    pl->sign = sign;
    size_t m = len;
    while (m % 4 == 0) { pl->fac[pl->nfac++] = 4; m /= 4; }
    while (m % 2 == 0) { pl->fac[pl->nfac++] = 2; m /= 2; }
    while (m % 3 == 0) { pl->fac[pl->nfac++] = 3; m /= 3; }
    while (m % 5 == 0) { pl->fac[pl->nfac++] = 5; m /= 5; }
    if (m != 1) return ORC_EC_INVALID_ARGUMENT;
    pl->tw = malloc(len * sizeof(orc_cplx));
    pl->work = malloc(len * sizeof(orc_cplx));
    if (!pl->tw || !pl->work) return ORC_EC_NOMEM;
    /* Table in extended precision, rounded once.  (KissFFT forms cos/sin of a double-rounded
     * angle, kiss_fft.c:357-363; the rounding of the angle is a bias common to every transform of
     * the tree and accumulates linearly in the number of products -- the checker avoids it.) */
    const long double tau = 6.283185307179586476925286766559005768L;
    /* first half by cosl/sinl, second half by the symmetry tw[len-k] = conj(tw[k]) */
    for (size_t k = 0; k <= len / 2; k++) {
        const long double ang = (long double)sign * tau * (long double)k / (long double)len;
        pl->tw[k] = (double)cosl(ang) + I * (double)sinl(ang);
    }
    for (size_t k = len / 2 + 1; k < len; k++) pl->tw[k] = conj(pl->tw[len - k]);
    return ORC_SUCCESS;
}

/* plan of the opposite direction of an existing one: its table is the complex conjugate */
static int plan_init_conj(orc_plan *pl, const orc_plan *other)
{
    memcpy(pl, other, sizeof(*pl));
    pl->sign = -other->sign;
    pl->tw = malloc(other->len * sizeof(orc_cplx));
    pl->work = malloc(other->len * sizeof(orc_cplx));
    if (!pl->tw || !pl->work) return ORC_EC_NOMEM;
    for (size_t k = 0; k < other->len; k++) pl->tw[k] = conj(other->tw[k]);
    return ORC_SUCCESS;
}

static void plan_free(orc_plan *pl)
{
    free(pl->tw);
    free(pl->work);
    pl->tw = pl->work = NULL;
}

/* One Stockham decimation-in-frequency pass of radix r on n_cur-point sub-problems:
 *   y[q + s*(r*p + k)] = w_{n_cur}^{p k} * sum_j x[q + s*(p + m*j)] w_r^{j k},  m = n_cur/r. */
static void stockham_pass(const orc_plan *pl, size_t n_cur, size_t s, int r,
                          const orc_cplx *x, orc_cplx *y)
{
    const size_t len = pl->len;
    const size_t m = n_cur / (size_t)r;
    const size_t tstep = len / n_cur; /* w_{n_cur}^a = tw[a*tstep]; a = k*p < r*m = n_cur, so a*tstep < len */
    const size_t rstep = len / (size_t)r;
    const size_t sm = s * m;
    const double sg = (double)pl->sign;
    const orc_cplx *tw = pl->tw;
    if (r == 2) {
        for (size_t p = 0; p < m; p++) {
            const orc_cplx w1 = tw[p * tstep];
            const orc_cplx *xin = x + s * p;
            orc_cplx *yout = y + s * 2 * p;
            for (size_t q = 0; q < s; q++) {
                const orc_cplx a = xin[q], b = xin[q + sm];
                yout[q] = a + b;
                yout[q + s] = (a - b) * w1;
            }
        }
    } else if (r == 4) {
        for (size_t p = 0; p < m; p++) {
            const orc_cplx w1 = tw[p * tstep], w2 = tw[2 * p * tstep], w3 = tw[3 * p * tstep];
            const orc_cplx *xin = x + s * p;
            orc_cplx *yout = y + s * 4 * p;
            for (size_t q = 0; q < s; q++) {
                const orc_cplx a = xin[q], b = xin[q + sm], c = xin[q + 2 * sm], d = xin[q + 3 * sm];
                const orc_cplx apc = a + c, amc = a - c, bpd = b + d, bmd = b - d;
                /* w_4^1 * (b - d) = sign*i*(b - d) */
                const orc_cplx jbmd = -sg * cimag(bmd) + I * (sg * creal(bmd));
                yout[q] = apc + bpd;
                yout[q + s] = (amc + jbmd) * w1;
                yout[q + 2 * s] = (apc - bpd) * w2;
                yout[q + 3 * s] = (amc - jbmd) * w3;
            }
        }
    } else {
        orc_cplx wr[25], wp[5];
        for (int k = 0; k < r; k++)
            for (int j = 0; j < r; j++) wr[k * r + j] = tw[((size_t)(j * k) % (size_t)r) * rstep];
        for (size_t p = 0; p < m; p++) {
            for (int k = 0; k < r; k++) wp[k] = tw[(size_t)k * p * tstep];
            const orc_cplx *xin = x + s * p;
            orc_cplx *yout = y + s * (size_t)r * p;
            for (size_t q = 0; q < s; q++) {
                orc_cplx v[5];
                for (int j = 0; j < r; j++) v[j] = xin[q + (size_t)j * sm];
                for (int k = 0; k < r; k++) {
                    orc_cplx acc = v[0];
                    for (int j = 1; j < r; j++) acc += v[j] * wr[k * r + j];
                    yout[q + (size_t)k * s] = acc * wp[k];
                }
            }
        }
    }
}

static void plan_exec(orc_plan *pl, const orc_cplx *in, orc_cplx *out)
{
    const size_t len = pl->len;
    if (pl->nfac == 0) { out[0] = in[0]; return; }
    /* ping-pong between out and work so that the last pass lands in out */
    orc_cplx *bufs[2];
    bufs[0] = (pl->nfac % 2 == 1) ? out : pl->work;
    bufs[1] = (pl->nfac % 2 == 1) ? pl->work : out;
    const orc_cplx *src = in;
    size_t n_cur = len, s = 1;
    for (int f = 0; f < pl->nfac; f++) {
        orc_cplx *dst = bufs[f % 2];
        stockham_pass(pl, n_cur, s, pl->fac[f], src, dst);
        n_cur /= (size_t)pl->fac[f];
        s *= (size_t)pl->fac[f];
        src = dst;
    }
}

int orc_fft(size_t len, const orc_cplx *in, orc_cplx *out, int sign)
{
    orc_plan pl;
    int rc = plan_init(&pl, len, sign);
    if (rc == ORC_SUCCESS) plan_exec(&pl, in, out);
    plan_free(&pl);
    return rc;
}

/* DFT of any length: the reference hands any D to KissFFT, which falls back to generic radix-p
 * butterflies (kiss_fft.c:220-279); here lengths with other prime factors go through Bluestein
 * with the chirp exp(sign*i*pi*n^2/len) formed from n^2 mod 2len in extended precision. */
int orc_dft(size_t len, const orc_cplx *in, orc_cplx *out, int sign)
{
    orc_plan pl;
    int rc = plan_init(&pl, len, sign);
    if (rc == ORC_SUCCESS) {
        plan_exec(&pl, in, out);
        plan_free(&pl);
        return rc;
    }
    plan_free(&pl);
    const size_t L = orc_next_fast_size(2 * len - 1);
    orc_cplx *c = malloc(len * sizeof(orc_cplx)), *a = calloc(L, sizeof(orc_cplx)),
             *b = calloc(L, sizeof(orc_cplx)), *fa = malloc(L * sizeof(orc_cplx)),
             *fb = malloc(L * sizeof(orc_cplx));
    if (!c || !a || !b || !fa || !fb) { free(c); free(a); free(b); free(fa); free(fb); return ORC_EC_NOMEM; }
    const long double pi = 3.141592653589793238462643383279502884L;
    for (size_t n = 0; n < len; n++) {
        const unsigned long long r = ((unsigned long long)n * n) % (2ull * len);
        const long double ang = (long double)sign * pi * (long double)r / (long double)len;
        c[n] = (double)cosl(ang) + I * (double)sinl(ang);   /* w^{n^2/2} */
    }
    for (size_t n = 0; n < len; n++) {
        a[n] = in[n] * c[n];
        b[n] = conj(c[n]);
        if (n) b[L - n] = conj(c[n]);
    }
    rc = orc_fft(L, a, fa, -1);
    if (rc == ORC_SUCCESS) rc = orc_fft(L, b, fb, -1);
    if (rc == ORC_SUCCESS) {
        for (size_t k = 0; k < L; k++) fa[k] *= fb[k];
        rc = orc_fft(L, fa, a, +1);
    }
    if (rc == ORC_SUCCESS)
        for (size_t k = 0; k < len; k++) out[k] = c[k] * a[k] / (double)L;
    free(c); free(a); free(b); free(fa); free(fb);
    return rc;
}

/* fnft__misc.c:326-407: band-limited shift of a periodically continued signal by delta */
int orc_misc_resample(size_t D, double eps_t, const orc_cplx *q, double delta, orc_cplx *q_new)
{
    if (!q || D <= 2 || !q_new || eps_t == 0.0) return ORC_EC_INVALID_ARGUMENT;
    orc_cplx *X = malloc(D * sizeof(orc_cplx));
    if (!X) return ORC_EC_NOMEM;
    int rc = orc_dft(D, q, X, -1);
    if (rc == ORC_SUCCESS) {
        const double scl = (double)D * eps_t, pi = acos(-1.0);
        for (size_t i = 0; i < D; i++) {
            const double freq = (i < D / 2) ? (double)i / scl : ((double)i - (double)D) / scl;
            X[i] *= cexp(2 * I * pi * delta * freq);
        }
        rc = orc_dft(D, X, q_new, +1);
        for (size_t i = 0; i < D; i++) q_new[i] /= (double)D;
    }
    free(X);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* fnft__poly_eval.c:25-53 -- Horner; for |z|>1 the reversed polynomial is evaluated in 1/z. */
int orc_poly_eval(size_t deg, const orc_cplx *p, size_t nz, orc_cplx *z)
{
    if (!p || !z) return ORC_EC_INVALID_ARGUMENT;
    for (size_t i = 0; i < nz; i++) {
        orc_cplx acc;
        if (cabs(z[i]) <= 1.0) {
            acc = p[0];
            for (size_t k = 1; k <= deg; k++) acc = p[k] + z[i] * acc;
        } else {
            orc_cplx zi = 1.0 / z[i];
            acc = p[deg];
            for (size_t k = deg; k-- > 0;) acc = p[k] + zi * acc;
            acc *= cpow(z[i], (double)deg);
        }
        z[i] = acc;
    }
    return ORC_SUCCESS;
}

/* fnft__misc.c:41-51 */
double orc_rel_err(size_t len, const orc_cplx *numer, const orc_cplx *exact)
{
    double n = 0.0, d = 0.0;
    for (size_t i = 0; i < len; i++) {
        n += cabs(numer[i] - exact[i]);
        d += cabs(exact[i]);
    }
    return n / d;
}

/* fnft__misc.c:316-324 */
size_t orc_nextpowerof2(size_t n)
{
    if (n == 0) return 0;
    size_t r = 1;
    while (r < n) r *= 2;
    return r;
}

/* fnft__misc.c:306-314 */
static orc_cplx orc_csinc(orc_cplx x)
{
    if (cabs(x) >= 1.0e-8) return csin(x) / x;
    return ccos(x / csqrt(3));
}

/* ------------------------------------------------------------------------------------------ */
/* product tree                                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* fnft__poly_fmult.c:40-43 */
size_t orc_poly_fmult2x2_numel(size_t deg, size_t n) { return 4 * (deg + 1) * orc_nextpowerof2(n); }

/* One 2x2 pair product (fnft__poly_fmult.c:239-328 with 50-121 inlined): every one of the 8
 * input polynomials is transformed once, the two partial products of each entry are summed in
 * the frequency domain (the reference's "mode 2/3"), 4 inverse transforms, scaling by 1/len.
 * The reference's "mode 0/1" (8 inverse transforms, sum in the coefficient domain) differs from
 * this only by rounding. */
static void pair_product(size_t deg, const orc_cplx *a11, size_t a_stride, const orc_cplx *b11,
                         size_t b_stride, orc_cplx *c11, size_t c_stride, orc_plan *fwd,
                         orc_plan *inv, orc_cplx *spec /* 8*len */, orc_cplx *tmp /* 2*len */)
{
    const size_t len = fwd->len;
    const orc_cplx *src[8] = {a11, a11 + a_stride, a11 + 2 * a_stride, a11 + 3 * a_stride,
                              b11, b11 + b_stride, b11 + 2 * b_stride, b11 + 3 * b_stride};
    for (int e = 0; e < 8; e++) {
        memcpy(tmp, src[e], (deg + 1) * sizeof(orc_cplx));
        memset(tmp + deg + 1, 0, (len - deg - 1) * sizeof(orc_cplx));
        plan_exec(fwd, tmp, spec + (size_t)e * len);
    }
    const orc_cplx *A11 = spec, *A12 = spec + len, *A21 = spec + 2 * len, *A22 = spec + 3 * len;
    const orc_cplx *B11 = spec + 4 * len, *B12 = spec + 5 * len, *B21 = spec + 6 * len,
                   *B22 = spec + 7 * len;
    const double dlen = (double)len; /* divide (fnft__poly_fmult.c:110-111): 1/len is not exact */
    for (int e = 0; e < 4; e++) {
        const orc_cplx *L1 = (e < 2) ? A11 : A21, *L2 = (e < 2) ? A12 : A22;
        const orc_cplx *R1 = (e % 2 == 0) ? B11 : B12, *R2 = (e % 2 == 0) ? B21 : B22;
        for (size_t k = 0; k < len; k++) tmp[k] = L1[k] * R1[k] + L2[k] * R2[k];
        plan_exec(inv, tmp, tmp + len);
        orc_cplx *dst = c11 + (size_t)e * c_stride;
        for (size_t k = 0; k < 2 * deg + 1; k++) dst[k] = tmp[len + k] / dlen;
    }
}

/* fnft__poly_fmult.c:330-374 */
static int32_t rescale2x2(size_t d, orc_cplx *c11, orc_cplx *c12, orc_cplx *c21, orc_cplx *c22)
{
    double mx = 0.0;
    for (size_t i = 0; i <= d; i++) {
        double v;
        v = cabs(c11[i]); if (v > mx) mx = v;
        v = cabs(c12[i]); if (v > mx) mx = v;
        v = cabs(c21[i]); if (v > mx) mx = v;
        v = cabs(c22[i]); if (v > mx) mx = v;
    }
    if (mx == 0.0) return 0;
    const int32_t a = (int32_t)floor(log2(mx));
    const double scl = pow(2.0, -a);
    for (size_t i = 0; i <= d; i++) {
        c11[i] *= scl; c12[i] *= scl; c21[i] *= scl; c22[i] *= scl;
    }
    return a;
}

/* fnft__poly_fmult.c:381-546 */
int orc_poly_fmult2x2(size_t *d, size_t n, orc_cplx *p, orc_cplx *result, int32_t *W_ptr)
{
    if (!d || !p || !result || n == 0) return ORC_EC_INVALID_ARGUMENT;
    const size_t deg0 = *d;
    size_t deg = deg0;
    const size_t n0 = n;
    const size_t n_excess = orc_nextpowerof2(n) - n;
    int32_t W = 0;

    /* :404-445 -- move entries to the padded stride, append n_excess copies of z^deg * I */
    if (n_excess > 0) {
        const size_t np = n + n_excess;
        for (int e = 3; e >= 1; e--)
            memmove(p + (size_t)e * np * (deg + 1), p + (size_t)e * n * (deg + 1),
                    n * (deg + 1) * sizeof(orc_cplx));
        for (int e = 0; e < 4; e++) {
            orc_cplx *pad = p + (size_t)e * np * (deg + 1) + n * (deg + 1);
            for (size_t i = 0; i < n_excess * (deg + 1); i++) pad[i] = 0.0;
            if (e == 0 || e == 3)
                for (size_t i = 0; i < n_excess; i++) pad[i * (deg + 1)] = 1.0;
        }
        n = np;
    }
    const size_t p_stride = n * (deg + 1);

    const size_t maxlen = orc_next_fast_size(2 * (deg * n / 2 + 1) - 1);
    orc_cplx *spec = malloc(8 * maxlen * sizeof(orc_cplx));
    orc_cplx *tmp = malloc(2 * maxlen * sizeof(orc_cplx));
    if (!spec || !tmp) { free(spec); free(tmp); return ORC_EC_NOMEM; }

    size_t r_stride = 0;
    int rc = ORC_SUCCESS;
    if (n < 2) { /* single matrix: nothing to multiply */
        for (int e = 0; e < 4; e++)
            memcpy(result + (size_t)e * (deg + 1), p + (size_t)e * p_stride,
                   (deg + 1) * sizeof(orc_cplx));
        r_stride = deg + 1;
    }
    while (n >= 2) { /* :460-519 */
        const size_t len = orc_next_fast_size(2 * (deg + 1) - 1);
        orc_plan fwd, inv;
        memset(&inv, 0, sizeof inv);
        rc = plan_init(&fwd, len, -1);
        if (rc == ORC_SUCCESS) rc = plan_init_conj(&inv, &fwd);
        if (rc != ORC_SUCCESS) { plan_free(&fwd); plan_free(&inv); break; }
        r_stride = (n / 2) * (2 * deg + 1);
        for (size_t i = 0; i < n; i += 2) {
            const size_t o1 = i * (deg + 1), o2 = o1 + (deg + 1), orr = (i / 2) * (2 * deg + 1);
            pair_product(deg, p + o1, p_stride, p + o2, p_stride, result + orr, r_stride, &fwd,
                         &inv, spec, tmp);
            if (W_ptr)
                W += rescale2x2(2 * deg, result + orr, result + r_stride + orr,
                                result + 2 * r_stride + orr, result + 3 * r_stride + orr);
        }
        plan_free(&fwd);
        plan_free(&inv);
        deg *= 2;
        n /= 2;
        if (n > 1)
            for (int e = 0; e < 4; e++)
                memcpy(p + (size_t)e * p_stride, result + (size_t)e * r_stride,
                       n * (deg + 1) * sizeof(orc_cplx));
    }
    free(spec);
    free(tmp);
    if (rc != ORC_SUCCESS) return rc;

    /* :522-533 -- drop the trailing zero coefficients the identity padding produced */
    if (n_excess > 0 && n0 > 0) {
        deg -= n_excess * deg0;
        for (int e = 1; e < 4; e++)
            memmove(result + (size_t)e * (deg + 1), result + (size_t)e * r_stride,
                    (deg + 1) * sizeof(orc_cplx));
    }
    *d = deg;
    if (W_ptr) *W_ptr = W;
    return ORC_SUCCESS;
}

/* ------------------------------------------------------------------------------------------ */
/* chirp z-transform, fnft__poly_chirpz.c:33-105                                               */
/* ------------------------------------------------------------------------------------------ */
int orc_poly_chirpz(size_t deg, const orc_cplx *p, orc_cplx A, orc_cplx W, size_t M,
                    orc_cplx *result)
{
    if (!p || M == 0 || !result) return ORC_EC_INVALID_ARGUMENT;
    const size_t N = deg + 1;
    const size_t L = orc_next_fast_size(N + M - 1);
    orc_cplx *Y = malloc(L * sizeof(orc_cplx)), *V = malloc(L * sizeof(orc_cplx)),
             *buf = malloc(L * sizeof(orc_cplx));
    orc_plan fwd, inv;
    int rc = (Y && V && buf) ? ORC_SUCCESS : ORC_EC_NOMEM;
    memset(&fwd, 0, sizeof fwd);
    memset(&inv, 0, sizeof inv);
    if (rc == ORC_SUCCESS) rc = plan_init(&fwd, L, -1);
    if (rc == ORC_SUCCESS) rc = plan_init_conj(&inv, &fwd);
    if (rc == ORC_SUCCESS) {
        for (size_t n = 0; n < N; n++) { /* :68-71 */
            const double dn = (double)n;
            buf[n] = p[deg - n] * cpow(A, -1.0 * dn) * cpow(W, 0.5 * dn * dn);
        }
        for (size_t n = N; n < L; n++) buf[n] = 0;
        plan_exec(&fwd, buf, Y);
        for (size_t n = 0; n < M; n++) { /* :76-82 */
            const double dn = (double)n;
            buf[n] = cpow(W, -0.5 * dn * dn);
        }
        for (size_t n = M; n <= L - N; n++) buf[n] = 0;
        for (size_t n = L - N + 1; n < L; n++) {
            const double dn = (double)(L - n);
            buf[n] = cpow(W, -0.5 * dn * dn);
        }
        plan_exec(&fwd, buf, V);
        for (size_t n = 0; n < L; n++) buf[n] = V[n] * Y[n];
        plan_exec(&inv, buf, V);
        for (size_t n = 0; n < M; n++) { /* :94-95 */
            const double dn = (double)n;
            result[n] = cpow(W, 0.5 * dn * dn) * V[n] / (double)L;
        }
    }
    plan_free(&fwd);
    plan_free(&inv);
    free(Y); free(V); free(buf);
    return rc;
}

int orc_poly_chirpz_p(size_t deg, const orc_cplx *p, const double *A, const double *W, size_t M,
                      orc_cplx *result)
{
    return orc_poly_chirpz(deg, p, A[0] + I * A[1], W[0] + I * W[1], M, result);
}

/* ------------------------------------------------------------------------------------------ */
/* per-sample AKNS coefficients                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* fnft__akns_discretization.c:29-67 (schemes this restatement covers) */
size_t orc_akns_degree(int disc)
{
    switch (disc) {
    case ORC_AKNS_2SPLIT2_MODAL: case ORC_AKNS_2SPLIT1A: case ORC_AKNS_2SPLIT1B:
    case ORC_AKNS_2SPLIT2A: case ORC_AKNS_2SPLIT2B: case ORC_AKNS_2SPLIT2S: return 1;
    case ORC_AKNS_2SPLIT3S: case ORC_AKNS_2SPLIT4B: case ORC_AKNS_4SPLIT4B: return 2;
    case ORC_AKNS_2SPLIT3A: case ORC_AKNS_2SPLIT3B: return 3;
    case ORC_AKNS_2SPLIT4A: case ORC_AKNS_4SPLIT4A: return 4;
    case ORC_AKNS_2SPLIT6B: return 6;
    case ORC_AKNS_2SPLIT6A: case ORC_AKNS_2SPLIT8B: return 12;
    case ORC_AKNS_2SPLIT5A: case ORC_AKNS_2SPLIT5B: return 15;
    case ORC_AKNS_2SPLIT8A: return 24;
    case ORC_AKNS_2SPLIT7A: case ORC_AKNS_2SPLIT7B: return 105;
    default: return 0;
    }
}

/* fnft__nse_discretization.c:108-200 */
int orc_nse_to_akns(int d)
{
    switch (d) {
    case ORC_NSE_2SPLIT2_MODAL: return ORC_AKNS_2SPLIT2_MODAL;
    case ORC_NSE_2SPLIT1A: return ORC_AKNS_2SPLIT1A;
    case ORC_NSE_2SPLIT1B: return ORC_AKNS_2SPLIT1B;
    case ORC_NSE_2SPLIT2A: return ORC_AKNS_2SPLIT2A;
    case ORC_NSE_2SPLIT2B: return ORC_AKNS_2SPLIT2B;
    case ORC_NSE_2SPLIT2S: return ORC_AKNS_2SPLIT2S;
    case ORC_NSE_2SPLIT3A: return ORC_AKNS_2SPLIT3A;
    case ORC_NSE_2SPLIT3B: return ORC_AKNS_2SPLIT3B;
    case ORC_NSE_2SPLIT3S: return ORC_AKNS_2SPLIT3S;
    case ORC_NSE_2SPLIT4A: return ORC_AKNS_2SPLIT4A;
    case ORC_NSE_2SPLIT4B: return ORC_AKNS_2SPLIT4B;
    case ORC_NSE_2SPLIT5A: return ORC_AKNS_2SPLIT5A;
    case ORC_NSE_2SPLIT5B: return ORC_AKNS_2SPLIT5B;
    case ORC_NSE_2SPLIT6A: return ORC_AKNS_2SPLIT6A;
    case ORC_NSE_2SPLIT6B: return ORC_AKNS_2SPLIT6B;
    case ORC_NSE_2SPLIT7A: return ORC_AKNS_2SPLIT7A;
    case ORC_NSE_2SPLIT7B: return ORC_AKNS_2SPLIT7B;
    case ORC_NSE_2SPLIT8A: return ORC_AKNS_2SPLIT8A;
    case ORC_NSE_2SPLIT8B: return ORC_AKNS_2SPLIT8B;
    case ORC_NSE_4SPLIT4A: return ORC_AKNS_4SPLIT4A;
    case ORC_NSE_4SPLIT4B: return ORC_AKNS_4SPLIT4B;
    default: return -1;
    }
}

/* expm([[0,q],[r,0]]*h) = [[c, q*s],[r*s, c]], fnft__akns_fscatter.c:46-59 */
typedef struct { orc_cplx c, qs, rs; } step_exp;
static step_exp zero_freq_step(double h, orc_cplx q, orc_cplx r)
{
    step_exp e;
    orc_cplx Delta = h * csqrt(-q * r);
    orc_cplx del = h * orc_csinc(Delta);
    e.c = ccos(Delta);
    e.qs = q * del;
    e.rs = r * del;
    return e;
}

/* ---- splitting schemes of order 5..8 (fnft__akns_fscatter.c:435-912) ------------------------
 * The reference spells these out coefficient by coefficient.  They are the Richardson-type
 * combinations  sum_n w_n Psi_n(h)  of the paper behind the library (Prins & Wahls, "Higher order
 * exponential splittings for the fast non-linear Fourier transform of the KdV equation", 2018):
 *   odd order p, n = 1,3,..,p:  Psi_n = n alternating Lie-Trotter sub-steps,
 *        "A": e^{A/n} e^{2B/n} e^{2A/n} ... e^{2A/n} e^{B/n},   "B": roles of A and B swapped;
 *   even order p, n = 1,..,p/2: Psi_n = n Strang sub-steps,
 *        "A": e^{A/2n} e^{B/n} e^{A/n} ... e^{B/n} e^{A/2n},    "B": swapped;
 *   w_n = n^(p-1) / prod_{m != n} (n^2 - m^2)   (625/384, -81/128, 1/192 for p = 5, :456-459).
 * With z = e^{2 i lambda h/deg}, e^{aA} is diag(1, z^{a*deg}) up to a scalar (compare 2SPLIT1A/1B,
 * :150-203) and e^{bB} = expm([[0,q],[r,0]] b h) (:46-59).  Here the products are multiplied out as
 * dense 2x2 polynomial matrices; the golden vectors of test/fnft__akns_fscatter/ pin the result. */
typedef struct { int is_B; int num, den; } split_factor;  /* fraction num/den of the step */

static size_t split_sequence(int order_odd, int b_first, int n, split_factor *f)
{
    size_t k = 0;
    if (order_odd) { /* X(1/n) Y(2/n) X(2/n) ... X(2/n) Y(1/n), (n+1)/2 of each */
        const int cnt = (n + 1) / 2;
        for (int i = 0; i < cnt; i++) {
            f[k++] = (split_factor){b_first, (i == 0) ? 1 : 2, n};
            f[k++] = (split_factor){!b_first, (i == cnt - 1) ? 1 : 2, n};
        }
    } else {         /* X(1/2n) [Y(1/n) X(1/n)]^(n-1) Y(1/n) X(1/2n) */
        f[k++] = (split_factor){b_first, 1, 2 * n};
        for (int i = 0; i < n; i++) {
            f[k++] = (split_factor){!b_first, 1, n};
            f[k++] = (split_factor){b_first, 1, (i == n - 1) ? 2 * n : n};
        }
    }
    return k;
}

static int split_scheme(int disc, int *order, int *b_first)
{
    switch (disc) {
    case ORC_AKNS_2SPLIT5A: *order = 5; *b_first = 0; return 1;
    case ORC_AKNS_2SPLIT5B: *order = 5; *b_first = 1; return 1;
    case ORC_AKNS_2SPLIT6A: *order = 6; *b_first = 0; return 1;
    case ORC_AKNS_2SPLIT6B: *order = 6; *b_first = 1; return 1;
    case ORC_AKNS_2SPLIT7A: *order = 7; *b_first = 0; return 1;
    case ORC_AKNS_2SPLIT7B: *order = 7; *b_first = 1; return 1;
    case ORC_AKNS_2SPLIT8A: *order = 8; *b_first = 0; return 1;
    case ORC_AKNS_2SPLIT8B: *order = 8; *b_first = 1; return 1;
    default: return 0;
    }
}

/* coefficients of one sample, out[e*(deg+1) + k], highest power first */
static void split_sample(int order, int b_first, size_t deg, double eps_t, orc_cplx q, orc_cplx r,
                         orc_cplx *out, orc_cplx *work)
{
    const size_t w = deg + 1;
    const int odd = order & 1;
    const int nterms = odd ? (order + 1) / 2 : order / 2;
    orc_cplx *P[4] = {work, work + w, work + 2 * w, work + 3 * w}; /* ascending powers */
    for (size_t i = 0; i < 4 * w; i++) out[i] = 0.0;
    for (int t = 0; t < nterms; t++) {
        const int n = odd ? 2 * t + 1 : t + 1;
        long double wn = powl((long double)n, (long double)(order - 1 - (odd ? 0 : 1)));
        /* odd p: n^(p-1); even p: n^(p-2) -- both are n^(2*(nterms-1)) */
        wn = powl((long double)n, 2.0L * (nterms - 1));
        for (int u = 0; u < nterms; u++) {
            const int m = odd ? 2 * u + 1 : u + 1;
            if (m != n) wn /= (long double)(n * n - m * m);
        }
        split_factor f[64];
        const size_t nf = split_sequence(odd, b_first, n, f);
        for (size_t i = 0; i < 4 * w; i++) work[i] = 0.0;
        P[0][0] = 1.0;
        P[3][0] = 1.0;
        for (size_t i = 0; i < nf; i++) {
            if (!f[i].is_B) { /* right-multiply by diag(1, z^k): shift the second column */
                const size_t k = (size_t)f[i].num * deg / (size_t)f[i].den;
                for (int row = 0; row < 2; row++) {
                    orc_cplx *c = P[2 * row + 1];
                    for (size_t j = w; j-- > 0;) c[j] = (j >= k) ? c[j - k] : 0.0;
                }
            } else {
                const step_exp e = zero_freq_step(eps_t * (double)f[i].num / (double)f[i].den, q, r);
                for (int row = 0; row < 2; row++) {
                    orc_cplx *c0 = P[2 * row], *c1 = P[2 * row + 1];
                    for (size_t j = 0; j < w; j++) {
                        const orc_cplx a = c0[j], b = c1[j];
                        c0[j] = a * e.c + b * e.rs;
                        c1[j] = a * e.qs + b * e.c;
                    }
                }
            }
        }
        for (int e = 0; e < 4; e++)
            for (size_t j = 0; j < w; j++) out[(size_t)e * w + (deg - j)] += (double)wn * P[e][j];
    }
}

size_t orc_akns_fscatter_numel(size_t D, int disc)
{
    size_t deg = orc_akns_degree(disc);
    return deg == 0 ? 0 : orc_poly_fmult2x2_numel(deg, D);
}

/* fnft__akns_fscatter.c:116-917: block j of each entry belongs to sample D-1-j,
 * coefficients highest power of z first. */
int orc_akns_coeffs(size_t D, const orc_cplx *q, const orc_cplx *r, double eps_t, orc_cplx *p,
                    int disc)
{
    const size_t deg = orc_akns_degree(disc);
    if (deg == 0) return ORC_EC_INVALID_ARGUMENT;
    const size_t w = deg + 1;
    orc_cplx *p11 = p, *p12 = p + D * w, *p21 = p + 2 * D * w, *p22 = p + 3 * D * w;
    const double h = eps_t / (double)deg;
    for (size_t j = 0; j < D; j++, p11 += w, p12 += w, p21 += w, p22 += w) {
        const orc_cplx qi = q[D - 1 - j], ri = r[D - 1 - j];
        for (size_t k = 0; k < w; k++) p11[k] = p12[k] = p21[k] = p22[k] = 0.0;
        switch (disc) {
        case ORC_AKNS_2SPLIT2_MODAL: { /* :118-148 */
            if (creal(qi) == creal(ri) && eps_t * cabs(qi) >= 1.0) return ORC_EC_OTHER;
            const orc_cplx s = 1.0 / csqrt(1 - eps_t * qi * eps_t * ri);
            p11[1] = s; p12[0] = s * eps_t * qi; p21[1] = s * eps_t * ri; p22[0] = s;
            break;
        }
        case ORC_AKNS_2SPLIT1A: { /* :150-176 */
            step_exp e = zero_freq_step(h, qi, ri);
            p11[1] = e.c; p12[1] = e.qs; p21[0] = e.rs; p22[0] = e.c;
            break;
        }
        case ORC_AKNS_2SPLIT1B: case ORC_AKNS_2SPLIT2A: { /* :178-203 */
            step_exp e = zero_freq_step(h, qi, ri);
            p11[1] = e.c; p12[0] = e.qs; p21[1] = e.rs; p22[0] = e.c;
            break;
        }
        case ORC_AKNS_2SPLIT2B: { /* :204-228 */
            step_exp e = zero_freq_step(0.5 * h, qi, ri);
            p11[0] = e.qs * e.rs; p11[1] = e.c * e.c;
            p12[0] = p12[1] = e.c * e.qs;
            p21[0] = p21[1] = e.c * e.rs;
            p22[0] = p11[1]; p22[1] = p11[0];
            break;
        }
        case ORC_AKNS_2SPLIT2S: { /* :230-254 */
            step_exp e = zero_freq_step(h, qi, ri);
            p11[1] = e.c; p12[0] = p12[1] = e.qs / 2; p21[0] = p21[1] = e.rs / 2; p22[0] = e.c;
            break;
        }
        case ORC_AKNS_2SPLIT3A: { /* :256-292 */
            step_exp e1 = zero_freq_step(h, qi, ri), e2 = zero_freq_step(2 * h, qi, ri),
                     e3 = zero_freq_step(3 * h, qi, ri);
            p11[1] = 9 * e1.rs * e2.qs / 8; p11[3] = (9 * e1.c * e2.c - e3.c) / 8;
            p12[1] = 9 * e1.c * e2.qs / 8;  p12[3] = (9 * e1.qs * e2.c - e3.qs) / 8;
            p21[0] = (9 * e1.rs * e2.c - e3.rs) / 8; p21[2] = 9 * e1.c * e2.rs / 8;
            p22[0] = p11[3]; p22[2] = 9 * e1.qs * e2.rs / 8;
            break;
        }
        case ORC_AKNS_2SPLIT3B: { /* :294-330 */
            step_exp e1 = zero_freq_step(h, qi, ri), e2 = zero_freq_step(2 * h, qi, ri),
                     e3 = zero_freq_step(3 * h, qi, ri);
            p11[1] = 9 * e1.qs * e2.rs / 8; p11[3] = (9 * e1.c * e2.c - e3.c) / 8;
            p12[0] = (9 * e1.qs * e2.c - e3.qs) / 8; p12[2] = 9 * e1.c * e2.qs / 8;
            p21[1] = 9 * e1.c * e2.rs / 8; p21[3] = (9 * e1.rs * e2.c - e3.rs) / 8;
            p22[0] = p11[3]; p22[2] = 9 * e1.rs * e2.qs / 8;
            break;
        }
        case ORC_AKNS_2SPLIT3S: { /* :331-361 */
            step_exp e1 = zero_freq_step(h, qi, ri), e2 = zero_freq_step(2 * h, qi, ri);
            p11[0] = 2 * e1.qs * e1.rs / 3; p11[2] = (2 * e1.c * e1.c + e2.c) / 3;
            p12[0] = p12[2] = (4 * e1.c * e1.qs - e2.qs) / 6; p12[1] = 2 * e2.qs / 3;
            p21[0] = p21[2] = (4 * e1.c * e1.rs - e2.rs) / 6; p21[1] = 2 * e2.rs / 3;
            p22[0] = p11[2]; p22[2] = p11[0];
            break;
        }
        case ORC_AKNS_2SPLIT4A: case ORC_AKNS_4SPLIT4A: { /* :362-401 */
            step_exp e2 = zero_freq_step(2 * h, qi, ri), e4 = zero_freq_step(4 * h, qi, ri);
            p11[2] = 4 * e2.qs * e2.rs / 3; p11[4] = (4 * e2.c * e2.c - e4.c) / 3;
            p12[1] = p12[3] = 4 * e2.c * e2.qs / 3; p12[2] = -e4.qs / 3;
            p21[1] = p21[3] = 4 * e2.c * e2.rs / 3; p21[2] = -e4.rs / 3;
            p22[0] = p11[4]; p22[2] = p11[2];
            break;
        }
        case ORC_AKNS_2SPLIT4B: case ORC_AKNS_4SPLIT4B: { /* :402-433 */
            step_exp eh = zero_freq_step(0.5 * h, qi, ri), e1 = zero_freq_step(h, qi, ri);
            p11[0] = (4 * e1.c * eh.qs * eh.rs - e1.qs * e1.rs) / 3;
            p11[1] = 4 * (e1.qs * eh.c * eh.rs + e1.rs * eh.c * eh.qs) / 3;
            p11[2] = (4 * e1.c * eh.c * eh.c - e1.c * e1.c) / 3;
            p12[0] = p12[2] = (4 * e1.c * eh.c * eh.qs - e1.c * e1.qs) / 3;
            p12[1] = 4 * (e1.qs * eh.c * eh.c + e1.rs * eh.qs * eh.qs) / 3;
            p21[0] = p21[2] = (4 * e1.c * eh.c * eh.rs - e1.c * e1.rs) / 3;
            p21[1] = 4 * (e1.rs * eh.c * eh.c + e1.qs * eh.rs * eh.rs) / 3;
            p22[0] = p11[2]; p22[1] = p11[1]; p22[2] = p11[0];
            break;
        }
        default: {
            int order, b_first;
            if (!split_scheme(disc, &order, &b_first)) return ORC_EC_INVALID_ARGUMENT;
            orc_cplx *tmp = malloc(8 * w * sizeof(orc_cplx));
            if (!tmp) return ORC_EC_NOMEM;
            split_sample(order, b_first, deg, eps_t, qi, ri, tmp, tmp + 4 * w);
            for (size_t k = 0; k < w; k++) {
                p11[k] = tmp[k]; p12[k] = tmp[w + k]; p21[k] = tmp[2 * w + k]; p22[k] = tmp[3 * w + k];
            }
            free(tmp);
            break;
        }
        }
    }
    return ORC_SUCCESS;
}

/* fnft__akns_fscatter.c:64-925 */
int orc_akns_fscatter(size_t D, const orc_cplx *q, const orc_cplx *r, double eps_t,
                      orc_cplx *result, size_t *deg_ptr, int32_t *W_ptr, int disc)
{
    if (D == 0 || !q || !r || !(eps_t > 0.0) || !result || !deg_ptr)
        return ORC_EC_INVALID_ARGUMENT;
    const size_t numel = orc_akns_fscatter_numel(D, disc);
    if (numel == 0) return ORC_EC_INVALID_ARGUMENT;
    orc_cplx *p = malloc(numel * sizeof(orc_cplx));
    if (!p) return ORC_EC_NOMEM;
    *deg_ptr = orc_akns_degree(disc);
    int rc = orc_akns_coeffs(D, q, r, eps_t, p, disc);
    if (rc == ORC_SUCCESS) rc = orc_poly_fmult2x2(deg_ptr, D, p, result, W_ptr);
    free(p);
    return rc;
}

/* fnft__nse_fscatter.c:34-42 */
size_t orc_nse_fscatter_numel(size_t D, int nse_disc)
{
    int a = orc_nse_to_akns(nse_disc);
    return a < 0 ? 0 : orc_akns_fscatter_numel(D, a);
}

/* fnft__nse_fscatter.c:44-91 */
int orc_nse_fscatter(size_t D, const orc_cplx *q, double eps_t, int kappa, orc_cplx *result,
                     size_t *deg_ptr, int32_t *W_ptr, int nse_disc)
{
    if (D == 0 || !q || !(eps_t > 0.0) || abs(kappa) != 1 || !result || !deg_ptr)
        return ORC_EC_INVALID_ARGUMENT;
    const int a = orc_nse_to_akns(nse_disc);
    if (a < 0) return ORC_EC_INVALID_ARGUMENT;
    orc_cplx *r = malloc(D * sizeof(orc_cplx));
    if (!r) return ORC_EC_NOMEM;
    for (size_t i = 0; i < D; i++) r[i] = (kappa == 1) ? -conj(q[i]) : conj(q[i]);
    int rc = orc_akns_fscatter(D, q, r, eps_t, result, deg_ptr, W_ptr, a);
    free(r);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* continuous spectrum                                                                          */
/* ------------------------------------------------------------------------------------------ */

/* fnft__nse_discretization.c:240-379; boundary coefficient 0.5 (fnft__akns_discretization.c:72-109) */
static void phase_factors(int nse_disc, double eps_t, size_t D, const double *T, double *rho,
                          double *a, double *b)
{
    const double bc = 0.5;
    const int shifted = (nse_disc == ORC_NSE_2SPLIT2A || nse_disc == ORC_NSE_2SPLIT2_MODAL);
    const double deg = (double)orc_akns_degree(orc_nse_to_akns(nse_disc));
    *rho = -2.0 * (T[1] + eps_t * bc) + (shifted ? eps_t / deg : 0.0);
    *a = -eps_t * (double)D + (T[1] + eps_t * bc) - (T[0] - eps_t * bc);
    *b = -eps_t * (double)D - (T[1] + eps_t * bc) - (T[0] - eps_t * bc) + (shifted ? eps_t / deg : 0.0);
}

/* fnft_nsev.c:744-891, fast (polynomial) branch */
size_t orc_nse_upsampling(int nse_disc)
{
    return (nse_disc == ORC_NSE_4SPLIT4A || nse_disc == ORC_NSE_4SPLIT4B) ? 2 : 1;
}

int orc_nsev_contspec(size_t deg, int32_t W, const orc_cplx *tm, const double *T, size_t D,
                      const double *XI, size_t M, orc_cplx *result, int nse_disc, int cstype)
{
    const int a_disc = orc_nse_to_akns(nse_disc);
    if (a_disc < 0) return ORC_EC_INVALID_ARGUMENT;
    /* D counts the preprocessed samples; step size and phase factors refer to D_given = D/upsampling
     * (fnft_nsev.c:766-774), lambda -> z uses degree*upsampling (fnft__akns_discretization.c:204-219) */
    const size_t ups = orc_nse_upsampling(nse_disc);
    const size_t D_eff = D;
    D = D_eff / ups;
    const double deg1 = (double)(orc_akns_degree(a_disc) * ups);
    const double eps_t = (T[1] - T[0]) / (double)(D - 1);
    const double eps_xi = (XI[1] - XI[0]) / (double)(M - 1);
    orc_cplx *H = malloc(2 * M * sizeof(orc_cplx));
    if (!H) return ORC_EC_NOMEM;
    orc_cplx *H11 = H, *H21 = H + M;
    /* lambda -> z (fnft__akns_discretization.c:204-219), fnft_nsev.c:822-827 */
    const orc_cplx V = cexp(2 * I * (orc_cplx)eps_xi * eps_t / deg1);
    const orc_cplx A = cexp(2 * I * (orc_cplx)(-XI[0]) * eps_t / deg1);
    int rc = orc_poly_chirpz(deg, tm, A, V, M, H11);
    if (rc == ORC_SUCCESS) rc = orc_poly_chirpz(deg, tm + 2 * (deg + 1), A, V, M, H21);
    if (rc != ORC_SUCCESS) { free(H); return rc; }
    double pf_rho, pf_a, pf_b;
    phase_factors(nse_disc, eps_t, D, T, &pf_rho, &pf_a, &pf_b);
    size_t offset = 0;
    if (cstype != ORC_CS_RHO && cstype != ORC_CS_AB && cstype != ORC_CS_BOTH) {
        free(H);
        return ORC_EC_INVALID_ARGUMENT;
    }
    if (cstype == ORC_CS_BOTH) offset = M;
    if (cstype == ORC_CS_RHO || cstype == ORC_CS_BOTH) { /* :844-855 */
        for (size_t i = 0; i < M; i++) {
            const orc_cplx xi = XI[0] + eps_xi * (double)i;
            if (H11[i] == 0.0) { free(H); return ORC_EC_DIV_BY_ZERO; }
            result[i] = H21[i] * cexp(I * xi * pf_rho) / H11[i];
        }
    }
    if (cstype == ORC_CS_AB || cstype == ORC_CS_BOTH) { /* :861-876 */
        const double scale = pow(2.0, W);
        for (size_t i = 0; i < M; i++) {
            const orc_cplx xi = XI[0] + eps_xi * (double)i;
            result[offset + i] = H11[i] * scale * cexp(I * xi * pf_a);
            result[offset + M + i] = H21[i] * scale * cexp(I * xi * pf_b);
        }
    }
    free(H);
    return ORC_SUCCESS;
}

/* fnft_nsev.c:458-565 (fnft_nsev_base), contspec-only subset */
static int orc_nsev_base(size_t D, const orc_cplx *q, const double *T, size_t M, orc_cplx *contspec,
                         const double *XI, int kappa, int nse_disc, int cstype, int normalization_flag);

/* fnft__nse_discretization.c:386-656 restricted to the splitting schemes: subsampling
 * (:419-427, :466-473) and, for 4SPLIT4A/B, the two band-limited resamplings at -/+ sqrt(3)/6 of
 * the (subsampled) step combined with the weights of fnft__akns_discretization.c:284-298
 * (:474-503).  *q_pre gets Dsub*upsampling entries; first_last as at :650-651. */
int orc_nse_preprocess(size_t D, const orc_cplx *q, double eps_t, size_t *Dsub_ptr, orc_cplx **q_pre,
                       size_t *first_last, int nse_disc)
{
    if (D < 2 || !q || !Dsub_ptr || !q_pre || !(eps_t > 0.0) || !first_last) return ORC_EC_INVALID_ARGUMENT;
    if (orc_nse_to_akns(nse_disc) < 0) return ORC_EC_INVALID_ARGUMENT;
    size_t Dsub = *Dsub_ptr;
    if (Dsub < 2) Dsub = 2;
    if (Dsub > D) Dsub = D;
    const size_t nskip = (size_t)round((double)D / (double)Dsub);
    Dsub = (size_t)round((double)D / (double)nskip);
    const size_t ups = orc_nse_upsampling(nse_disc);
    const size_t D_eff = Dsub * ups;
    orc_cplx *out = malloc(D_eff * sizeof(orc_cplx));
    if (!out) return ORC_EC_NOMEM;
    int rc = ORC_SUCCESS;
    if (ups == 1) {
        for (size_t i = 0; i < D_eff; i++) out[i] = q[i * nskip];
    } else {
        orc_cplx *q1 = malloc(D * sizeof(orc_cplx)), *q2 = malloc(D * sizeof(orc_cplx));
        if (!q1 || !q2) rc = ORC_EC_NOMEM;
        const double scl = sqrt(3.0) / 6.0;
        if (rc == ORC_SUCCESS) rc = orc_misc_resample(D, eps_t, q, -eps_t * scl * (double)nskip, q1);
        if (rc == ORC_SUCCESS) rc = orc_misc_resample(D, eps_t, q, eps_t * scl * (double)nskip, q2);
        if (rc == ORC_SUCCESS) {
            const double w0 = 0.25 + scl, w1 = 0.25 - scl;
            for (size_t i = 0, is = 0; is < D_eff; is += 2, i += nskip) {
                out[is] = w0 * q1[i] + w1 * q2[i];
                out[is + 1] = w1 * q1[i] + w0 * q2[i];
            }
        }
        free(q1);
        free(q2);
    }
    if (rc != ORC_SUCCESS) { free(out); return rc; }
    *q_pre = out;
    *Dsub_ptr = Dsub;
    first_last[0] = 0;
    first_last[1] = (Dsub - 1) * nskip;
    return ORC_SUCCESS;
}

/* fnft_nsev.c:133-453, contspec-only subset: preprocessing (:272), base call (:312), Richardson
 * extrapolation (:316-406): a second transform of every other sample and
 * (s*fine - coarse)/(s - 1), s = (eps_sub/eps)^order, order = 2 for the 2SPLIT schemes and 4 for
 * 4SPLIT4A/B (fnft__akns_discretization.c:157-192), where |xi| < 0.9*pi/(2*eps_sub). */
int orc_fnft_nsev_ex(size_t D, const orc_cplx *q, const double *T, size_t M, orc_cplx *contspec,
                     const double *XI, int kappa, int nse_disc, int cstype, int normalization_flag,
                     int richardson_flag)
{
    if (D < 2 || !q || !T || !(T[0] < T[1])) return ORC_EC_INVALID_ARGUMENT;
    if (contspec && (!XI || !(XI[0] < XI[1]))) return ORC_EC_INVALID_ARGUMENT;
    if (abs(kappa) != 1) return ORC_EC_INVALID_ARGUMENT;
    if (orc_nse_to_akns(nse_disc) < 0) return ORC_EC_INVALID_ARGUMENT;
    const size_t ups = orc_nse_upsampling(nse_disc);
    const double eps_t = (T[1] - T[0]) / (double)(D - 1);
    size_t Dsub = D, fl[2];
    orc_cplx *q_pre = NULL;
    int rc = orc_nse_preprocess(D, q, eps_t, &Dsub, &q_pre, fl, nse_disc);
    if (rc != ORC_SUCCESS) return -abs(rc);
    rc = orc_nsev_base(D * ups, q_pre, T, M, contspec, XI, kappa, nse_disc, cstype, normalization_flag);
    free(q_pre);
    if (rc != ORC_SUCCESS || !richardson_flag || !contspec || M == 0) return rc;
    Dsub = D / 2; /* CEIL(D/2) on integers, fnft_nsev.c:376 */
    orc_cplx *qsub = NULL;
    rc = orc_nse_preprocess(D, q, eps_t, &Dsub, &qsub, fl, nse_disc);
    if (rc != ORC_SUCCESS) return -abs(rc);
    const size_t cs_len = M * (cstype == ORC_CS_RHO ? 1 : (cstype == ORC_CS_AB ? 2 : 3));
    orc_cplx *csub = malloc(cs_len * sizeof(orc_cplx));
    if (!csub) { free(qsub); return ORC_EC_NOMEM; }
    const double Tsub[2] = {T[0] + (double)fl[0] * eps_t, T[0] + (double)fl[1] * eps_t};
    const double eps_sub = (Tsub[1] - Tsub[0]) / (double)(Dsub - 1);
    rc = orc_nsev_base(Dsub * ups, qsub, Tsub, M, csub, XI, kappa, nse_disc, cstype, normalization_flag);
    if (rc == ORC_SUCCESS) {
        const double order = (ups == 2) ? 4.0 : 2.0;
        const double scl_num = pow(eps_sub / eps_t, order), scl_den = scl_num - 1.0;
        const double dxi = (XI[1] - XI[0]) / (double)(M - 1);
        const double pi = acos(-1.0);
        for (size_t i = 0; i < M; i++)
            if (fabs(XI[0] + dxi * (double)i) < 0.9 * pi / (2.0 * eps_sub))
                for (size_t j = 0; j < cs_len; j += M)
                    contspec[i + j] = (scl_num * contspec[i + j] - csub[i + j]) / scl_den;
    }
    free(qsub);
    free(csub);
    return rc;
}

int orc_fnft_nsev(size_t D, const orc_cplx *q, const double *T, size_t M, orc_cplx *contspec,
                  const double *XI, int kappa, int nse_disc, int cstype, int normalization_flag)
{
    return orc_fnft_nsev_ex(D, q, T, M, contspec, XI, kappa, nse_disc, cstype, normalization_flag, 0);
}

/* D = number of preprocessed samples (D_given * upsampling) */
static int orc_nsev_base(size_t D, const orc_cplx *q, const double *T, size_t M, orc_cplx *contspec,
                         const double *XI, int kappa, int nse_disc, int cstype, int normalization_flag)
{
    if (D < 2 || !q || !T || !(T[0] < T[1])) return ORC_EC_INVALID_ARGUMENT;
    if (contspec && (!XI || !(XI[0] < XI[1]))) return ORC_EC_INVALID_ARGUMENT;
    if (abs(kappa) != 1) return ORC_EC_INVALID_ARGUMENT;
    const size_t numel = orc_nse_fscatter_numel(D, nse_disc);
    if (numel == 0) return ORC_EC_INVALID_ARGUMENT;
    const double eps_t = (T[1] - T[0]) / (double)(D / orc_nse_upsampling(nse_disc) - 1);
    orc_cplx *tm = malloc(numel * sizeof(orc_cplx));
    if (!tm) return ORC_EC_NOMEM;
    size_t deg = 0;
    int32_t W = 0;
    double t0 = now_s();
    int rc = orc_nse_fscatter(D, q, eps_t, kappa, tm, &deg, normalization_flag ? &W : NULL, nse_disc);
    double t1 = now_s();
    if (rc == ORC_SUCCESS && contspec && M > 0)
        rc = orc_nsev_contspec(deg, W, tm, T, D, XI, M, contspec, nse_disc, cstype);
    double t2 = now_s();
    g_timings[0] = t1 - t0;
    g_timings[1] = t2 - t1;
    free(tm);
    /* subroutine failures surface as -abs(ec), fnft__errwarn.h:50-57,101 */
    return rc == ORC_SUCCESS ? rc : -abs(rc);
}

/* ------------------------------------------------------------------------------------------ */
/* Korteweg-de Vries, vanishing boundaries (fnft_kdvv)                                          */
/* ------------------------------------------------------------------------------------------ */

/* fnft__kdv_discretization.c:86-150: the 18 2SPLIT schemes share their names with the AKNS ones;
 * both enumerations list them in the same order, AKNS has 2SPLIT2_MODAL in front */
int orc_kdv_to_akns(int kdv_disc)
{
    return (kdv_disc >= ORC_KDV_2SPLIT1A && kdv_disc <= ORC_KDV_2SPLIT8B) ? kdv_disc + 1 : -1;
}

/* fnft__kdv_fscatter.c:36-43 */
size_t orc_kdv_fscatter_numel(size_t D, int kdv_disc)
{
    const int a = orc_kdv_to_akns(kdv_disc);
    return a < 0 ? 0 : orc_akns_fscatter_numel(D, a);
}

/* fnft__kdv_fscatter.c:45-83: r = -1 */
int orc_kdv_fscatter(size_t D, const orc_cplx *u, double eps_t, orc_cplx *result, size_t *deg_ptr,
                     int32_t *W_ptr, int kdv_disc)
{
    if (D == 0 || !u || !(eps_t > 0.0) || !result || !deg_ptr) return ORC_EC_INVALID_ARGUMENT;
    const int a = orc_kdv_to_akns(kdv_disc);
    if (a < 0) return ORC_EC_INVALID_ARGUMENT;
    orc_cplx *r = malloc(D * sizeof(orc_cplx));
    if (!r) return ORC_EC_NOMEM;
    for (size_t i = 0; i < D; i++) r[i] = -1.0;
    const int rc = orc_akns_fscatter(D, u, r, eps_t, result, deg_ptr, W_ptr, a);
    free(r);
    return rc;
}

/* fnft_kdvv.c:59-209: transfer matrix without normalisation, entries 12 and 22 on the grid
 * -(XI0 + i*eps_xi), reflection coefficient e^{2 i xi (T1 + eps/2)} H12 / (2 i xi H22 - H12) */
int orc_fnft_kdvv(size_t D, const orc_cplx *u, const double *T, size_t M, orc_cplx *contspec,
                  const double *XI, int kdv_disc)
{
    if (D < 2 || !u || !T || !(T[0] < T[1]) || !contspec || !XI || !(XI[0] < XI[1]))
        return ORC_EC_INVALID_ARGUMENT;
    const size_t numel = orc_kdv_fscatter_numel(D, kdv_disc);
    if (numel == 0) return ORC_EC_INVALID_ARGUMENT;
    orc_cplx *tm = malloc(numel * sizeof(orc_cplx)), *H = malloc(2 * M * sizeof(orc_cplx));
    if (!tm || !H) { free(tm); free(H); return ORC_EC_NOMEM; }
    const double eps_t = (T[1] - T[0]) / (double)(D - 1);
    const double eps_xi = (XI[1] - XI[0]) / (double)(M - 1);
    size_t deg = 0;
    int rc = orc_kdv_fscatter(D, u, eps_t, tm, &deg, NULL, kdv_disc);
    if (rc == ORC_SUCCESS) {
        const double deg1 = (double)orc_akns_degree(orc_kdv_to_akns(kdv_disc));
        const double bc = 0.5; /* fnft__akns_discretization.c:72-109 */
        const orc_cplx V = cexp(-2.0 * I * eps_xi * eps_t / deg1);
        const orc_cplx A = cexp(2.0 * I * XI[0] * eps_t / deg1);
        orc_cplx *H12 = H, *H22 = H + M;
        rc = orc_poly_chirpz(deg, tm + (deg + 1), A, V, M, H12);
        if (rc == ORC_SUCCESS) rc = orc_poly_chirpz(deg, tm + 3 * (deg + 1), A, V, M, H22);
        if (rc == ORC_SUCCESS) {
            for (size_t i = 0; i < M; i++) {
                const double xi = -XI[0] - (double)i * eps_xi;
                if (kdv_disc == ORC_KDV_2SPLIT2A) H12[i] /= cexp(I * xi * eps_t / deg1); /* :186-195 */
                contspec[i] = cexp(2.0 * I * xi * (T[1] + bc * eps_t)) * H12[i];
                contspec[i] /= 2.0 * I * xi * H22[i] - H12[i];
            }
        }
    }
    free(tm);
    free(H);
    return rc == ORC_SUCCESS ? rc : -abs(rc);
}

/* ------------------------------------------------------------------------------------------ */
/* discrete spectrum (bound states) -- the slow scatterer the reference refines with            */
/* ------------------------------------------------------------------------------------------ */

/* fnft__nse_scatter_bound_states.c:40-668 for the two base schemes the splitting discretizations
 * fall back to (fnft_nsev.c:669-676, :937-942): Boffetta-Osborne (ups = 1) and CF4_2 (ups = 2,
 * q already preprocessed, spectral parameter scaled by w0 + w1 = 1/2, derivative by 1/2).
 * One step with constant potential: U = [[ch - i l sh, q sh], [r sh, ch + i l sh]],
 * k^2 = q r - l^2, ch = cosh(k eps), sh = sinh(k eps)/k, and dU/dl from dk/dl = -l/k.
 * phi starts as (e^{-i lam (T0 - eps/2)}, 0) and is carried to T1 together with d phi/d lam;
 * psi starts as (0, e^{i lam (T1 + eps/2)}) at T1 and is carried back with eps -> -eps.
 * a = phi1(T1) e^{i lam (T1+eps/2)}; b = phi1/psi1 at the grid point where
 * |log|(phi2/psi2)/(phi1/psi1)||/2 is smallest (:640-652).  r = -conj(q) (kappa = +1). */
int orc_nse_scatter_bound_states(size_t D, const orc_cplx *q, const double *T, size_t K,
                                 const orc_cplx *lam, orc_cplx *a_vals, orc_cplx *aprime_vals,
                                 orc_cplx *b_vals, int ups, int skip_b)
{
    if (D == 0 || !q || !T || !lam || !a_vals || !aprime_vals || !b_vals) return ORC_EC_INVALID_ARGUMENT;
    if (ups != 1 && ups != 2) return ORC_EC_INVALID_ARGUMENT;
    if (ups == 2 && D % 2 != 0) return ORC_EC_OTHER;
    const size_t Dg = D / (size_t)ups;
    const double eps = (T[1] - T[0]) / (double)(Dg - 1);
    const double lw = (ups == 2) ? 0.5 : 1.0, scl = (ups == 2) ? 0.5 : 1.0, bc = 0.5;
    orc_cplx *P1 = malloc((Dg + 1) * sizeof(orc_cplx)), *P2 = malloc((Dg + 1) * sizeof(orc_cplx));
    orc_cplx *S1 = malloc((Dg + 1) * sizeof(orc_cplx)), *S2 = malloc((Dg + 1) * sizeof(orc_cplx));
    if (!P1 || !P2 || !S1 || !S2) { free(P1); free(P2); free(S1); free(S2); return ORC_EC_NOMEM; }
    for (size_t e = 0; e < K; e++) {
        const orc_cplx lc = lam[e], l = lc * lw;
        orc_cplx p1 = cexp(-I * lc * (T[0] - eps * bc)), p2 = 0.0;
        orc_cplx d1 = p1 * (-I * (T[0] - eps * bc)), d2 = 0.0;
        P1[0] = p1; P2[0] = p2;
        size_t ng = 0;
        int count = ups - 1;
        for (size_t n = 0; n < D; n++) {
            const orc_cplx qn = q[n], rn = -conj(q[n]);
            const orc_cplx ks = qn * rn - l * l, k = csqrt(ks);
            const orc_cplx ch = ccosh(k * eps);
            const orc_cplx sh = (ks != 0.0) ? csinh(k * eps) / k : (orc_cplx)eps;
            const orc_cplx g = (eps * ch - sh) / ks;               /* -d(sh)/dl / l */
            const orc_cplx u00 = ch - I * l * sh, u01 = qn * sh, u10 = rn * sh, u11 = ch + I * l * sh;
            const orc_cplx v00 = -eps * l * sh - I * sh + I * l * l * g;
            const orc_cplx v11 = -eps * l * sh + I * sh - I * l * l * g;
            const orc_cplx v01 = -qn * l * g, v10 = -rn * l * g;
            const orc_cplx n1 = v00 * p1 + v01 * p2 + u00 * d1 + u01 * d2;
            d2 = v10 * p1 + v11 * p2 + u10 * d1 + u11 * d2;
            d1 = n1;
            const orc_cplx t = u10 * p1 + u11 * p2;
            p1 = u00 * p1 + u01 * p2;
            p2 = t;
            if (count == 0) { count = ups - 1; ng++; P1[ng] = p1; P2[ng] = p2; } else count--;
        }
        const orc_cplx ph = cexp(I * lc * (T[1] + eps * bc));
        a_vals[e] = P1[Dg] * ph;
        aprime_vals[e] = scl * (d1 * ph + (I * (T[1] + eps * bc)) * a_vals[e]);
        if (skip_b) continue;
        orc_cplx s1 = 0.0, s2 = ph;
        S1[Dg] = s1; S2[Dg] = s2;
        ng = Dg;
        count = ups - 1;
        for (size_t n = D; n-- > 0;) {
            const orc_cplx qn = q[n], rn = -conj(q[n]);
            const orc_cplx ks = qn * rn - l * l, k = csqrt(ks);
            const orc_cplx ch = ccosh(-k * eps);
            const orc_cplx sh = (ks != 0.0) ? csinh(-k * eps) / k : (orc_cplx)(-eps);
            const orc_cplx u00 = ch - I * l * sh, u01 = qn * sh, u10 = rn * sh, u11 = ch + I * l * sh;
            const orc_cplx t = u10 * s1 + u11 * s2;
            s1 = u00 * s1 + u01 * s2;
            s2 = t;
            if (count == 0) { count = ups - 1; ng--; S1[ng] = s1; S2[ng] = s2; } else count--;
        }
        double best = INFINITY;
        for (size_t n = 0; n <= Dg; n++) {
            const double m = fabs(0.5 * log(cabs((P2[n] / S2[n]) / (P1[n] / S1[n]))));
            if (m < best) { b_vals[e] = P1[n] / S1[n]; best = m; }
        }
    }
    free(P1); free(P2); free(S1); free(S2);
    return ORC_SUCCESS;
}

/* fnft__nse_scatter_matrix (src/private/fnft__nse_scatter_matrix.c:33-86 -> fnft__akns_scatter_matrix.c:112-230), BO
 * (ups = 1) and CF4_2 (ups = 2: two preprocessed samples per step, lambda/2 per sample, derivative times 1/2):
 * S = T_{D-1} ... T_0 with T_n = [[U, 0], [U', U]]; result: 8 values per lambda [S11 S12 S21 S22 S11' S12' S21' S22']
 * (4 without the derivative).  r = -kappa conj(q). */
int orc_nse_scatter_matrix(size_t D, const orc_cplx *q, double eps_t, int kappa, size_t K, const orc_cplx *lam,
                           orc_cplx *result, int ups, int derivative)
{
    if (D == 0 || !q || !(eps_t > 0) || (kappa != 1 && kappa != -1) || K == 0 || !lam || !result)
        return ORC_EC_INVALID_ARGUMENT;
    if (ups != 1 && ups != 2) return ORC_EC_INVALID_ARGUMENT;
    if (ups == 2 && D % 2 != 0) return ORC_EC_OTHER;
    const double lw = (ups == 2) ? 0.5 : 1.0, scl = (ups == 2) ? 0.5 : 1.0;
    for (size_t e = 0; e < K; e++) {
        const orc_cplx l = lam[e] * lw;
        orc_cplx m00 = 1.0, m01 = 0.0, m10 = 0.0, m11 = 1.0, d00 = 0.0, d01 = 0.0, d10 = 0.0, d11 = 0.0;
        for (size_t n = 0; n < D; n++) {
            const orc_cplx qn = q[n], rn = -(double)kappa * conj(q[n]);
            const orc_cplx ks = qn * rn - l * l, k = csqrt(ks);
            const orc_cplx ch = ccosh(k * eps_t);
            const orc_cplx sh = (ks != 0.0) ? csinh(k * eps_t) / k : (orc_cplx)eps_t;
            const orc_cplx u1 = l * sh * I;
            const orc_cplx u00 = ch - u1, u01 = qn * sh, u10 = rn * sh, u11 = ch + u1;
            if (derivative) {
                const orc_cplx chi = ch / ks;
                const orc_cplx ud1 = eps_t * l * l * chi * I, ud2 = l * (eps_t * ch - sh) / ks;   /* :184-185 */
                const orc_cplx v00 = ud1 - (l * eps_t + I + (l * l * I) / ks) * sh, v01 = -qn * ud2;
                const orc_cplx v10 = -rn * ud2, v11 = -ud1 - (l * eps_t - I - (l * l * I) / ks) * sh;
                const orc_cplx e00 = v00 * m00 + v01 * m10 + u00 * d00 + u01 * d10;
                const orc_cplx e01 = v00 * m01 + v01 * m11 + u00 * d01 + u01 * d11;
                const orc_cplx e10 = v10 * m00 + v11 * m10 + u10 * d00 + u11 * d10;
                const orc_cplx e11 = v10 * m01 + v11 * m11 + u10 * d01 + u11 * d11;
                d00 = e00; d01 = e01; d10 = e10; d11 = e11;
            }
            const orc_cplx f00 = u00 * m00 + u01 * m10, f01 = u00 * m01 + u01 * m11;
            const orc_cplx f10 = u10 * m00 + u11 * m10, f11 = u10 * m01 + u11 * m11;
            m00 = f00; m01 = f01; m10 = f10; m11 = f11;
        }
        orc_cplx *o = result + e * (derivative ? 8 : 4);
        o[0] = m00; o[1] = m01; o[2] = m10; o[3] = m11;
        if (derivative) { o[4] = d00 * scl; o[5] = d01 * scl; o[6] = d10 * scl; o[7] = d11 * scl; }
    }
    return ORC_SUCCESS;
}

/* fnft__misc.c:90-112 */
double orc_l2norm2(size_t N, const orc_cplx *Z, double a, double b)
{
    if (N < 2 || a >= b) return NAN;
    const double h = (b - a) / (double)N;
    double val = 0.5 * h * cabs(Z[0]) * cabs(Z[0]);
    for (size_t i = 1; i < N - 1; i++) val += h * cabs(Z[i]) * cabs(Z[i]);
    val += 0.5 * h * cabs(Z[N - 1]) * cabs(Z[N - 1]);
    return val;
}

