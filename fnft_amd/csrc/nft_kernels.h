// nft_kernels.h -- kernel bodies of the MI355X fnft_nsev hot path.
//
// Reference functions replaced (file:line relative to the FNFT source tree):
//   body_coeffs            src/private/fnft__akns_fscatter.c:116-917 (per-sample 2x2 coefficients),
//                          src/private/fnft__nse_fscatter.c:71-84 (r = -kappa*conj(q))
//   body_pair_school       src/private/fnft__poly_fmult.c:239-328 for tiny degrees (direct products)
//   body_pair_fft          src/private/fnft__poly_fmult.c:239-328,330-374 (one pair product + rescale)
//   body_col_fwd/mid/inv   the same pair product when one transform no longer fits a workgroup
//   body_finalize_scales   src/private/fnft__poly_fmult.c:330-374 (rescale bookkeeping)
//   body_export_tm         src/private/fnft__poly_fmult.c:522-538 (final layout)
//   body_chirp_*           src/private/fnft__poly_chirpz.c:52-95, src/fnft_nsev.c:837-884
//
// HBM layout ("body/tail"): a level holds n matrices of degree d.  Entry e (11,12,21,22) is a
// plane of `plane` complex128; matrix j's coefficients of powers d..1 (highest first, as in the
// reference) are body[e*plane + j*d + k], k < d, and its constant term is tail[e*n + j].  Keeping
// the odd "+1" coefficient out of the body makes every body a power-of-two run of 16-byte
// elements and lets a product of two degree-d matrices use a cyclic transform of length 2d: the
// only aliased coefficient, index 2d, is the product of the two constant terms and is formed
// directly.  scale[j] is a pending power of two (true matrix = stored * scale[j])
// and wexp[j]
// carries the exponents taken out of everything below matrix j (summed up the tree without
// atomics), so that at the root true product = stored * 2^W (fnft__poly_fmult.c:493).
#pragma once
#include "fft_dev.h"

// Timing-ablation hooks (skip loads / transforms / stores of the pair kernels) exist only in
// diagnostic builds made with -DFNFT_AMD_ABLATION; the product build compiles them out.
#ifdef FNFT_AMD_ABLATION
#define FA_DBG(x) (x)
#else
#define FA_DBG(x) 0
#endif

// ---------------------------------------------------------------------------------------------
// parameter blocks
// ---------------------------------------------------------------------------------------------
struct TreeLevel {
    const cplx *body_in;
    const cplx *tail_in;
    const double *scale_in;
    cplx *body_out;
    cplx *tail_out;
    double *scale_out;
    unsigned *max2_out;            // per output matrix, kMax2Slots slots: high dword of max |coef|^2 (split levels)
    const unsigned *max2_in;       // per input matrix, valid when in_pending
    int in_pending;                // inputs come from a split level whose rescale is still pending:
                                   // scale = 2^-a(max2_in), exponent = wexp_in + a(max2_in)
    const int *wexp_in;            // per input matrix: power-of-two exponent taken out so far
    int *wexp_out;                 // per output matrix: wexp_in[2P] + wexp_in[2P+1] + this level's
    size_t plane;                  // body plane stride, elements
    int n_in;                      // matrices entering this level (all signals)
    int d;                         // their degree
    int pairs_per_signal;          // (n_in/2)/batch
    const cplx *tw;                // exp(-2 pi i j/N) table of the transform length used
    int ne;                        // stored entries per matrix: 4 general, 2 symmetric (11, 21)
    int kappa;                     // +1 focusing / -1 defocusing (symmetric form only)
    int dbg;                       // diagnostic builds (-DFNFT_AMD_ABLATION) only: 1 skip transforms, 2 skip loads, 4 skip stores
    const cplx *twm[3];            // multi-level kernel (body_multi_fft): tables for N0, 2*N0, 4*N0
    const cplx *twx = nullptr;     // real-coefficient path (nft_real.h): exp(-2 pi i j/(4M)), M = transform length
};

// floor(log2(sqrt(m2))) for m2 > 0 (normal), exactly, from the exponent field
FA_HD int half_exponent(double m2)
{
    union { double d; unsigned long long u; } cv;
    cv.d = m2;
    const int e = (int)((cv.u >> 52) & 0x7ffull) - 1023;
    return e >> 1;  // arithmetic shift = floor(e/2)
}
FA_HD double pow2i(int a)
{
    union { double d; unsigned long long u; } cv;
    cv.u = (unsigned long long)(a + 1023) << 52;
    return cv.d;
}
FA_HD unsigned long long dbits(double x)
{
    union { double d; unsigned long long u; } cv;
    cv.d = x;
    return cv.u;
}
FA_HD double bitsd(unsigned long long u)
{
    union { double d; unsigned long long u; } cv;
    cv.u = u;
    return cv.d;
}

// The running maximum of |coef|^2 of a product matrix of a split level is kept as the HIGH dword of
// the double (positive doubles order like their high dwords, and only the exponent field is used), in
// kMax2Slots slots per matrix: the waves of the column kernels spread their atomicMax over the slots
// (the top levels have one to four matrices, and thousands of atomics on one address serialise at the
// memory side); readers take the maximum over the slots.
constexpr int kMax2Slots = 64;
FA_HD int exponent_of_max2(unsigned hi)
{
    // m2 > 0 and m2 < 1e300 (high dword 0x7E37E43C): floor(log2(sqrt(m2))) from the exponent field
    if (hi == 0u || hi >= 0x7E37E43Cu) return 0;
    const int e = (int)((hi >> 20) & 0x7ffu) - 1023;
    return e >> 1;
}
FA_DEV int max2_slot() { return (FA_BID_Y * 5 + FA_BID * (FA_BDIM / 64) + FA_TID / 64) & (kMax2Slots - 1); }
// exponent still to be taken out of input matrix `mat` of a level (see TreeLevel::in_pending).
// WAVE-UNIFORM call only: every lane of the wave active, same `mat` (the slots are read one per lane
// and reduced across the wave).
FA_DEV int level_in_pending_exp(const TreeLevel &L, long long mat)
{
    return L.in_pending ? exponent_of_max2(fa_slots_max_u32(L.max2_in + (size_t)mat * kMax2Slots)) : 0;
}
// pending scale / accumulated exponent of input matrix `mat` of a level (wave-uniform calls only)
FA_DEV double level_in_scale(const TreeLevel &L, long long mat)
{
    if (!L.in_pending) return L.scale_in[mat];
    return pow2i(-level_in_pending_exp(L, mat));
}
FA_DEV int level_in_wexp(const TreeLevel &L, long long mat)
{
    return L.wexp_in[mat] + level_in_pending_exp(L, mat);
}

// ---------------------------------------------------------------------------------------------
// complex elementary functions needed by the coefficient kernel
// (libm csqrt/ccos/csin of src/private/fnft__akns_fscatter.c:46-59, fnft__misc.c:306-314)
// ---------------------------------------------------------------------------------------------
FA_DEV cplx c_sqrt(cplx z)
{
    if (z.x == 0.0 && z.y == 0.0) return cmake(0.0, z.y);
    const double ax = fabs(z.x), ay = fabs(z.y);
    const double hm = sqrt(fma(ax, ax, ay * ay));
    const double t = sqrt(0.5 * (ax + hm));
    if (z.x >= 0.0) return cmake(t, z.y / (2.0 * t));
    return cmake(ay / (2.0 * t), z.y < 0.0 ? -t : t);
}
FA_DEV cplx c_cos(cplx z)
{
    double s, c;
    fa_sincos(z.x, &s, &c);
    return cmake(c * cosh(z.y), -s * sinh(z.y));
}
FA_DEV cplx c_sin(cplx z)
{
    double s, c;
    fa_sincos(z.x, &s, &c);
    return cmake(s * cosh(z.y), c * sinh(z.y));
}
// 1/x where a division sits in an inner loop or on a dependent chain (Aberth sums, layer peeling): hardware reciprocal + two Newton steps (no scaling / fix-up of the IEEE division:
// the arguments are squared distances of estimates, far from the ends of the exponent range)
FA_DEV double aberth_rcp(double x)
{
    double r = fa_rcp_approx(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

FA_DEV cplx c_div(cplx a, cplx b)
{
    // Smith's algorithm
    if (fabs(b.x) >= fabs(b.y)) {
        const double r = b.y / b.x, den = b.x + b.y * r;
        return cmake((a.x + a.y * r) / den, (a.y - a.x * r) / den);
    }
    const double r = b.x / b.y, den = b.x * r + b.y;
    return cmake((a.x * r + a.y) / den, (a.y * r - a.x) / den);
}
FA_DEV cplx c_sinc(cplx z)
{
    if (sqrt(cnorm2(z)) >= 1.0e-8) return c_div(c_sin(z), z);
    return c_cos(z * 0.57735026918962576451);  // cos(z/sqrt(3)), fnft__misc.c:313
}

// expm([[0,q],[r,0]]*h) = [[c, q*s],[r*s, c]]  (fnft__akns_fscatter.c:46-59)
struct StepExp {
    cplx c, qs, rs;
};
FA_DEV StepExp zero_freq_step(double h, cplx q, cplx r)
{
    StepExp e;
    const cplx mqr = cmake(-(q.x * r.x - q.y * r.y), -(q.x * r.y + q.y * r.x));
    const cplx Delta = c_sqrt(mqr) * h;
    const cplx del = c_sinc(Delta) * h;
    e.c = c_cos(Delta);
    e.qs = q * del;
    e.rs = r * del;
    return e;
}

// ---------------------------------------------------------------------------------------------
// K1: per-sample coefficients
// ---------------------------------------------------------------------------------------------
struct CoeffParams {
    const cplx *q;       // batch*D
    const cplx *r;       // batch*D or NULL (then r = -kappa*conj(q))
    cplx *body;          // 4 planes
    cplx *tail;          // 4 planes of n = batch*Dpad
    double *scale;       // n
    int *wexp;           // n: exponent taken out (0 at level 0, set by the leaf kernel)
    int *status;         // bit 0: MODAL step-size check failed
    int ne;              // stored entries per matrix: 4 general, 2 symmetric (11 and 21)
    size_t plane;
    double eps_t;
    int D, Dpad, batch, kappa;
    int disc;            // fnft__akns_discretization_t ordinal
    int deg;
    int real_out = 0;    // real-coefficient path (nft_real.h): body/tail are arrays of double (real parts only);
                         // a sample with a non-zero imaginary part sets bit 3 of the status word
};

// coefficient tables: P[e][k], k <= deg, highest power first
template <int DEG> struct CoefMat {
    cplx p[4][DEG + 1];
};

template <int DEG> FA_DEV void coeffs_zero(CoefMat<DEG> &m)
{
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
        for (int k = 0; k <= DEG; k++) m.p[e][k] = cmake(0.0, 0.0);
}

// Returns false when the MODAL step-size condition is violated (fnft__akns_fscatter.c:122-126).
template <int DEG> FA_DEV bool sample_coeffs(int disc, double eps_t, cplx q, cplx r, CoefMat<DEG> &m)
{
    coeffs_zero(m);
    const double h = eps_t / (double)DEG;
    if constexpr (DEG == 1) {
        if (disc == 0) {  // 2SPLIT2_MODAL, :118-148
            bool ok = true;
            if (q.x == r.x && eps_t * sqrt(cnorm2(q)) >= 1.0) ok = false;
            const cplx eq = q * eps_t, er = r * eps_t;
            const cplx one_m = cmake(1.0 - (eq.x * er.x - eq.y * er.y), -(eq.x * er.y + eq.y * er.x));
            const cplx s = c_div(cmake(1.0, 0.0), c_sqrt(one_m));
            m.p[0][1] = s;
            m.p[1][0] = s * eq;
            m.p[2][1] = s * er;
            m.p[3][0] = s;
            return ok;
        }
        if (disc == 1) {  // 2SPLIT1A, :150-176
            const StepExp e = zero_freq_step(h, q, r);
            m.p[0][1] = e.c; m.p[1][1] = e.qs; m.p[2][0] = e.rs; m.p[3][0] = e.c;
            return true;
        }
        if (disc == 2 || disc == 3) {  // 2SPLIT1B / 2SPLIT2A, :178-203
            const StepExp e = zero_freq_step(h, q, r);
            m.p[0][1] = e.c; m.p[1][0] = e.qs; m.p[2][1] = e.rs; m.p[3][0] = e.c;
            return true;
        }
        if (disc == 4) {  // 2SPLIT2B, :204-228
            const StepExp e = zero_freq_step(0.5 * h, q, r);
            m.p[0][0] = e.qs * e.rs; m.p[0][1] = e.c * e.c;
            m.p[1][0] = m.p[1][1] = e.c * e.qs;
            m.p[2][0] = m.p[2][1] = e.c * e.rs;
            m.p[3][0] = m.p[0][1]; m.p[3][1] = m.p[0][0];
            return true;
        }
        // 2SPLIT2S, :230-254
        const StepExp e = zero_freq_step(h, q, r);
        m.p[0][1] = e.c;
        m.p[1][0] = m.p[1][1] = e.qs * 0.5;
        m.p[2][0] = m.p[2][1] = e.rs * 0.5;
        m.p[3][0] = e.c;
        return true;
    } else if constexpr (DEG == 2) {
        if (disc == 8) {  // 2SPLIT3S, :331-361
            const StepExp e1 = zero_freq_step(h, q, r), e2 = zero_freq_step(2 * h, q, r);
            m.p[0][0] = (e1.qs * e1.rs) * (2.0 / 3.0);
            m.p[0][2] = ((e1.c * e1.c) * 2.0 + e2.c) * (1.0 / 3.0);
            m.p[1][0] = m.p[1][2] = ((e1.c * e1.qs) * 4.0 - e2.qs) * (1.0 / 6.0);
            m.p[1][1] = e2.qs * (2.0 / 3.0);
            m.p[2][0] = m.p[2][2] = ((e1.c * e1.rs) * 4.0 - e2.rs) * (1.0 / 6.0);
            m.p[2][1] = e2.rs * (2.0 / 3.0);
            m.p[3][0] = m.p[0][2]; m.p[3][2] = m.p[0][0];
            return true;
        }
        // 2SPLIT4B (and 4SPLIT4B), :402-433
        const StepExp eh = zero_freq_step(0.5 * h, q, r), e1 = zero_freq_step(h, q, r);
        const double t3 = 1.0 / 3.0;
        m.p[0][0] = ((e1.c * eh.qs * eh.rs) * 4.0 - e1.qs * e1.rs) * t3;
        m.p[0][1] = (e1.qs * eh.c * eh.rs + e1.rs * eh.c * eh.qs) * (4.0 * t3);
        m.p[0][2] = ((e1.c * eh.c * eh.c) * 4.0 - e1.c * e1.c) * t3;
        m.p[1][0] = m.p[1][2] = ((e1.c * eh.c * eh.qs) * 4.0 - e1.c * e1.qs) * t3;
        m.p[1][1] = (e1.qs * eh.c * eh.c + e1.rs * eh.qs * eh.qs) * (4.0 * t3);
        m.p[2][0] = m.p[2][2] = ((e1.c * eh.c * eh.rs) * 4.0 - e1.c * e1.rs) * t3;
        m.p[2][1] = (e1.rs * eh.c * eh.c + e1.qs * eh.rs * eh.rs) * (4.0 * t3);
        m.p[3][0] = m.p[0][2]; m.p[3][1] = m.p[0][1]; m.p[3][2] = m.p[0][0];
        return true;
    } else if constexpr (DEG == 3) {
        const StepExp e1 = zero_freq_step(h, q, r), e2 = zero_freq_step(2 * h, q, r),
                      e3 = zero_freq_step(3 * h, q, r);
        const double n8 = 9.0 / 8.0, i8 = 1.0 / 8.0;
        if (disc == 6) {  // 2SPLIT3A, :256-292
            m.p[0][1] = (e1.rs * e2.qs) * n8; m.p[0][3] = ((e1.c * e2.c) * 9.0 - e3.c) * i8;
            m.p[1][1] = (e1.c * e2.qs) * n8;  m.p[1][3] = ((e1.qs * e2.c) * 9.0 - e3.qs) * i8;
            m.p[2][0] = ((e1.rs * e2.c) * 9.0 - e3.rs) * i8; m.p[2][2] = (e1.c * e2.rs) * n8;
            m.p[3][0] = m.p[0][3]; m.p[3][2] = (e1.qs * e2.rs) * n8;
        } else {  // 2SPLIT3B, :294-330
            m.p[0][1] = (e1.qs * e2.rs) * n8; m.p[0][3] = ((e1.c * e2.c) * 9.0 - e3.c) * i8;
            m.p[1][0] = ((e1.qs * e2.c) * 9.0 - e3.qs) * i8; m.p[1][2] = (e1.c * e2.qs) * n8;
            m.p[2][1] = (e1.c * e2.rs) * n8; m.p[2][3] = ((e1.rs * e2.c) * 9.0 - e3.rs) * i8;
            m.p[3][0] = m.p[0][3]; m.p[3][2] = (e1.rs * e2.qs) * n8;
        }
        return true;
    } else {  // DEG == 4: 2SPLIT4A (and 4SPLIT4A), :362-401
        const StepExp e2 = zero_freq_step(2 * h, q, r), e4 = zero_freq_step(4 * h, q, r);
        const double t3 = 1.0 / 3.0;
        m.p[0][2] = (e2.qs * e2.rs) * (4.0 * t3);
        m.p[0][4] = ((e2.c * e2.c) * 4.0 - e4.c) * t3;
        m.p[1][1] = m.p[1][3] = (e2.c * e2.qs) * (4.0 * t3);
        m.p[1][2] = e4.qs * (-t3);
        m.p[2][1] = m.p[2][3] = (e2.c * e2.rs) * (4.0 * t3);
        m.p[2][2] = e4.rs * (-t3);
        m.p[3][0] = m.p[0][4]; m.p[3][2] = m.p[0][2];
        return true;
    }
}

// one lane per matrix slot j of the padded level 0; slot j of signal b holds sample D-1-j
// (fnft__akns_fscatter.c:120) or the identity pad z^deg*I (fnft__poly_fmult.c:422-438).
template <int DEG> FA_DEV void body_coeffs(const CoeffParams &P)
{
    const long long gid = (long long)FA_BID * FA_BDIM + FA_TID;
    const long long n = (long long)P.batch * P.Dpad;
    if (gid >= n) return;
    const int b = (int)(gid / P.Dpad), j = (int)(gid % P.Dpad);
    CoefMat<DEG> m;
    if (j < P.D) {
        const size_t src = (size_t)b * P.D + (size_t)(P.D - 1 - j);
        const cplx q = P.q[src];
        const cplx r = P.r ? P.r[src] : (P.kappa == 1 ? cmake(-q.x, q.y) : cmake(q.x, -q.y));
        if (!sample_coeffs<DEG>(P.disc, P.eps_t, q, r, m)) fa_atomic_or_i32(P.status, 1);
        if (P.real_out && (q.y != 0.0 || r.y != 0.0)) fa_atomic_or_i32(P.status, 8);
    } else {
        coeffs_zero(m);
        if (P.ne == 4) {
            m.p[0][0] = cmake(1.0, 0.0);
            m.p[3][0] = cmake(1.0, 0.0);
        } else {
            m.p[0][DEG] = cmake(1.0, 0.0);
            m.p[3][0] = cmake(1.0, 0.0);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (P.ne == 2 && (e & 1)) continue;
        const int s = (P.ne == 2) ? (e >> 1) : e;
        if (P.real_out) {   // real-coefficient path: arrays of double
            double *rb = (double *)P.body, *rt = (double *)P.tail;
#pragma unroll
            for (int k = 0; k < DEG; k++) rb[(size_t)s * P.plane + (size_t)gid * DEG + k] = m.p[e][k].x;
            rt[(size_t)s * n + gid] = m.p[e][DEG].x;
            continue;
        }
#pragma unroll
        for (int k = 0; k < DEG; k++) P.body[(size_t)s * P.plane + (size_t)gid * DEG + k] = m.p[e][k];
        P.tail[(size_t)s * n + gid] = m.p[e][DEG];
    }
    P.scale[gid] = 1.0;
    P.wexp[gid] = 0;
}

// Splitting schemes of order 5..8 (fnft__akns_fscatter.c:435-912): the coefficients are sums of
// monomials in the elements of e^{bB} for a few step fractions b; the list ("program") is built
// on the host by nft_schemes.h and is the same for every sample.
struct CoeffProgParams {
    CoeffParams c;
    const double *bfrac;          // nB fractions of eps_t
    const int *tgt_ptr;           // 4*(deg+1)+1
    const double *mw;             // monomial weights
    const unsigned char *mfac;    // maxf ids per monomial, 255 = unused
    int nB, maxf;
};

// One lane per sample.  The step-matrix elements (up to 24 per sample) are indexed by the program, so they live in
// LDS ([element][lane]: the program is the same for every lane, a read is one contiguous line) instead of a
// dynamically indexed register array (= scratch memory); the coefficients are collected in LDS too, 16 per sample
// and round, and leave as runs of 256 bytes (a lane's own coefficients of one power are deg*16 bytes apart).
FA_DEV void body_coeffs_prog(const CoeffProgParams &Q)
{
    FA_LDS_DECL
    cplx *el = (cplx *)FA_LDS_PTR;                       // 24 x 64
    cplx *stage = el + 24 * 64;                          // 16 x 64: one staging round
    const CoeffParams &P = Q.c;
    const int lane = FA_TID;                             // 64 lanes
    const long long gid0 = (long long)FA_BID * FA_BDIM;
    const long long gid = gid0 + lane;
    const long long n = (long long)P.batch * P.Dpad;
    const bool act = gid < n;
    const int b = act ? (int)(gid / P.Dpad) : 0, j = act ? (int)(gid % P.Dpad) : 0;
    const int deg = P.deg;
    const bool pad = j >= P.D;
    if (act && !pad) {
        const size_t src = (size_t)b * P.D + (size_t)(P.D - 1 - j);
        const cplx q = P.q[src];
        const cplx r = P.r ? P.r[src] : (P.kappa == 1 ? cmake(-q.x, q.y) : cmake(q.x, -q.y));
        if (P.real_out && (q.y != 0.0 || r.y != 0.0)) fa_atomic_or_i32(P.status, 8);
        for (int i = 0; i < Q.nB; i++) {
            const StepExp e = zero_freq_step(P.eps_t * Q.bfrac[i], q, r);
            el[(3 * i) * 64 + lane] = e.c;
            el[(3 * i + 1) * 64 + lane] = e.qs;
            el[(3 * i + 2) * 64 + lane] = e.rs;
        }
    }
    const long long nact = (n - gid0 < 64) ? n - gid0 : 64;   // samples of this workgroup
    constexpr int CH = 16;                                    // coefficients per staging round
    for (int e = 0; e < 4; e++) {
        if (P.ne == 2 && (e & 1)) continue;
        const int s = (P.ne == 2) ? (e >> 1) : e;
        cplx *body = P.body + (size_t)s * P.plane + (size_t)gid0 * deg;
        for (int k0 = 0; k0 <= deg; k0 += CH) {
            for (int k = k0; k < k0 + CH && k <= deg; k++) {
                cplx acc = cmake(0.0, 0.0);
                if (pad) {
                    // z^deg * I in the general form, diag(1, z^deg) in the symmetric form (body_coeffs)
                    const bool one = (P.ne == 4) ? ((e == 0 || e == 3) && k == 0)
                                                 : ((e == 0 && k == deg) || (e == 3 && k == 0));
                    if (one) acc = cmake(1.0, 0.0);
                } else if (act) {
                    const int t = e * (deg + 1) + k;
                    for (int m = Q.tgt_ptr[t]; m < Q.tgt_ptr[t + 1]; m++) {
                        cplx prod = cmake(Q.mw[m], 0.0);
                        for (int f = 0; f < Q.maxf; f++) {
                            const int id = Q.mfac[(size_t)m * Q.maxf + f];
                            if (id == 255) break;
                            prod = prod * el[id * 64 + lane];
                        }
                        acc = acc + prod;
                    }
                }
                if (k < deg) stage[lane * CH + (k - k0)] = acc;
                else if (act) {
                    if (P.real_out) ((double *)P.tail)[(size_t)s * n + gid] = acc.x;
                    else P.tail[(size_t)s * n + gid] = acc;
                }
            }
            FA_SYNC();
            const int cw = (deg - k0 < CH) ? deg - k0 : CH;   // body coefficients of this round (the last one is the tail)
            double *rbody = (double *)P.body + (size_t)s * P.plane + (size_t)gid0 * deg;
            for (long long i = lane; i < nact * cw; i += 64) {
                const long long smp = i / cw;
                const int kk = (int)(i % cw);
                if (P.real_out) rbody[smp * deg + k0 + kk] = stage[smp * CH + kk].x;
                else body[smp * deg + k0 + kk] = stage[smp * CH + kk];
            }
            FA_SYNC();
        }
    }
    if (act) {
        P.scale[gid] = 1.0;
        P.wexp[gid] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// K1+: leaf kernel = coefficients of SPT consecutive samples AND their ordered product, all in
// registers of one lane (levels 0 .. log2(SPT)-1 of the tree never touch HBM).  Replaces the
// reference's coefficient loop plus the first log2(SPT) rounds of fnft__poly_fmult.c:460-519.
// Output: matrices of degree DEG*SPT in the body/tail layout, already rescaled (scale = 1).
// ---------------------------------------------------------------------------------------------
struct LeafParams {
    CoeffParams c;       // q, r, eps_t, ... ; body/tail/scale/wexp are the OUTPUT level
    int spt;
};

template <int DEG, int SPT, int S> struct LeafStep {
    // acc (degree DEG*S) <- acc * u (degree DEG)
    static FA_DEV void mul(cplx (&acc)[4][DEG * SPT + 1], const CoefMat<DEG> &u)
    {
        constexpr int DC = DEG * S;
#pragma unroll
        for (int row = 0; row < 2; row++) {
            cplx a0[DC + 1], a1[DC + 1];
#pragma unroll
            for (int i = 0; i <= DC; i++) {
                a0[i] = acc[2 * row][i];
                a1[i] = acc[2 * row + 1][i];
            }
#pragma unroll
            for (int col = 0; col < 2; col++) {
                cplx r[DC + DEG + 1];
#pragma unroll
                for (int k = 0; k <= DC + DEG; k++) r[k] = cmake(0.0, 0.0);
#pragma unroll
                for (int i = 0; i <= DC; i++)
#pragma unroll
                    for (int j = 0; j <= DEG; j++) {
                        r[i + j] = cfma(a0[i], u.p[col][j], r[i + j]);
                        r[i + j] = cfma(a1[i], u.p[2 + col][j], r[i + j]);
                    }
#pragma unroll
                for (int k = 0; k <= DC + DEG; k++) acc[2 * row + col][k] = r[k];
            }
        }
    }
};

template <int DEG> FA_DEV bool leaf_sample(const CoeffParams &P, int b, long long j, CoefMat<DEG> &m)
{
    if (j < P.D) {
        const size_t src = (size_t)b * P.D + (size_t)(P.D - 1 - j);
        const cplx q = P.q[src];
        const cplx r = P.r ? P.r[src] : (P.kappa == 1 ? cmake(-q.x, q.y) : cmake(q.x, -q.y));
        if (P.real_out && (q.y != 0.0 || r.y != 0.0)) fa_atomic_or_i32(P.status, 8);
        return sample_coeffs<DEG>(P.disc, P.eps_t, q, r, m);
    }
    coeffs_zero(m);
    if (P.ne == 4) {   // z^deg * I, fnft__poly_fmult.c:422-438
        m.p[0][0] = cmake(1.0, 0.0);
        m.p[3][0] = cmake(1.0, 0.0);
    } else {           // zero-sample matrix diag(1, z^deg): keeps the NSE symmetry
        m.p[0][DEG] = cmake(1.0, 0.0);
        m.p[3][0] = cmake(1.0, 0.0);
    }
    return true;
}

template <int DEG, int SPT, int S> struct LeafLoop {
    static FA_DEV void run(const CoeffParams &P, int b, long long j0, cplx (&acc)[4][DEG * SPT + 1], bool &ok)
    {
        if constexpr (S < SPT) {
            CoefMat<DEG> u;
            ok = leaf_sample<DEG>(P, b, j0 + S, u) && ok;
            LeafStep<DEG, SPT, S>::mul(acc, u);
            LeafLoop<DEG, SPT, S + 1>::run(P, b, j0, acc, ok);
        }
    }
};

template <int DEG, int SPT> FA_DEV void body_leaf(const LeafParams &LP)
{
    FA_LDS_DECL
    cplx *stage = (cplx *)FA_LDS_PTR;  // FA_BDIM * d elements
    const CoeffParams &P = LP.c;
    const int tid = FA_TID;
    const long long g0 = (long long)FA_BID * FA_BDIM;
    const long long gid = g0 + tid;
    const long long per = P.Dpad / SPT;            // output matrices per signal
    const long long n_out = (long long)P.batch * per;
    const bool active = gid < n_out;
    const int b = active ? (int)(gid / per) : 0;
    const long long j0 = active ? (gid % per) * SPT : 0;
    constexpr int d = DEG * SPT;
    cplx acc[4][d + 1];
    bool ok = true;
    double sc = 1.0;
    if (active) {
        {
            CoefMat<DEG> m;
            ok = leaf_sample<DEG>(P, b, j0, m);
#pragma unroll
            for (int e = 0; e < 4; e++) {
#pragma unroll
                for (int k = 0; k <= DEG; k++) acc[e][k] = m.p[e][k];
            }
        }
        LeafLoop<DEG, SPT, 1>::run(P, b, j0, acc, ok);
        if (!ok) fa_atomic_or_i32(P.status, 1);
        double m2 = 0.0;
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int k = 0; k <= d; k++) m2 = fmax(m2, cnorm2(acc[e][k]));
        int a = 0;
        if (m2 > 0.0 && m2 < 1.0e300) {
            a = half_exponent(m2);
            sc = pow2i(-a);
        }
        P.scale[gid] = 1.0;
        P.wexp[gid] = a;
    }
    // bodies leave through LDS so that lanes write consecutive 16-byte elements (full lines)
    const long long nvalid = (n_out - g0 < FA_BDIM) ? n_out - g0 : FA_BDIM;
    const int total = (int)nvalid * d;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (P.ne == 2 && (e & 1)) continue;          // symmetric form stores 11 and 21 only
        const int s = (P.ne == 2) ? (e >> 1) : e;    // destination plane
        if (active) {
#pragma unroll
            for (int k = 0; k < d; k++) stage[(size_t)tid * d + ((k + tid) % d)] = acc[e][k] * sc;
            if (P.real_out) ((double *)P.tail)[(size_t)s * n_out + gid] = acc[e][d].x * sc;
            else P.tail[(size_t)s * n_out + gid] = acc[e][d] * sc;
        }
        FA_SYNC();
        cplx *out0 = P.body + (size_t)s * P.plane + (size_t)g0 * d;
        double *rout0 = (double *)P.body + (size_t)s * P.plane + (size_t)g0 * d;
        for (int m = tid; m < total; m += FA_BDIM) {
            const int t2 = m / d, k2 = m - t2 * d;
            if (P.real_out) rout0[m] = stage[(size_t)t2 * d + ((k2 + t2) % d)].x;
            else out0[m] = stage[(size_t)t2 * d + ((k2 + t2) % d)];
        }
        FA_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------
// K2a: direct ("schoolbook") pair product, one lane per pair, for degrees too small for an FFT
// ---------------------------------------------------------------------------------------------
template <int DEG> FA_DEV void body_pair_school(const TreeLevel &L)
{
    const long long P = (long long)FA_BID * FA_BDIM + FA_TID;
    const int n_out = L.n_in / 2;
    if (P >= n_out) return;
    constexpr int d = DEG;
    cplx A[4][d + 1], Bm[4][d + 1];
    const double sA = L.scale_in[2 * P], sB = L.scale_in[2 * P + 1];
#pragma unroll
    for (int e = 0; e < 4; e++) {
#pragma unroll
        for (int k = 0; k < d; k++) {
            A[e][k] = L.body_in[(size_t)e * L.plane + (size_t)(2 * P) * d + k] * sA;
            Bm[e][k] = L.body_in[(size_t)e * L.plane + (size_t)(2 * P + 1) * d + k] * sB;
        }
        A[e][d] = L.tail_in[(size_t)e * L.n_in + 2 * P] * sA;
        Bm[e][d] = L.tail_in[(size_t)e * L.n_in + 2 * P + 1] * sB;
    }
    cplx C[4][2 * d + 1];
    double m2 = 0.0;
#pragma unroll
    for (int row = 0; row < 2; row++)
#pragma unroll
        for (int col = 0; col < 2; col++) {
            const int e = 2 * row + col;
#pragma unroll
            for (int k = 0; k <= 2 * d; k++) C[e][k] = cmake(0.0, 0.0);
#pragma unroll
            for (int i = 0; i <= d; i++)
#pragma unroll
                for (int j = 0; j <= d; j++) {
                    C[e][i + j] = cfma(A[2 * row][i], Bm[col][j], C[e][i + j]);
                    C[e][i + j] = cfma(A[2 * row + 1][i], Bm[2 + col][j], C[e][i + j]);
                }
#pragma unroll
            for (int k = 0; k <= 2 * d; k++) m2 = fmax(m2, cnorm2(C[e][k]));
        }
    int a = 0;
    double sc = 1.0;
    if (m2 > 0.0 && m2 < 1.0e300) {
        a = half_exponent(m2);
        sc = pow2i(-a);
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
#pragma unroll
        for (int k = 0; k < 2 * d; k++)
            L.body_out[(size_t)e * L.plane + (size_t)P * (2 * d) + k] = C[e][k] * sc;
        L.tail_out[(size_t)e * n_out + P] = C[e][2 * d] * sc;
    }
    L.scale_out[P] = 1.0;
    L.wexp_out[P] = L.wexp_in[2 * P] + L.wexp_in[2 * P + 1] + a;
}

// ---------------------------------------------------------------------------------------------
// K2b: FFT pair product, one group of N/R lanes per pair, B pairs per workgroup.
//
// General form (NE = 4 stored entries): A's four spectra stay in registers; B is streamed one
// column at a time:  C(:,col) = A * B(:,col)  ->  2 forward transforms, 8 complex multiply-adds
// per bin, 2 inverse transforms, stored; total 8 forward + 4 inverse as in the reference.
//
// Symmetric form (NE = 2, NSE with r = -kappa*conj(q)): every transfer matrix of degree d obeys
//     p22[k] = conj(p11[d-k]),   p12[k] = -kappa*conj(p21[d-k])        (highest power first),
// a property products preserve.  Only the first column (11 -> plane 0, 21 -> plane 1) is stored
// and transformed; on the transform grid  X22 = g*conj(X11),  X12 = -kappa*g*conj(X21)  with
// g[m] = exp(-2 pi i d m / N)  (= (-1)^m when N = 2d).  4 forward + 2 inverse transforms, half
// the HBM traffic, half the registers.  The numbers produced are the ones the general form
// produces up to rounding (tests compare both with the oracle).
//
// IO supplies the loads/stores so that the same body serves a whole tree level (TreeIO) and the
// row step of a transform that is split across workgroups (MidIO).
// ---------------------------------------------------------------------------------------------
// Twiddle tables are copied into LDS once per workgroup: every radix pass needs r-1 table
// entries per butterfly, and at 1-2 waves per SIMD a global (L2) fetch per pass is pure stall.
template <int N, int T> FA_DEV cplx *stage_twiddles(cplx *dst, const cplx *__restrict__ src)
{
    for (int j = FA_TID; j < N; j += T) dst[j] = src[j];
    FA_SYNC_LDS();
    return dst;
}

template <int N, int R, int B, bool DB, bool TWC = false, class IO>
FA_DEV void pair_product_core(IO &io, cplx *lds, const cplx *tw)
{
    const int tid = FA_TID;
    const int c = tid % B, v = tid / B;
    int parity = 0;
    cplx a[4][R];
    // all four loads are issued before the first transform so that their latency overlaps it
#pragma unroll
    for (int e = 0; e < 4; e++) io.load(0, e, a[e], v, c);
#pragma unroll
    for (int e = 0; e < 4; e++) fft_wg<N, R, B, -1, DB, TWC>(a[e], lds, v, c, tw, parity);
#pragma unroll
    for (int col = 0; col < 2; col++) {
        cplx b1[R], b2[R];
        io.load(1, col, b1, v, c);
        io.load(1, 2 + col, b2, v, c);
        fft_wg<N, R, B, -1, DB, TWC>(b1, lds, v, c, tw, parity);
        fft_wg<N, R, B, -1, DB, TWC>(b2, lds, v, c, tw, parity);
#pragma unroll
        for (int i = 0; i < R; i++) {
            const cplx x1 = b1[i], x2 = b2[i];
            b1[i] = cfma(a[1][i], x2, a[0][i] * x1);
            b2[i] = cfma(a[3][i], x2, a[2][i] * x1);
        }
        fft_wg<N, R, B, +1, DB, TWC>(b1, lds, v, c, tw, parity);
        io.store(col, b1, v, c, lds, parity);
        fft_wg<N, R, B, +1, DB, TWC>(b2, lds, v, c, tw, parity);
        io.store(2 + col, b2, v, c, lds, parity);
    }
}

template <int N, int R, int B, bool DB, bool TWC = false, class IO>
FA_DEV void pair_product_core_sym(IO &io, cplx *lds, const cplx *tw, int kappa)
{
    const int tid = FA_TID;
    const int c = tid % B, v = tid / B;
    int parity = 0;
    cplx a11[R], a21[R], b11[R], b21[R];
    const int dbg = io.dbg();

    if (!(dbg & 2)) {
        io.load(0, 0, a11, v, c);
        io.load(0, 1, a21, v, c);
        io.load(1, 0, b11, v, c);
        io.load(1, 1, b21, v, c);
    } else {
#pragma unroll
        for (int i = 0; i < R; i++) a11[i] = a21[i] = b11[i] = b21[i] = cmake(1.0 + v, 0.5 * i);
    }
    if (!(dbg & 1)) {
        fft_wg<N, R, B, -1, DB, TWC>(a11, lds, v, c, tw, parity);
        fft_wg<N, R, B, -1, DB, TWC>(a21, lds, v, c, tw, parity);
        fft_wg<N, R, B, -1, DB, TWC>(b11, lds, v, c, tw, parity);
        fft_wg<N, R, B, -1, DB, TWC>(b21, lds, v, c, tw, parity);
    }
    const double mk = (double)(-kappa);
#pragma unroll
    for (int i = 0; i < R; i++) {
        const cplx g = io.gfac(v, i, tw);
        const cplx gb = g * b21[i];                       // g * B21
        const cplx c11 = cfma(cconj(a21[i]) * mk, gb, a11[i] * b11[i]);  // A11 B11 - k g A21* B21
        const cplx c21 = cfma(cconj(a11[i]), gb, a21[i] * b11[i]);       // A21 B11 +   g A11* B21
        b11[i] = c11;
        b21[i] = c21;
    }
    if (!(dbg & 1)) fft_wg<N, R, B, +1, DB, TWC>(b11, lds, v, c, tw, parity);
    if (!(dbg & 4)) io.store(0, b11, v, c, lds, parity);
    if (!(dbg & 1)) fft_wg<N, R, B, +1, DB, TWC>(b21, lds, v, c, tw, parity);
    if (!(dbg & 4)) io.store(1, b21, v, c, lds, parity);
    if (dbg & 4) io.sink(b11, b21);
}

// constant term ("tail") of entry e / plane s of the product of two matrices given the tails
// and, for the symmetric form, the leading coefficients of the left factor
struct TailSet {
    cplx tA[4], tB[4];   // general: 4 tails each; symmetric: [0] = 11, [1] = 21
    cplx leadA[2];       // symmetric: leading coefficients of A11, A21
};
FA_DEV cplx tail_product_general(const TailSet &t, int e)
{
    // selects, not runtime subscripts: a dynamically indexed TailSet would live in scratch memory
    const bool row = (e >> 1) != 0, col = (e & 1) != 0;
    const cplx a0 = row ? t.tA[2] : t.tA[0], a1 = row ? t.tA[3] : t.tA[1];
    const cplx b0 = col ? t.tB[1] : t.tB[0], b1 = col ? t.tB[3] : t.tB[2];
    return cfma(a1, b1, a0 * b0);
}
FA_DEV cplx tail_product_sym(const TailSet &t, int s, int kappa)
{
    // A12[d] = -kappa*conj(A21[0]),  A22[d] = conj(A11[0])
    if (s == 0) return cfma(cconj(t.leadA[1]) * (double)(-kappa), t.tB[1], t.tA[0] * t.tB[0]);
    return cfma(cconj(t.leadA[0]), t.tB[1], t.tA[1] * t.tB[0]);
}

template <int N, int R, int B, int NE> struct TreeIO {
    const TreeLevel &L;
    long long P;      // pair handled by this lane's group
    bool active;
    double sc[2];
    double m2;        // running max |coef|^2 of this lane's outputs

    FA_DEV TreeIO(const TreeLevel &L_, int c) : L(L_)
    {
        P = (long long)FA_BID * B + c;
        active = P < L.n_in / 2;
        m2 = 0.0;
        sc[0] = active ? L.scale_in[2 * P] : 0.0;
        sc[1] = active ? L.scale_in[2 * P + 1] : 0.0;
    }
    FA_DEV cplx tail(int which, int e) const
    {
        return L.tail_in[(size_t)e * L.n_in + 2 * P + which] * sc[which];
    }
    // which: 0 = left factor (A), 1 = right factor (B); e = plane
    FA_DEV void load(int which, int e, cplx (&x)[R], int v, int)
    {
        const int d = L.d;
        const cplx *src = L.body_in + (size_t)e * L.plane + (size_t)(2 * P + which) * d;
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int idx = v + (N / R) * i;
            cplx val = cmake(0.0, 0.0);
            if (active) {
                if (idx < d) val = src[idx] * sc[which];
                else if (idx == d) val = tail(which, e);
            }
            x[i] = val;
        }
    }
    FA_DEV int dbg() const { return FA_DBG(L.dbg); }
    FA_DEV void sink(cplx (&x)[R], cplx (&y)[R])
    {   // keeps ablated work alive: one lane may store
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < R; i++) s += x[i].x + y[i].y;
        if (s == 1.2345e301) L.body_out[0] = cmake(s, s);
    }
    // g[m] = exp(-2 pi i d m / N) for the bin held in register i
    FA_DEV cplx gfac(int v, int i, const cplx *tw) const
    {
        const int m = v + (N / R) * i;
        const int j = (int)(((long long)L.d * m) % N);
        return tw[j];
    }
    FA_DEV cplx tail_product(int e) const
    {
        TailSet t;
        if (NE == 4) {
#pragma unroll
            for (int q = 0; q < 4; q++) { t.tA[q] = tail(0, q); t.tB[q] = tail(1, q); }
            return tail_product_general(t, e);
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            t.tA[q] = tail(0, q);
            t.tB[q] = tail(1, q);
            t.leadA[q] = L.body_in[(size_t)q * L.plane + (size_t)(2 * P) * L.d] * sc[0];
        }
        return tail_product_sym(t, e, L.kappa);
    }
    // Results leave through LDS when a workgroup holds several pairs (B > 1): lane (v, c) owns a
    // strided sliver of pair c, and 16-byte stores at a 2d*16-byte lane stride reach HBM as
    // partial lines.  The idle transform buffer lds[parity] is the staging area (rows rotated
    // by c against bank conflicts); flipping the parity afterwards keeps the
    // one-barrier-per-exchange hand-over of fft_wg valid.
    FA_DEV void store(int e, cplx (&x)[R], int v, int c, cplx *lds, int &parity)
    {
        const int d2 = 2 * L.d;
        const int n_out = L.n_in / 2;
        const double inv = 1.0 / (double)N;
        cplx *stage = lds + ((N > R) ? (size_t)parity * (size_t)(N * B) : 0);
        cplx *dst = L.body_out + (size_t)e * L.plane + (size_t)P * d2;
        if (B > 1 && N == R) FA_SYNC_LDS();  // no transform barrier separates consecutive stores
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int idx = v + (N / R) * i;
            cplx val = x[i] * inv;
            if (active && idx == 0) {
                const cplx tp = tail_product(e);
                if (N == d2) val = val - tp;  // un-alias coefficient 2d folded onto 0
                L.tail_out[(size_t)e * n_out + P] = tp;
                m2 = fmax(m2, cnorm2(tp));
            }
            if (active && idx < d2) {
                m2 = fmax(m2, cnorm2(val));
                if (B > 1) {
                    int rot = idx + c;
                    rot = rot >= d2 ? rot - d2 * (rot / d2) : rot;
                    stage[(size_t)c * d2 + rot] = val;
                } else {
                    dst[idx] = val;
                }
            }
        }
        if (B > 1) {
            FA_SYNC_LDS();
            const long long P0 = (long long)FA_BID * B;
            const long long nvalid = (n_out - P0 < B) ? n_out - P0 : B;
            const int total = (int)nvalid * d2;
            cplx *out0 = L.body_out + (size_t)e * L.plane + (size_t)P0 * d2;
            for (int m = FA_TID; m < total; m += B * (N / R)) {
                const int c2 = m / d2, i2 = m - c2 * d2;
                int rot = i2 + c2;
                rot = rot >= d2 ? rot - d2 * (rot / d2) : rot;
                out0[m] = stage[(size_t)c2 * d2 + rot];
            }
            if (N > R) parity ^= 1;
        }
    }
};

// LDS: transform buffers (2*N*B, or N*B when !DB; N == R: one N*B staging buffer), the twiddle
// table (N entries, N > R), then B u64 slots for the per-pair maxima.
template <int N, int R, int B, bool DB, int NE> FA_DEV void body_pair_fft(const TreeLevel &L)
{
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    constexpr size_t kBuf = (N > R) ? (size_t)(DB ? 2 : 1) * N * B : (size_t)N * B;
    // larger tables would halve the residency; N = 4096 runs one workgroup per CU anyway
    constexpr bool kTwLds = (N > R) && (N <= 512 || N == 4096);
    cplx *twl = lds + kBuf;
    unsigned long long *mx = (unsigned long long *)(twl + (kTwLds ? N : 0));
    const int tid = FA_TID;
    const int c = tid % B, v = tid / B;
    if (v == 0) mx[c] = 0ull;
    const cplx *tw = L.tw;
    if (kTwLds) tw = stage_twiddles<N, B *(N / R)>(twl, L.tw);
    TreeIO<N, R, B, NE> io(L, c);
    // tables left in L2 (N = 1024, 2048) are read sparingly
    if (NE == 4) pair_product_core<N, R, B, DB, (N > R) && !kTwLds>(io, lds, tw);
    else pair_product_core_sym<N, R, B, DB, (N > R) && !kTwLds>(io, lds, tw, L.kappa);
    // every lane's first fft_wg exchange has passed a barrier after mx was zeroed when N > R;
    // for N == R (single pass, no barrier) the group is one lane, so order is trivial.
    if (N > R) {
        if (io.active) fa_atomic_max_u64(&mx[c], dbits(io.m2));
        FA_SYNC_LDS();
    } else {
        mx[c] = dbits(io.m2);
    }
    if (v == 0 && io.active) {
        const double m2 = bitsd(mx[c]);
        int a = 0;
        if (m2 > 0.0 && m2 < 1.0e300) a = half_exponent(m2);
        L.scale_out[io.P] = pow2i(-a);
        L.wexp_out[io.P] = L.wexp_in[2 * io.P] + L.wexp_in[2 * io.P + 1] + a;
    }
}

// ---------------------------------------------------------------------------------------------
// K2c: STAGES consecutive levels of the tree in one launch (symmetric form, N = 2d).  A workgroup
// owns 2^STAGES * BF consecutive matrices: stage s multiplies pairs with transforms of length
// N0*2^s, and the products stay on chip -- they cross to the lanes of the next stage through the
// transform buffer in LDS (coefficient arrays rotated by the pair index against bank conflicts).
// Only the first stage reads HBM and only the last one writes it; intermediate products are not
// rescaled (two or three products of rescaled factors cannot overflow).  T = BF * NF / R lanes,
// NF = N0 * 2^(STAGES-1).
// LDS: T*R transform/staging elements, then per stage the tails (2 per product), then BF maxima.
// ---------------------------------------------------------------------------------------------
// DB (buffering mode of the multi-level kernels): 0 one LDS transform buffer (two barriers per exchange),
// 1 two buffers used alternately (one barrier), 2 the transforms of a stage run in pairs (fft_wg2: two single LDS buffers bufX = lds,
// bufY = lds + T*R).  The products then cross to the next stage through bufX (entry 11) and bufY
// (entry 21): bufX is free once fft_wg2 returns (its last reads precede the final barrier), bufY after
// the barrier that follows the entry-11 hand-over.
template <int N, int R, int PAIRS, int DB, bool TWC> struct MultiStage {
    // symmetric pair product of the `PAIRS` pairs held by the workgroup; on entry a11.. hold the
    // factors (natural order, zero padded), on exit c11 / c21 the cyclic products times N
    static FA_DEV void product(cplx (&a11)[R], cplx (&a21)[R], cplx (&b11)[R], cplx (&b21)[R], cplx *lds, int v,
                               int c, const cplx *tw, int kappa, int &parity)
    {
        if constexpr (DB == 2) {
            fft_wg2<N, R, PAIRS, -1, TWC>(a11, a21, lds, v, c, tw);
            fft_wg2<N, R, PAIRS, -1, TWC>(b11, b21, lds, v, c, tw);
        } else {
            fft_wg<N, R, PAIRS, -1, DB == 1, TWC>(a11, lds, v, c, tw, parity);
            fft_wg<N, R, PAIRS, -1, DB == 1, TWC>(a21, lds, v, c, tw, parity);
            fft_wg<N, R, PAIRS, -1, DB == 1, TWC>(b11, lds, v, c, tw, parity);
            fft_wg<N, R, PAIRS, -1, DB == 1, TWC>(b21, lds, v, c, tw, parity);
        }
        // g[m] = exp(-2 pi i d m/N) = (-1)^m for N = 2d; m = v + (N/R) i has the parity of v (N/R even)
        const double g = ((N / R) % 2 == 0) ? ((v & 1) ? -1.0 : 1.0) : 0.0;
        const double mk = (double)(-kappa);
#pragma unroll
        for (int i = 0; i < R; i++) {
            const double gi = ((N / R) % 2 == 0) ? g : (((v + (N / R) * i) & 1) ? -1.0 : 1.0);
            const cplx gb = b21[i] * gi;
            const cplx c11 = cfma(cconj(a21[i]) * mk, gb, a11[i] * b11[i]);
            const cplx c21 = cfma(cconj(a11[i]), gb, a21[i] * b11[i]);
            b11[i] = c11;
            b21[i] = c21;
        }
        if constexpr (DB == 2) {
            fft_wg2<N, R, PAIRS, +1, TWC>(b11, b21, lds, v, c, tw);
        } else {
            fft_wg<N, R, PAIRS, +1, DB == 1, TWC>(b11, lds, v, c, tw, parity);
            fft_wg<N, R, PAIRS, +1, DB == 1, TWC>(b21, lds, v, c, tw, parity);
        }
    }
};


// tail (constant term) of the symmetric product from the factors' constant terms and the left
// factor's leading coefficients (tail_product_sym), all taken from registers: the lane with v == 0
// holds element 0 (leading coefficient) in x[0]; the constant terms come from `tA`, `tB`
FA_DEV cplx multi_tail(int e, cplx tA0, cplx tA1, cplx tB0, cplx tB1, cplx lead0, cplx lead1, int kappa)
{
    TailSet t;
    t.tA[0] = tA0; t.tA[1] = tA1; t.tB[0] = tB0; t.tB[1] = tB1; t.leadA[0] = lead0; t.leadA[1] = lead1;
    return tail_product_sym(t, e, kappa);
}

template <int N0, int STAGES, int R, int BF, int DB, int S> struct MultiRun {
    static constexpr int N = N0 << S;                      // transform length of this stage
    static constexpr int PAIRS = BF << (STAGES - 1 - S);   // pairs of this stage
    static constexpr int T = BF * (N0 << (STAGES - 1)) / R;
    static constexpr bool last = (S == STAGES - 1);

    static FA_DEV void run(const TreeLevel &L, cplx *lds, cplx *tails, unsigned long long *mx, cplx (&a11)[R],
                           cplx (&a21)[R], cplx (&b11)[R], cplx (&b21)[R], long long mat0, const cplx *const *twp,
                           int &parity, const cplx *ltail = nullptr, const int *lwexp = nullptr)
    {
        const int tid = FA_TID;
        const int c = tid % PAIRS, v = tid / PAIRS;
        constexpr int d = N / 2;                            // degree of the factors of this stage
        constexpr size_t kBufElems = (size_t)T * R;         // one transform buffer
        const int n_stage = L.n_in >> S;                    // matrices entering this stage (all signals)
        const long long pair_g = (mat0 >> (S + 1)) + c;     // global index of the product
        const bool act = pair_g < (n_stage >> 1);
        // constant terms of the two factors and leading coefficients of the left one
        cplx tA0, tA1, tB0, tB1;
        if (S == 0 && ltail != nullptr) {
            // leaf fused in front: constant terms of the (already rescaled) factors sit in LDS
            tA0 = ltail[(size_t)(2 * c) * 2]; tA1 = ltail[(size_t)(2 * c) * 2 + 1];
            tB0 = ltail[(size_t)(2 * c + 1) * 2]; tB1 = ltail[(size_t)(2 * c + 1) * 2 + 1];
        } else if (S == 0) {
            // N0 = 2d: the constant terms are element d = (N0/R)*(R/2) of the zero-padded factors, i.e.
            // register R/2 of the lanes with v == 0 -- the only lanes that use them
            tA0 = a11[R / 2]; tA1 = a21[R / 2]; tB0 = b11[R / 2]; tB1 = b21[R / 2];
        } else {
            const cplx *tp = tails + (size_t)(S - 1) * 2 * (2 * (BF << (STAGES - 1)) / 2);
            tA0 = tp[(size_t)(2 * c) * 2]; tA1 = tp[(size_t)(2 * c) * 2 + 1];
            tB0 = tp[(size_t)(2 * c + 1) * 2]; tB1 = tp[(size_t)(2 * c + 1) * 2 + 1];
        }
        const cplx lead0 = a11[0], lead1 = a21[0];          // meaningful in the lanes with v == 0
        // tables that did not fit into LDS are read sparingly (three entries per butterfly)
        MultiStage<N, R, PAIRS, DB, (N0 * ((1 << STAGES) - 1) > 1024)>::product(a11, a21, b11, b21, lds, v, c, twp[S],
                                                                                L.kappa, parity);
        const double inv = 1.0 / (double)N;
        cplx tp0 = cmake(0.0, 0.0), tp1 = tp0;
        if (v == 0) {
            tp0 = multi_tail(0, tA0, tA1, tB0, tB1, lead0, lead1, L.kappa);
            tp1 = multi_tail(1, tA0, tA1, tB0, tB1, lead0, lead1, L.kappa);
        }
        if constexpr (!last) {
            // hand the products over to the lanes of the next stage: X[pair][(idx + pair) mod N]
            cplx *tnext = tails + (size_t)S * 2 * (BF << (STAGES - 1));
            constexpr int PN = PAIRS / 2, NN = 2 * N;          // next stage: pairs, length
            const int c2 = tid % PN, v2 = tid / PN;
            cplx na[R], nb[R];
            for (int e = 0; e < 2; e++) {
                // single buffer: wait until the last exchange has been read; double buffer: the idle
                // one is free by the hand-over rule of fft_wg
                cplx *stg = (DB == 2) ? lds + (size_t)e * kBufElems : ((DB == 1) ? lds + (size_t)parity * kBufElems : lds);
                if (DB == 0) FA_SYNC_LDS();
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const int idx = v + (N / R) * i;
                    cplx val = (e == 0 ? b11[i] : b21[i]) * inv;
                    if (idx == 0) val = val - (e == 0 ? tp0 : tp1);   // un-alias coefficient 2d folded onto 0
                    const int rot = (idx + c) & (N - 1);
                    stg[(size_t)c * N + rot] = val;
                }
                if (v == 0) tnext[(size_t)c * 2 + e] = (e == 0 ? tp0 : tp1);
                FA_SYNC_LDS();
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const int idx = v2 + (NN / R) * i;         // element of the next stage's factor
                    cplx xa = cmake(0.0, 0.0), xb = xa;
                    if (idx < N) {
                        const int ra = (idx + 2 * c2) & (N - 1), rb = (idx + 2 * c2 + 1) & (N - 1);
                        xa = stg[(size_t)(2 * c2) * N + ra];
                        xb = stg[(size_t)(2 * c2 + 1) * N + rb];
                    } else if (idx == N) {
                        xa = tnext[(size_t)(2 * c2) * 2 + e];
                        xb = tnext[(size_t)(2 * c2 + 1) * 2 + e];
                    }
                    na[i] = xa; nb[i] = xb;
                }
                if (e == 0) {
#pragma unroll
                    for (int i = 0; i < R; i++) { a11[i] = na[i]; b11[i] = nb[i]; }
                } else {
#pragma unroll
                    for (int i = 0; i < R; i++) { a21[i] = na[i]; b21[i] = nb[i]; }
                }
                if (DB == 1) parity ^= 1;
            }
            MultiRun<N0, STAGES, R, BF, DB, S + 1>::run(L, lds, tails, mx, a11, a21, b11, b21, mat0, twp, parity, ltail,
                                                        lwexp);
        } else {
            // ---- last stage: maxima, pending scale, coalesced stores through LDS ------------------
            const int n_out = n_stage >> 1;
            double m2 = 0.0;
            for (int e = 0; e < 2; e++) {
                cplx *stg = (DB == 2) ? lds + (size_t)e * kBufElems : ((DB == 1) ? lds + (size_t)parity * kBufElems : lds);
                if (DB == 0) FA_SYNC_LDS();
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const int idx = v + (N / R) * i;
                    cplx val = (e == 0 ? b11[i] : b21[i]) * inv;
                    if (idx == 0) {
                        const cplx tp = (e == 0 ? tp0 : tp1);
                        val = val - tp;
                        if (act) {
                            L.tail_out[(size_t)e * n_out + pair_g] = tp;
                            m2 = fmax(m2, cnorm2(tp));
                        }
                    }
                    if (act) m2 = fmax(m2, cnorm2(val));
                    const int rot = (idx + c) & (N - 1);
                    stg[(size_t)c * N + rot] = val;
                }
                FA_SYNC_LDS();
                const long long Pg0 = mat0 >> (S + 1);
                const long long nvalid = (n_out - Pg0 < PAIRS) ? n_out - Pg0 : PAIRS;
                const int total = (int)nvalid * N;
                cplx *out0 = L.body_out + (size_t)e * L.plane + (size_t)Pg0 * N;
                for (int m = tid; m < total; m += T) {
                    const int c3 = m / N, i3 = m - c3 * N;
                    const int rot = (i3 + c3) & (N - 1);
                    out0[m] = stg[(size_t)c3 * N + rot];
                }
                if (DB == 1) parity ^= 1;
            }
            if (act) fa_atomic_max_u64(&mx[c], dbits(m2));
            FA_SYNC_LDS();
            if (v == 0 && act) {
                const double mm = bitsd(mx[c]);
                int a = 0;
                if (mm > 0.0 && mm < 1.0e300) a = half_exponent(mm);
                L.scale_out[pair_g] = pow2i(-a);
                int w = a;
                const long long m_first = pair_g << STAGES;      // input matrices of this product
                for (int j = 0; j < (1 << STAGES); j++)
                    w += lwexp ? lwexp[(m_first - mat0) + j] : L.wexp_in[m_first + j];
                L.wexp_out[pair_g] = w;
            }
        }
    }
};
template <int N0, int STAGES, int R, int BF, int DB, int S>
FA_DEV void multi_stage_run(const TreeLevel &L, cplx *lds, cplx *tails, unsigned long long *mx, cplx (&a11)[R],
                            cplx (&a21)[R], cplx (&b11)[R], cplx (&b21)[R], long long mat0, const cplx *const *twp)
{
    int parity = 0;
    MultiRun<N0, STAGES, R, BF, DB, S>::run(L, lds, tails, mx, a11, a21, b11, b21, mat0, twp, parity);
}

template <int N0, int STAGES, int R, int BF, int DB> FA_DEV void body_multi_fft(const TreeLevel &L)
{
    constexpr int NF = N0 << (STAGES - 1);
    constexpr int T = BF * NF / R;
    constexpr int P0 = BF << (STAGES - 1);          // pairs of stage 0
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    cplx *tails = lds + (size_t)((DB != 0) ? 2 : 1) * T * R;   // [stage][pair][2]
    unsigned long long *mx = (unsigned long long *)(tails + (size_t)2 * 2 * P0);
    const int tid = FA_TID;
    const long long blk = FA_BID;
    const int n_in = L.n_in;
    const long long mat0 = blk * (2 * P0);          // first input matrix of the workgroup
    if (tid < BF) mx[tid] = 0ull;
    // twiddle tables of all stages in LDS when they are small (N0 + 2 N0 + .. <= 1024 entries)
    constexpr int kTwTotal = N0 * ((1 << STAGES) - 1);
    constexpr bool kTwLds = kTwTotal <= 1024;
    const cplx *twp[3] = {L.twm[0], L.twm[1], L.twm[2]};
    if (kTwLds) {
        cplx *twl = (cplx *)(mx + ((BF + 1) & ~1));   // 16-byte aligned after the maxima
        int off = 0;
        for (int s = 0; s < STAGES; s++) {
            const int len = N0 << s;
            for (int j = tid; j < len; j += T) twl[off + j] = L.twm[s][j];
            twp[s] = twl + off;
            off += len;
        }
        FA_SYNC_LDS();
    }
    cplx a11[R], a21[R], b11[R], b21[R];
    // ---- stage 0: factors from HBM -----------------------------------------------------------
    // N0 = 2d: registers i < R/2 hold coefficients, register R/2 of the lanes with v == 0 the constant
    // term, everything else is zero padding.  All loads are unconditional (clamped for the inactive
    // tail of the grid) and issued back to back; scales and masks are applied afterwards.
    {
        const int c = tid % P0, v = tid / P0;
        const long long mA = mat0 + 2 * c, mB = mA + 1;
        const bool act = mB < n_in;
        const long long mAc = act ? mA : 0, mBc = act ? mB : 0;
        const size_t d = (size_t)L.d;
        const cplx *pA0 = L.body_in + (size_t)mAc * d, *pA1 = pA0 + L.plane;
        const cplx *pB0 = L.body_in + (size_t)mBc * d, *pB1 = pB0 + L.plane;
#pragma unroll
        for (int i = 0; i < R / 2; i++) {
            const int idx = v + (N0 / R) * i;
            a11[i] = pA0[idx]; a21[i] = pA1[idx]; b11[i] = pB0[idx]; b21[i] = pB1[idx];
        }
        const cplx tA0 = L.tail_in[mAc], tA1 = L.tail_in[(size_t)n_in + mAc];
        const cplx tB0 = L.tail_in[mBc], tB1 = L.tail_in[(size_t)n_in + mBc];
        const double sA = act ? L.scale_in[mAc] : 0.0, sB = act ? L.scale_in[mBc] : 0.0;
#pragma unroll
        for (int i = 0; i < R / 2; i++) {
            a11[i] = a11[i] * sA; a21[i] = a21[i] * sA; b11[i] = b11[i] * sB; b21[i] = b21[i] * sB;
        }
        const double tA = (v == 0) ? sA : 0.0, tB = (v == 0) ? sB : 0.0;
        a11[R / 2] = tA0 * tA; a21[R / 2] = tA1 * tA; b11[R / 2] = tB0 * tB; b21[R / 2] = tB1 * tB;
#pragma unroll
        for (int i = R / 2 + 1; i < R; i++) a11[i] = a21[i] = b11[i] = b21[i] = cmake(0.0, 0.0);
    }
    // ---- stages ---------------------------------------------------------------------------------
    multi_stage_run<N0, STAGES, R, BF, DB, 0>(L, lds, tails, mx, a11, a21, b11, b21, mat0, twp);
}

// Leaf kernel and the first multi-level launch in one: lane t of the workgroup forms matrix t of the
// block from SPT samples in registers (body_leaf), the 2*P0 = T matrices cross to the stage-0 lanes
// (pair t/2, element v + 2 i) through the transform buffer, and STAGES levels follow on chip.
struct LeafMultiParams {
    LeafParams lp;
    TreeLevel L;      // n_in = matrices the leaf produces, d = DEG*SPT; *_in are unused
};
template <int DEG, int SPT, int STAGES, int R, int BF, int DB> FA_DEV void body_leaf_multi(const LeafMultiParams &Q)
{
    constexpr int d = DEG * SPT;
    constexpr int N0 = 2 * d;
    static_assert(N0 / R == 2, "one lane per matrix: N0/R must be 2");
    constexpr int NF = N0 << (STAGES - 1);
    constexpr int T = BF * NF / R;
    constexpr int P0 = BF << (STAGES - 1);
    static_assert(T == 2 * P0, "lanes = matrices of the block");
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    cplx *tails = lds + (size_t)((DB != 0) ? 2 : 1) * T * R;
    unsigned long long *mx = (unsigned long long *)(tails + (size_t)2 * 2 * P0);
    cplx *twl = (cplx *)(mx + ((BF + 1) & ~1));
    constexpr int kTwTotal = N0 * ((1 << STAGES) - 1);
    cplx *ltail = twl + kTwTotal;                   // [T][2]
    int *lwexp = (int *)(ltail + (size_t)2 * T);   // [T]
    const CoeffParams &P = Q.lp.c;
    const TreeLevel &L = Q.L;
    const int tid = FA_TID;
    const long long mat0 = (long long)FA_BID * T;
    if (tid < BF) mx[tid] = 0ull;
    const cplx *twp[3] = {L.twm[0], L.twm[1], L.twm[2]};
    {
        int off = 0;
        for (int s = 0; s < STAGES; s++) {
            const int len = N0 << s;
            for (int j = tid; j < len; j += T) twl[off + j] = L.twm[s][j];
            twp[s] = twl + off;
            off += len;
        }
    }
    // ---- leaf: matrix mat0 + tid ----------------------------------------------------------------
    const long long gid = mat0 + tid;
    const long long per = P.Dpad / SPT;
    const long long n_out = (long long)P.batch * per;
    const bool active = gid < n_out;
    cplx acc[4][d + 1];
    double sc = 0.0;
    int aexp = 0;
    if (active) {
        const int b = (int)(gid / per);
        const long long j0 = (gid % per) * SPT;
        bool ok = true;
        {
            CoefMat<DEG> m;
            ok = leaf_sample<DEG>(P, b, j0, m);
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int k2 = 0; k2 <= DEG; k2++) acc[e][k2] = m.p[e][k2];
        }
        LeafLoop<DEG, SPT, 1>::run(P, b, j0, acc, ok);
        if (!ok) fa_atomic_or_i32(P.status, 1);
        double m2 = 0.0;
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int k2 = 0; k2 <= d; k2++) m2 = fmax(m2, cnorm2(acc[e][k2]));
        sc = 1.0;
        if (m2 > 0.0 && m2 < 1.0e300) {
            aexp = half_exponent(m2);
            sc = pow2i(-aexp);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int k2 = 0; k2 <= d; k2++) acc[e][k2] = cmake(0.0, 0.0);
    }
    lwexp[tid] = aexp;
    ltail[(size_t)tid * 2] = acc[0][d] * sc;
    ltail[(size_t)tid * 2 + 1] = acc[2][d] * sc;
    // ---- hand-over: bodies of entry 11, then 21, through the transform buffer ---------------------
    cplx a11[R], a21[R], b11[R], b21[R];
    const int c = tid % P0, v = tid / P0;     // stage-0 lane: pair c, elements v + 2 i
    for (int e = 0; e < 2; e++) {
        FA_SYNC_LDS();
#pragma unroll
        for (int k2 = 0; k2 < d; k2++) lds[(size_t)tid * d + ((k2 + tid) & (d - 1))] = acc[2 * e][k2] * sc;
        FA_SYNC_LDS();
        cplx xa[R], xb[R];
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int idx = v + 2 * i;
            cplx ya = cmake(0.0, 0.0), yb = ya;
            if (idx < d) {
                ya = lds[(size_t)(2 * c) * d + ((idx + 2 * c) & (d - 1))];
                yb = lds[(size_t)(2 * c + 1) * d + ((idx + 2 * c + 1) & (d - 1))];
            } else if (idx == d) {
                ya = ltail[(size_t)(2 * c) * 2 + e];
                yb = ltail[(size_t)(2 * c + 1) * 2 + e];
            }
            xa[i] = ya; xb[i] = yb;
        }
        if (e == 0) {
#pragma unroll
            for (int i = 0; i < R; i++) { a11[i] = xa[i]; b11[i] = xb[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < R; i++) { a21[i] = xa[i]; b21[i] = xb[i]; }
        }
    }
    FA_SYNC_LDS();   // staging fully read before the first transform writes the buffer (DB: buffer 0)
    int parity = 0;
    MultiRun<N0, STAGES, R, BF, DB, 0>::run(L, lds, tails, mx, a11, a21, b11, b21, mat0, twp, parity, ltail, lwexp);
}

// ---------------------------------------------------------------------------------------------
// Large transforms: N = N1 * N2 with element index n = n1*N2 + n2 and bin k = k1 + N1*k2.
//   forward : column DFT over n1 (size N1), twiddle w_N^{n2 k1}, row DFT over n2 (size N2)
//   inverse : row IDFT over k2, twiddle conj, column IDFT over k1
// Y/Z scratch layout: [poly][k1][n2], poly = e*n_mats + mat.
// ---------------------------------------------------------------------------------------------
struct BigLevel {
    TreeLevel L;
    cplx *Y;         // forward scratch: 4*n_in polys of N
    cplx *Z;         // inverse scratch: 4*(n_in/2) polys of N
    BigTwiddle btw;  // exp(-2 pi i j/N)
    const cplx *tw1; // table for N1
    const cplx *tw2; // table for N2
    int N1, N2;
    // bridge to the next level (body_col_bridge): tables for 2*N1 and 2*N, Y holds unscaled data
    const cplx *tw1x2;
    BigTwiddle btw2;
    int y_unscaled;  // row kernel multiplies by scale_in on load
    // spectral doubling (body_col_bridge2): Y holds only the ODD rows k1 = 2j+1 of the next
    // level's column transform ([poly][j][n2]); the EVEN rows k1 = 2j are the previous level's Z
    // rows j, read in place from Zprev
    const cplx *Zprev;
    int y_split;
    // first split level with N1 = 4 and N = 2d: no column kernel -- the row kernel forms the length-4
    // column transform itself, Y[k1][n2] = x[n2] + i^(-k1) x[N2 + n2] (+ (-1)^k1 tail at n2 = 0)
    int y_direct;
    // real-coefficient path (nft_real.h): btw is the table of length 4*N and the row twiddle of row k1 is
    // w_{4N}^{(4 k1 - 1) n2} (transform of the twisted sequence: evaluation at the roots of x^N = i);
    // twq / twq2: exp(-2 pi i j/(4 N1)) and exp(-2 pi i j/(8 N1)) for the column kernels
    int rtwist = 0;
    const cplx *twq = nullptr, *twq2 = nullptr;
    // column length N1 = 3*K (body_r3*): tables exp(-2 pi i j/N1) and exp(-2 pi i j/(2 N1)); tw1 / tw1x2 then belong to K, 2K
    const cplx *tw3 = nullptr, *tw3x2 = nullptr;
    unsigned long long row_mod = 0;   // rtwist: 4*N1*N2 when that is not a power of two (row_tw_index reduces modulo it)
    int stagger;     // row kernel: start delay of the second half of the grid, units of ~1024 clocks (0: none)
    // diagnostic builds (-DFNFT_AMD_STAMPS) only: per-wave s_memtime stamps of the row kernel's phases,
    // 16 slots per wave, or NULL
    unsigned long long *stamps;
};

#ifdef FNFT_AMD_STAMPS
#define FA_STAMP(ptr, slot)                                                                              \
    do {                                                                                                 \
        if (ptr) {                                                                                       \
            const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();   /* 100 MHz, chip-wide */                                  \
            if ((FA_TID & 63) == 0) (ptr)[((size_t)FA_BID * (FA_BDIM / 64) + FA_TID / 64) * 16 + (slot)] = t_; \
        }                                                                                                \
    } while (0)
#else
#define FA_STAMP(ptr, slot) do { } while (0)
#endif

// Element (row, n2) of one polynomial's Y/Z scratch block of `rows` x N2 elements.  The block is stored
// in tiles of FA_YZ_TILE consecutive n2: [n2 tile][row][n2 in tile].  A column kernel that owns BC
// consecutive n2 for every row then streams a contiguous region (rows x 512 B per tile) instead of
// touching `rows` places 16*N2 bytes apart, and a row kernel still moves whole 512 B pieces per half-wave
// (a wave instruction covers 64 consecutive n2 = two tiles).  FA_YZ_TILE = 0: plain [row][n2].  Tile width measured
// at cfg 2 (ms per step, three runs each): 128: 0.704, 64: 0.704-0.714, 32: 0.696-0.698, 16: 0.699, 8: 0.700 -- the
// column kernels own 8 or 16 consecutive n2, so narrower tiles put more of a tile into one workgroup.
#ifndef FA_YZ_TILE
#define FA_YZ_TILE 32
#endif
FA_HD size_t yz_index(int rows, int N2, int row, int n2)
{
#if FA_YZ_TILE > 0
    (void)N2;
    return ((size_t)(n2 / FA_YZ_TILE) * (size_t)rows + (size_t)row) * FA_YZ_TILE + (size_t)(n2 % FA_YZ_TILE);
#else
    (void)rows;
    return (size_t)row * (size_t)N2 + (size_t)n2;
#endif
}

// column step of the forward transform of every input polynomial of the level
//   grid.x = N2/BC tiles, grid.y = 4*n_in polynomials
template <int N1, int R, int BC, bool DB> FA_DEV void body_col_fwd(const BigLevel &G)
{
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    const TreeLevel &L = G.L;
    const int tid = FA_TID;
    const int c = tid % BC, v = tid / BC;
    const int n2 = FA_BID * BC + c;
    const int poly = FA_BID_Y;
    const int e = poly / L.n_in, mat = poly % L.n_in;
    const int d = L.d, N2 = G.N2;
    const double sc = level_in_scale(L, mat);
    const cplx *src = L.body_in + (size_t)e * L.plane + (size_t)mat * d;
    cplx x[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int n1 = v + (N1 / R) * i;
        const long long idx = (long long)n1 * N2 + n2;
        cplx val = cmake(0.0, 0.0);
        if (idx < d) val = src[idx] * sc;
        else if (idx == d) val = L.tail_in[(size_t)e * L.n_in + mat] * sc;
        x[i] = val;
    }
    int parity = 0;
    fft_wg<N1, R, BC, -1, DB, true>(x, lds, v, c, G.tw1, parity);
    // the twiddle w_N^{n2 k1} between column and row step is applied by the row kernel, where
    // one set of R factors per lane serves every polynomial of the pair
    cplx *dst = G.Y + (size_t)poly * N1 * N2;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int k1 = v + (N1 / R) * i;
        dst[yz_index(N1, N2, k1, n2)] = x[i];
    }
}

// row step: one workgroup per (pair, k1): forward row transforms of every stored entry of both
// factors, products, inverse row transforms (8 + 4 general, 4 + 2 symmetric)
// index into the row-twiddle table of a split level: w_N^{k1 x}, or -- real-coefficient path (BigLevel::rtwist, table of
// length 4N) -- w_{4N}^{(4 k1 - 1) x}
FA_DEV unsigned row_tw_index(const BigLevel &G, int k1, int x)
{
    if (!G.rtwist) return (unsigned)k1 * (unsigned)x;
    const unsigned long long len4 = 4ull * (unsigned long long)G.N1 * (unsigned long long)G.N2;
    const unsigned long long c1 = 4ull * (unsigned long long)k1 + len4 - 1ull;                     // 4 k1 - 1 (mod 4N)
    if (G.row_mod) return (unsigned)((c1 * (unsigned long long)x) % G.row_mod);                    // N1 = 3*2^j
    return (unsigned)((c1 * (unsigned long long)x) & (len4 - 1ull));                               // a power of two
}

template <int N2, int R> struct MidIO {
    const BigLevel &G;
    long long P;
    int k1;
    double sc[2];
    cplx wu[R];   // workgroup-uniform factors w^{k1 (N2/R) i}: scalar registers
    cplx wbase;   // per-lane w^{k1 v}
    FA_DEV MidIO(const BigLevel &G_) : G(G_)
    {
        P = FA_BID / G.N1;
        k1 = FA_BID % G.N1;
#pragma unroll
        for (int i = 1; i < R; i++) {
            const cplx w = big_twiddle(G.btw, row_tw_index(G, k1, (N2 / R) * i));
            wu[i] = cmake(fa_uniform(w.x), fa_uniform(w.y));
        }
        wu[0] = cmake(1.0, 0.0);
        wbase = big_twiddle(G.btw, row_tw_index(G, k1, FA_TID));
        sc[0] = (G.y_unscaled || G.y_direct) ? level_in_scale(G.L, 2 * P) : 1.0;
        sc[1] = (G.y_unscaled || G.y_direct) ? level_in_scale(G.L, 2 * P + 1) : 1.0;
        // bookkeeping of the level, done once per pair before the column kernel that follows:
        // exponent carried so far (this level's own is added by the consumer / the final
        // finalize) and a clean slot for this level's maximum
        const int wsum = level_in_wexp(G.L, 2 * P) + level_in_wexp(G.L, 2 * P + 1);   // wave-uniform
        if (k1 == 0 && FA_TID == 0) G.L.wexp_out[P] = wsum;
        if (k1 == 0 && FA_TID < kMax2Slots) G.L.max2_out[(size_t)P * kMax2Slots + FA_TID] = 0u;
    }
    // w_N^{k1 n2} for element n2 = v + (N2/R)*i of this lane (same for every polynomial) =
    // per-lane look-up w^{k1 v} times the workgroup-uniform factor w^{k1 (N2/R) i}; formed per
    // element (no register array: the row kernel is at its VGPR budget)
    FA_DEV cplx twiddle(cplx base, int i) const
    {
        return (i == 0) ? base : base * wu[i];
    }
    FA_DEV void load(int which, int e, cplx (&x)[R], int v, int)
    {
        const int n_in = G.L.n_in;
        const size_t pi = (size_t)e * n_in + 2 * P + which;
        if (G.y_direct) {
            // w_4^{k1} = (-i)^{k1}; element n1 = 2 exists only as the constant term (index 2*N2 = d)
            const size_t mat = (size_t)(2 * P + which);
            const cplx *b0 = G.L.body_in + (size_t)e * G.L.plane + mat * (size_t)G.L.d;
            const cplx tl = G.L.tail_in[(size_t)e * n_in + mat];
            const cplx base = wbase * sc[which];
#pragma unroll
            for (int i = 0; i < R; i++) {
                const int n2 = v + (N2 / R) * i;
                const cplx x0 = b0[n2], x1 = b0[N2 + n2];
                cplx r1;   // x1 * (-i)^k1
                switch (k1 & 3) {
                case 0: r1 = x1; break;
                case 1: r1 = cmake(x1.y, -x1.x); break;
                case 2: r1 = cmake(-x1.x, -x1.y); break;
                default: r1 = cmake(-x1.y, x1.x); break;
                }
                cplx y = x0 + r1;
                if (n2 == 0) y = (k1 & 1) ? y - tl : y + tl;
                x[i] = y * twiddle(base, i);
            }
            return;
        }
        const cplx *src;
        int rows, row;
        if (G.y_split) {
            const cplx *half = (k1 & 1) ? G.Y : G.Zprev;
            rows = G.N1 / 2;
            row = k1 >> 1;
            src = half + pi * (size_t)rows * N2;
        } else {
            rows = G.N1;
            row = k1;
            src = G.Y + pi * (size_t)rows * N2;
        }
        const cplx base = wbase * sc[which];
#pragma unroll
        for (int i = 0; i < R; i++) x[i] = src[yz_index(rows, N2, row, v + (N2 / R) * i)] * twiddle(base, i);
    }
    FA_DEV int dbg() const { return FA_DBG(G.L.dbg); }
    FA_DEV void sink(cplx (&x)[R], cplx (&y)[R])
    {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < R; i++) s += x[i].x + y[i].y;
        if (s == 1.2345e301) G.Z[0] = cmake(s, s);
    }
    // g = exp(-2 pi i d k / N) at bin k = k1 + N1*k2, k2 = v + (N2/R)*i
    FA_DEV cplx gfac(int v, int i, const cplx *) const
    {
        const long long N = (long long)G.N1 * N2;
        const long long kbin = (long long)k1 + (long long)G.N1 * (v + (N2 / R) * i);
        const long long j = ((long long)G.L.d * kbin) % N;
        return big_twiddle(G.btw, (unsigned)j);
    }
    FA_DEV void store(int e, cplx (&x)[R], int v, int, cplx *, int &)
    {
        const int n_out = G.L.n_in / 2;
        cplx *dst = G.Z + (size_t)((size_t)e * n_out + P) * G.N1 * N2;
        const cplx base = wbase * (1.0 / (double)N2);
#pragma unroll
        for (int i = 0; i < R; i++) dst[yz_index(G.N1, N2, k1, v + (N2 / R) * i)] = x[i] * cconj(twiddle(base, i));
    }
};

template <int N2, int R, int NE> FA_DEV void body_mid(const BigLevel &G)
{
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    MidIO<N2, R> io(G);
    const cplx *tw = G.tw2;
    // the 2048-entry table stays in L2: three reads per butterfly, other powers by multiplication
    if (NE == 4) pair_product_core<N2, R, 1, true, true>(io, lds, tw);
    else pair_product_core_sym<N2, R, 1, true, true>(io, lds, tw, G.L.kappa);
}

// loads of the right factor's two rows, FA_MID_BSTEP elements per barrier interval of the left factor's
// transforms (fft_wg2 hook)
#ifndef FA_MID_BSTEP
#define FA_MID_BSTEP 2
#endif
template <int N2, int R> struct MidBLoader {
    const cplx *p0, *p1;
    int rows, row, v;
    cplx (&b11)[R];
    cplx (&b21)[R];
    FA_DEV void operator()(int k) const
    {
#pragma unroll
        for (int i = 0; i < R; i++)
            if (i / FA_MID_BSTEP == k) {
                const size_t o = yz_index(rows, N2, row, v + (N2 / R) * i);
                b11[i] = p0[o];
                b21[i] = p1[o];
            }
    }
};

// Row step of the symmetric form, written for memory-level parallelism: the row pointers of the four
// input polynomials are formed without branches, all 4*R (DIRECT: 2 x 4*R) 16-byte loads of the lane
// are issued back to back before anything consumes them (the generic IO object above branches per
// polynomial on the level's flags, which made every polynomial its own memory round trip), the
// transforms run in pairs (fft_wg2: a11 with a21, b11 with b21, c11 with c21), and every barrier is
// an LDS-only one, so the stores of c11 stay in flight under the tail of the c21 transform.
//   DIRECT: first split level with N1 = 4 and N = 2d -- the row kernel forms the length-4 column
//   transform itself, Y[k1][n2] = x[n2] + (-i)^k1 x[N2 + n2]  (+ (-1)^k1 * tail at n2 = 0).
template <int N2, int R, bool DIRECT> FA_DEV void body_mid_sym(const BigLevel &G)
{
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    const TreeLevel &L = G.L;
    const int v = FA_TID;
    const int n_in = L.n_in, n_out = n_in / 2;
    long long P;
    int k1;
    if (DIRECT && (n_out % 8) == 0) {
        // the four k1 of a pair read the same two coefficient rows: keep them on one XCD (workgroups
        // b and b + 8 share one), so the rows come out of that XCD's L2 three times out of four
        const int q = FA_BID / 8, x = FA_BID % 8;
        k1 = q % 4;
        P = (long long)(q / 4) * 8 + x;
    } else {
        P = FA_BID / G.N1;
        k1 = FA_BID % G.N1;
    }
    const long long mA = 2 * P, mB = 2 * P + 1;
    cplx a11[R], a21[R], b11[R], b21[R];
    MidBLoader<N2, R> bld{nullptr, nullptr, 0, 0, 0, b11, b21};
    // One level is one round of resident workgroups (two per CU), which all start together and then
    // sit in the same phase -- everybody loading (HBM saturated, vector units idle), then everybody
    // transforming (vector units saturated, HBM idle).  Holding back the second half of the grid -- the
    // second workgroup of every CU -- by about one load phase puts the two halves in complementary phases.
    fa_stagger(G.stagger, FA_BID >= FA_GDIM / 2);
    FA_STAMP(G.stamps, 0);
    // ---- loads ------------------------------------------------------------------------------------
    if constexpr (DIRECT) {
        const size_t d = (size_t)L.d;
        const cplx *pA0 = L.body_in + (size_t)mA * d, *pA1 = L.body_in + L.plane + (size_t)mA * d;
        const cplx *pB0 = L.body_in + (size_t)mB * d, *pB1 = L.body_in + L.plane + (size_t)mB * d;
        cplx h0[R], h1[R];
#pragma unroll
        for (int i = 0; i < R; i++) { a11[i] = pA0[v + (N2 / R) * i]; a21[i] = pA1[v + (N2 / R) * i]; }
#pragma unroll
        for (int i = 0; i < R; i++) { h0[i] = pA0[N2 + v + (N2 / R) * i]; h1[i] = pA1[N2 + v + (N2 / R) * i]; }
#pragma unroll
        for (int i = 0; i < R; i++) { b11[i] = pB0[v + (N2 / R) * i]; b21[i] = pB1[v + (N2 / R) * i]; }
        const cplx tA0 = L.tail_in[mA], tA1 = L.tail_in[(size_t)n_in + mA];
        const cplx tB0 = L.tail_in[mB], tB1 = L.tail_in[(size_t)n_in + mB];
        const int kq = k1 & 3;
        auto rot = [kq](cplx z) -> cplx {   // z * (-i)^k1
            return kq == 0 ? z : (kq == 1 ? cmake(z.y, -z.x) : (kq == 2 ? cmake(-z.x, -z.y) : cmake(-z.y, z.x)));
        };
        const double sg = (k1 & 1) ? -1.0 : 1.0;
#pragma unroll
        for (int i = 0; i < R; i++) { a11[i] = a11[i] + rot(h0[i]); a21[i] = a21[i] + rot(h1[i]); }
#pragma unroll
        for (int i = 0; i < R; i++) { h0[i] = pB0[N2 + v + (N2 / R) * i]; h1[i] = pB1[N2 + v + (N2 / R) * i]; }
#pragma unroll
        for (int i = 0; i < R; i++) { b11[i] = b11[i] + rot(h0[i]); b21[i] = b21[i] + rot(h1[i]); }
        if (v == 0) {   // n2 = 0 is element 0 of lane 0
            a11[0] = a11[0] + tA0 * sg; a21[0] = a21[0] + tA1 * sg;
            b11[0] = b11[0] + tB0 * sg; b21[0] = b21[0] + tB1 * sg;
        }
    } else {
        const bool split = G.y_split != 0;
        const int rows = split ? G.N1 / 2 : G.N1;
        const int row = split ? (k1 >> 1) : k1;
        const cplx *base = split ? ((k1 & 1) ? G.Y : G.Zprev) : G.Y;
        const size_t blk = (size_t)rows * N2;
        const cplx *pA0 = base + (size_t)mA * blk, *pA1 = base + ((size_t)n_in + mA) * blk;
        const cplx *pB0 = base + (size_t)mB * blk, *pB1 = base + ((size_t)n_in + mB) * blk;
#pragma unroll
        for (int i = 0; i < R; i++) {
            const size_t o = yz_index(rows, N2, row, v + (N2 / R) * i);
            a11[i] = pA0[o]; a21[i] = pA1[o];
        }
        // the right factor's rows are requested from inside the left factor's transforms (MidBLoader):
        // a wave cannot run ahead of a load it has not been able to issue yet, and with all 4*R requests of
        // every wave queued at once the first transform starts only when the whole level has been fetched
        bld.p0 = pB0; bld.p1 = pB1; bld.rows = rows; bld.row = row; bld.v = v;
    }
    FA_STAMP(G.stamps, 1);
    // ---- bookkeeping of the level (once per pair) and the factors every polynomial shares ------------
    const int pendA = level_in_pending_exp(L, mA), pendB = level_in_pending_exp(L, mB);   // workgroup-uniform
    if (k1 == 0 && v == 0) L.wexp_out[P] = L.wexp_in[mA] + pendA + L.wexp_in[mB] + pendB;
    if (k1 == 0 && v < kMax2Slots) L.max2_out[(size_t)P * kMax2Slots + v] = 0u;
    const bool rescale = DIRECT || G.y_unscaled;
    const double scA = !rescale ? 1.0 : (L.in_pending ? pow2i(-pendA) : L.scale_in[mA]);
    const double scB = !rescale ? 1.0 : (L.in_pending ? pow2i(-pendB) : L.scale_in[mB]);
    // w_N^{k1 n2}, n2 = v + (N2/R) i: per-lane w^{k1 v} times workgroup-uniform w^{k1 (N2/R) i}
    const cplx wbase = big_twiddle(G.btw, (unsigned)k1 * (unsigned)v);
    cplx wu[R];
    wu[0] = cmake(1.0, 0.0);
#pragma unroll
    for (int i = 1; i < R; i++) {
        const cplx w = big_twiddle(G.btw, (unsigned)k1 * (unsigned)((N2 / R) * i));
        wu[i] = cmake(fa_uniform(w.x), fa_uniform(w.y));
    }
    // the left factor's rows are the first half of the loads: its transforms start while the right
    // factor's rows are still arriving
#pragma unroll
    for (int i = 0; i < R; i++) {
        const cplx wA = ((i == 0) ? wbase : wbase * wu[i]) * scA;
        a11[i] = a11[i] * wA; a21[i] = a21[i] * wA;
    }
    // ---- transforms and product -----------------------------------------------------------------------
    const cplx *tw = G.tw2;
    FA_STAMP(G.stamps, 2);
    if constexpr (DIRECT) fft_wg2<N2, R, 1, -1, true>(a11, a21, lds, v, 0, tw);
    else fft_wg2<N2, R, 1, -1, true>(a11, a21, lds, v, 0, tw, bld);
    FA_STAMP(G.stamps, 3);
#pragma unroll
    for (int i = 0; i < R; i++) {
        const cplx wB = ((i == 0) ? wbase : wbase * wu[i]) * scB;
        b11[i] = b11[i] * wB; b21[i] = b21[i] * wB;
    }
    fft_wg2<N2, R, 1, -1, true>(b11, b21, lds, v, 0, tw);
    FA_STAMP(G.stamps, 4);
    const double mk = (double)(-L.kappa);
    const long long N = (long long)G.N1 * N2;
    if (N == 2 * (long long)L.d) {
        // g = exp(-2 pi i d k/N) = (-1)^k, k = k1 + N1*k2: the parity of k1 (N1 is even)
        const double g = (k1 & 1) ? -1.0 : 1.0;
#pragma unroll
        for (int i = 0; i < R; i++) {
            const cplx gb = b21[i] * g;
            const cplx c11 = cfma(cconj(a21[i]) * mk, gb, a11[i] * b11[i]);   // A11 B11 - k g A21* B21
            const cplx c21 = cfma(cconj(a11[i]), gb, a21[i] * b11[i]);        // A21 B11 +   g A11* B21
            b11[i] = c11;
            b21[i] = c21;
        }
    } else {
#pragma unroll
        for (int i = 0; i < R; i++) {
            const long long kbin = (long long)k1 + (long long)G.N1 * (v + (N2 / R) * i);
            const cplx g = big_twiddle(G.btw, (unsigned)(((long long)L.d * kbin) % N));
            const cplx gb = g * b21[i];
            const cplx c11 = cfma(cconj(a21[i]) * mk, gb, a11[i] * b11[i]);
            const cplx c21 = cfma(cconj(a11[i]), gb, a21[i] * b11[i]);
            b11[i] = c11;
            b21[i] = c21;
        }
    }
    FA_STAMP(G.stamps, 5);
    fft_wg2<N2, R, 1, +1, true>(b11, b21, lds, v, 0, tw);
    FA_STAMP(G.stamps, 6);
    // ---- stores: conj twiddle and 1/N2 -------------------------------------------------------------------
    cplx *d0 = G.Z + (size_t)P * G.N1 * N2, *d1 = G.Z + ((size_t)n_out + P) * G.N1 * N2;
    const cplx wb = wbase * (1.0 / (double)N2);
#pragma unroll
    for (int i = 0; i < R; i++) {
        const cplx w = cconj((i == 0) ? wb : wb * wu[i]);
        const size_t o = yz_index(G.N1, N2, k1, v + (N2 / R) * i);
        d0[o] = b11[i] * w;
        d1[o] = b21[i] * w;
    }
    FA_STAMP(G.stamps, 7);
}

// constant term ("tail") of entry e of the product of pair P of a split level, from the factors' tails
// (and, symmetric form, the left factor's leading coefficients); sA, sB: pending scales of the factors
FA_DEV cplx split_tail_product(const TreeLevel &L, int P, int e, double sA, double sB)
{
    TailSet t;
    if (L.ne == 4) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            t.tA[q] = L.tail_in[(size_t)q * L.n_in + 2 * P] * sA;
            t.tB[q] = L.tail_in[(size_t)q * L.n_in + 2 * P + 1] * sB;
        }
        return tail_product_general(t, e);
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        t.tA[q] = L.tail_in[(size_t)q * L.n_in + 2 * P] * sA;
        t.tB[q] = L.tail_in[(size_t)q * L.n_in + 2 * P + 1] * sB;
        t.leadA[q] = L.body_in[(size_t)q * L.plane + (size_t)(2 * P) * L.d] * sA;
    }
    return tail_product_sym(t, e, L.kappa);
}

// column step of the inverse transform of every output polynomial
//   grid.x = N2/BC tiles, grid.y = 4*n_out polynomials
template <int N1, int R, int BC, bool DB> FA_DEV void body_col_inv(const BigLevel &G)
{
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    const TreeLevel &L = G.L;
    const int tid = FA_TID;
    const int c = tid % BC, v = tid / BC;
    const int n2 = FA_BID * BC + c;
    const int poly = FA_BID_Y;
    const int n_out = L.n_in / 2;
    const int e = poly / n_out, P = poly % n_out;
    const int N2 = G.N2;
    const long long N = (long long)N1 * N2;
    const int d2 = 2 * L.d;
    const cplx *src = G.Z + (size_t)poly * N1 * N2;
    const double sA = level_in_scale(L, 2 * P), sB = level_in_scale(L, 2 * P + 1);   // wave-uniform
    cplx x[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int k1 = v + (N1 / R) * i;
        x[i] = src[yz_index(N1, N2, k1, n2)];   // conj twiddle already applied by the row kernel
    }
    int parity = 0;
    fft_wg<N1, R, BC, +1, DB, true>(x, lds, v, c, G.tw1, parity);
    cplx *dst = L.body_out + (size_t)e * L.plane + (size_t)P * d2;
    const double inv = 1.0 / (double)N1;
    double m2 = 0.0;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int n1 = v + (N1 / R) * i;
        const long long idx = (long long)n1 * N2 + n2;
        cplx val = x[i] * inv;
        if (idx == 0) {
            // tails of the two factors -> constant term of the product, un-alias coefficient 0
            const cplx tp = split_tail_product(L, P, e, sA, sB);
            if (N == d2) val = val - tp;
            L.tail_out[(size_t)e * n_out + P] = tp;
            m2 = fmax(m2, cnorm2(tp));
        }
        if (idx < d2) {
            dst[idx] = val;
            m2 = fmax(m2, cnorm2(val));
        }
    }
    fa_wave_atomic_max_hi32(&L.max2_out[(size_t)P * kMax2Slots + max2_slot()], m2);   // P is uniform in the workgroup
}

// Bridge between two consecutive split levels: inverse column step of level l (N = N1*N2)
// immediately followed by the forward column step of level l+1 (2N = 2N1*N2) of the same
// polynomial, in registers.  The degree-2d coefficients never go to HBM: only their maximum
// (for the pending scale), the new tail, and coefficient 0 (the next level's "lead") are kept.
// The pending scale is not known here, so Y' is written unscaled and the row kernel of level
// l+1 applies scale_in on load (BigLevel::y_unscaled).
//   grid.x = N2/BC tiles, grid.y = ne*n_out polynomials;  lanes hold R points of the inverse and
//   2R points (upper half zero) of the forward transform.
template <int N1, int R, int BC, bool DB> FA_DEV void body_col_bridge(const BigLevel &G)
{
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    const TreeLevel &L = G.L;
    const int tid = FA_TID;
    const int c = tid % BC, v = tid / BC;
    const int n2 = FA_BID * BC + c;
    const int poly = FA_BID_Y;
    const int n_out = L.n_in / 2;
    const int e = poly / n_out, P = poly % n_out;
    const int N2 = G.N2;
    const long long N = (long long)N1 * N2;
    const int d2 = 2 * L.d;
    const cplx *src = G.Z + (size_t)poly * N1 * N2;
    const double sA = level_in_scale(L, 2 * P), sB = level_in_scale(L, 2 * P + 1);   // wave-uniform
    cplx x[2 * R];
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int k1 = v + (N1 / R) * i;
        x[i] = src[yz_index(N1, N2, k1, n2)];   // conj twiddle already applied by the row kernel
    }
    int parity = 0;
    {
        cplx lo[R];
#pragma unroll
        for (int i = 0; i < R; i++) lo[i] = x[i];
        fft_wg<N1, R, BC, +1, DB, true>(lo, lds, v, c, G.tw1, parity);
#pragma unroll
        for (int i = 0; i < R; i++) x[i] = lo[i];
    }
    const double inv = 1.0 / (double)N1;
    double m2 = 0.0;
    // constant term of the product (needed by the lanes holding index 0 and index 2d)
    auto tail_prod = [&]() -> cplx { return split_tail_product(L, P, e, sA, sB); };
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int n1 = v + (N1 / R) * i;
        const long long idx = (long long)n1 * N2 + n2;
        cplx val = x[i] * inv;
        cplx up = cmake(0.0, 0.0);   // element n1 + N1 of the zero-padded next-level column
        if (idx == 0) {
            const cplx tp = tail_prod();
            if (N == d2) {
                val = val - tp;      // un-alias coefficient 2d folded onto 0
                up = tp;             // index 2d = N of the degree-2d polynomial: its tail
            }
            L.tail_out[(size_t)e * n_out + P] = tp;
            // coefficient 0 is the only body element later levels read (tail products)
            L.body_out[(size_t)e * L.plane + (size_t)P * d2] = val;
            m2 = fmax(m2, cnorm2(tp));
        }
        if (idx < d2) m2 = fmax(m2, cnorm2(val));
        else if (idx == d2) val = tail_prod();   // N > 2d: the tail sits inside the lower half
        else val = cmake(0.0, 0.0);
        x[i] = val;
        x[R + i] = up;
    }
    fa_wave_atomic_max_hi32(&L.max2_out[(size_t)P * kMax2Slots + max2_slot()], m2);   // P is uniform in the workgroup
    // forward column step of the next level: length 2*N1, 2R points per lane
    fft_wg<2 * N1, 2 * R, BC, -1, DB>(x, lds, v, c, G.tw1x2, parity);
    cplx *dst = G.Y + (size_t)poly * (2 * N1) * N2;
#pragma unroll
    for (int i = 0; i < 2 * R; i++) {
        const int k1 = v + (N1 / R) * i;   // (2N1)/(2R) = N1/R
        dst[yz_index(2 * N1, N2, k1, n2)] = x[i];  // twiddle applied by the next level's row kernel
    }
}

// Bridge with spectral doubling.  The next level's column transform of length 2*N1
// of a column whose upper half is zero has, exactly,
//     even rows  Y'[2j]   = this level's Z row j                     (nothing to compute or move),
//     odd rows   Y'[2j+1] = DFT_N1( x[n1] * w_{2N1}^{n1} )[j]  (- t for column 0, t = product tail),
// where x = IDFT_N1(z)/N1 are the coefficients (with the alias fix x[0] -= t in column 0).  So the
// bridge is one inverse and one forward transform of length N1, R points per lane, and writes
// only the odd rows; the row kernel of the next level reads the even rows from Z in place.
template <int N1, int R, int BC, bool DB> FA_DEV void body_col_bridge2(const BigLevel &G)
{
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    const TreeLevel &L = G.L;
    const int tid = FA_TID;
    const int c = tid % BC, v = tid / BC;
    const int n2 = FA_BID * BC + c;
    const int poly = FA_BID_Y;
    const int n_out = L.n_in / 2;
    const int e = poly / n_out, P = poly % n_out;
    const int N2 = G.N2;
    const int d2 = 2 * L.d;
    const cplx *src = G.Z + (size_t)poly * N1 * N2;
    const double sA = level_in_scale(L, 2 * P), sB = level_in_scale(L, 2 * P + 1);   // wave-uniform
    cplx x[R];
#pragma unroll
    for (int i = 0; i < R; i++) x[i] = src[yz_index(N1, N2, v + (N1 / R) * i, n2)];
    int parity = 0;
    fft_wg<N1, R, BC, +1, DB, true>(x, lds, v, c, G.tw1, parity);
    const double inv = 1.0 / (double)N1;
    double m2 = 0.0;
    cplx tp = cmake(0.0, 0.0);
    if (n2 == 0 && v == 0) {   // the lane that holds index 0
        tp = split_tail_product(L, P, e, sA, sB);
        L.tail_out[(size_t)e * n_out + P] = tp;
        m2 = cnorm2(tp);
    }
    // N = 2d: coefficient 2d is folded onto 0 (un-aliased here) and the tail sits at index N, the first element
    // of the zero upper half.  N > 2d (degrees that are not powers of two: fnft_kdvv): nothing is aliased, the
    // tail is element 2d of the lower half, and what the transform left above it is rounding noise (zeroed here;
    // the even rows, which are this level's Z rows, keep it -- at the level of the transform's own error)
    const bool exact2d = ((long long)N1 * N2 == (long long)d2);
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int n1 = v + (N1 / R) * i;
        const long long idx = (long long)n1 * N2 + n2;
        cplx val = x[i] * inv;
        if (idx == 0) {
            if (exact2d) val = val - tp;   // un-alias coefficient 2d folded onto 0
            L.body_out[(size_t)e * L.plane + (size_t)P * d2] = val;   // the next level's "lead"
        }
        if (!exact2d && idx > d2) val = cmake(0.0, 0.0);
        m2 = fmax(m2, cnorm2(val));
        x[i] = val * G.tw1x2[n1];   // * w_{2N1}^{n1}
    }
    fa_wave_atomic_max_hi32(&L.max2_out[(size_t)P * kMax2Slots + max2_slot()], m2);   // P is uniform in the workgroup
    fft_wg<N1, R, BC, -1, DB, true>(x, lds, v, c, G.tw1, parity);
    cplx *dst = G.Y + (size_t)poly * N1 * N2;   // odd rows only: [poly][j][n2]
    // N = 2d, column 0: the tail at index N contributes t * w_{2N1}^{N1 (2j+1)} = -t to every odd row
    const bool col0 = exact2d && (n2 == 0);
    cplx tpc = cmake(0.0, 0.0);
    if (col0) {
        // every lane of column 0 needs t: recompute it (cheap, uniform within the few lanes)
        tpc = (v == 0) ? tp : split_tail_product(L, P, e, sA, sB);
    }
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int j = v + (N1 / R) * i;
        dst[yz_index(N1, N2, j, n2)] = x[i] - tpc;
    }
}

// one lane per output matrix of the LAST split level: turn its maxima into the pending scale and
// add its exponent (intermediate split levels are finalized by their consumers, TreeLevel::in_pending)
FA_DEV void body_finalize_scales(const TreeLevel &L)
{
    // one wave per output matrix: the kMax2Slots maxima are read by one lane each
    const long long P = FA_BID;
    const unsigned hi = fa_slots_max_u32(L.max2_out + (size_t)P * kMax2Slots);
    if (FA_TID != 0) return;
    const int a = exponent_of_max2(hi);
    L.scale_out[P] = pow2i(-a);
    L.wexp_out[P] = L.wexp_out[P] + a;
}

// ---------------------------------------------------------------------------------------------
// level 0 of a tree from n coefficient matrices that are already in DEVICE memory in the reference's input layout
// (fnft__poly_fmult.c:398-401: entry-major, matrix j of the product at p[e*n*(deg+1) + j*(deg+1) + k]) -- the body/tail
// layout of this file, identity padding z^deg * I up to the power of two (:422-438), scale 1, exponent 0
struct ImportParams {
    const cplx *p;
    cplx *body, *tail;
    double *scale;
    int *wexp;
    size_t plane;
    long long n, npad, deg;
};
FA_DEV void body_import_level0(const ImportParams &P)
{
    const long long i = (long long)FA_BID * FA_BDIM + FA_TID;
    const long long w = P.deg + 1;
    if (i >= 4 * P.npad * w) return;
    const int e = (int)(i / (P.npad * w));
    const long long r = i % (P.npad * w);
    const long long j = r / w, k = r % w;
    cplx v = cmake(0.0, 0.0);
    if (j < P.n) v = P.p[(long long)e * P.n * w + j * w + k];
    else if (k == 0 && (e == 0 || e == 3)) v = cmake(1.0, 0.0);
    if (k < P.deg) P.body[(size_t)e * P.plane + (size_t)(j * P.deg + k)] = v;
    else P.tail[(size_t)e * P.npad + j] = v;
    if (e == 0 && k == 0) { P.scale[j] = 1.0; P.wexp[j] = 0; }
}

// ---------------------------------------------------------------------------------------------
// final transfer matrix in the reference's result layout (fnft__poly_fmult.c:522-538):
// [r11|r12|r21|r22], each deg+1 coefficients, highest power first; deg = D*deg0 (the identity
// padding contributes trailing zero coefficients which are dropped).
// ---------------------------------------------------------------------------------------------
struct ExportParams {
    const cplx *body;   // ne planes; signal b's matrix at b*deg_tot
    const cplx *tail;   // ne planes of batch
    const double *scale;
    cplx *out;          // batch * 4*(deg+1)
    size_t plane;
    long long deg_tot;  // Dpad*deg0
    long long deg;      // D*deg0
    int batch;
    int ne;             // 4 general, 2 symmetric
    int kappa;
    long long out_stride = 0;   // != 0 (batch 1): entry e goes to out[e*out_stride + k] instead of out[e*(deg+1) + k]
    const int *W = nullptr;     // != NULL: values are multiplied by 2^W[0] (un-normalised result)
    int real_layout = 0;        // body/tail are arrays of double (real-coefficient path, nft_real.h)
};
// coefficient k (highest power first, k <= deg) of stored plane s of signal b.  General form:
// identity padding z^deg0*I leaves TRAILING zeros, index k.  Symmetric form: padding with the
// zero-sample matrix diag(1, z^deg0) leaves the first column unchanged as polynomials, i.e.
// deg_tot - deg LEADING zeros, index k + (deg_tot - deg).
FA_DEV cplx stored_coef(const cplx *body, const cplx *tail, size_t plane, long long deg_tot,
                        long long deg, int batch, int ne, int s, int b, long long k)
{
    const long long kk = (ne == 2) ? k + (deg_tot - deg) : k;
    if (kk < deg_tot) return body[(size_t)s * plane + (size_t)b * deg_tot + kk];
    return tail[(size_t)s * batch + b];
}
// the same for the real-coefficient layout (arrays of double, general form only)
FA_DEV cplx stored_coef_real(const cplx *body, const cplx *tail, size_t plane, long long deg_tot, int batch, int s, int b,
                             long long k)
{
    if (k < deg_tot) return cmake(((const double *)body)[(size_t)s * plane + (size_t)b * deg_tot + k], 0.0);
    return cmake(((const double *)tail)[(size_t)s * batch + b], 0.0);
}
FA_DEV void body_export_tm(const ExportParams &E)
{
    const long long gid = (long long)FA_BID * FA_BDIM + FA_TID;
    const long long per = 4 * (E.deg + 1);
    if (gid >= per * E.batch) return;
    const int b = (int)(gid / per);
    const long long r = gid % per;
    const int e = (int)(r / (E.deg + 1));
    const long long k = r % (E.deg + 1);
    const double sc = E.scale[b];
    cplx val;
    if (E.real_layout) {
        val = stored_coef_real(E.body, E.tail, E.plane, E.deg_tot, E.batch, e, b, k);
    } else if (E.ne == 4) {
        val = stored_coef(E.body, E.tail, E.plane, E.deg_tot, E.deg, E.batch, 4, e, b, k);
    } else if (e == 0 || e == 2) {       // 11, 21 are stored
        val = stored_coef(E.body, E.tail, E.plane, E.deg_tot, E.deg, E.batch, 2, e >> 1, b, k);
    } else if (e == 3) {                 // p22[k] = conj(p11[deg-k])
        val = cconj(stored_coef(E.body, E.tail, E.plane, E.deg_tot, E.deg, E.batch, 2, 0, b, E.deg - k));
    } else {                             // p12[k] = -kappa*conj(p21[deg-k])
        val = cconj(stored_coef(E.body, E.tail, E.plane, E.deg_tot, E.deg, E.batch, 2, 1, b, E.deg - k))
              * (double)(-E.kappa);
    }
    const double sw = E.W ? sc * pow2i(E.W[0]) : sc;
    if (E.out_stride != 0) E.out[(long long)e * E.out_stride + k] = val * sw;
    else E.out[gid] = val * sw;
}

// ---------------------------------------------------------------------------------------------
// chirp z-transform (fnft__poly_chirpz.c:52-95) of the a- and b-polynomials + spectrum epilogue
// (fnft_nsev.c:837-884).  Transform length Lc = N1*N2 as above.
// cpow(z, x) for real x is evaluated as exp(x*log z) like libm does, with log z formed on the
// host (so that a |z| that is not exactly 1 after rounding has the same effect as in the
// reference).
// ---------------------------------------------------------------------------------------------
struct ChirpParams {
    // polynomials: entries 11 (slot 0) and 21 (slot 1) of the final transfer matrix, or an
    // explicit coefficient array for the stand-alone fnft__poly_chirpz entry point
    const cplx *body;    // 4 planes (tree result) or NULL
    const cplx *tail;
    const double *scale;
    const cplx *poly;    // explicit polynomial(s): npoly * (deg+1), highest first, or NULL
    int poly_tm;         // poly holds whole transfer matrices (4 entries of deg+1 per signal); entry[] picks two
    size_t plane;
    long long deg_tot;   // Dpad*deg0 (tree layout)
    long long deg;       // polynomial degree actually evaluated
    int batch;
    int npoly;           // polynomials per signal (2 for nsev, 1 for stand-alone)
    int entry[2];        // stored planes to evaluate (general: 0 and 2; symmetric: 0 and 1)
    int ne;              // stored entries per matrix (4 general, 2 symmetric)
    // chirp
    double logA[2], logW[2];  // log A, log W (complex)
    long long M;
    int N1, N2;          // Lc = N1*N2
    cplx *Ybuf;          // batch*npoly*Lc
    cplx *Vbuf;          // Lc
    cplx *Hbuf;          // stand-alone: output M per polynomial (host copies)
    BigTwiddle btw;
    const cplx *tw1, *tw2;
    // epilogue
    cplx *contspec;      // batch * cs_len
    const int *W;        // per signal
    int *status;         // bit 1: division by zero
    double xi0, eps_xi, pf_rho, pf_a, pf_b;
    int cstype;          // fnft_nsev_cstype_t ordinal, or -1: raw H values to Hbuf
    int use_W;
    int jobs_per_group;  // row step: jobs handled by one workgroup (grid.y = ceil(jobs/this))
    // DFT mode (band-limited resampling, fnft__misc.c:326-407): A = 1, W = exp(dft_sign*2*pi*i/dft_len);
    // the chirp W^(n^2/2) = cis(dft_sign*pi*(n^2 mod 2*dft_len)/dft_len) is formed from the integer
    // residue, and poly holds coefficients in ascending order (x[n] multiplies z^n)
    long long dft_len;   // 0: off
    int dft_sign;
    // spectrum of the chirp filter v (depends on W, M, deg and the transform length only): kept
    // between calls with the same grids.  v_mode 0: compute; 1: compute and store; 2: load
    cplx *VS;
    int v_mode;
    int real_layout;     // body/tail are arrays of double (real-coefficient path, nft_real.h)
    // != NULL: the tree's root has not been finalized (body_finalize_scales): the column kernel reduces the 64 maxima
    // of its signal itself (scale = 2^-a instead of scale[b]) and one workgroup per signal stores scale and exponent
    const unsigned *fin_max2;
    double *fin_scale;
    int *fin_wexp;
};

// exp((xr + i*xi)) with a real multiplier folded in: returns exp(t*lr) * cis(t*li)
// exp(x) for the tiny exponents t*log|z| of numbers that are on the unit circle up to rounding
FA_DEV double exp_small(double x)
{
    if (x == 0.0) return 1.0;
    if (fabs(x) < 1.0e-5) return 1.0 + x * (1.0 + 0.5 * x);   // relative error < 2e-16
    return exp(x);
}
FA_DEV cplx cpow_real(const double lg[2], double t)
{
    double s, c;
    fa_sincos(t * lg[1], &s, &c);
    const double mag = exp_small(t * lg[0]);
    return cmake(mag * c, mag * s);
}
// cpow(A, ta) * cpow(W, tw) with one sincos and one exp: the two exponents are added (one extra
// rounding of the phase, < 1e-16 relative to its magnitude)
FA_DEV cplx cpow_real2(const double la[2], double ta, const double lw[2], double tw)
{
    double s, c;
    fa_sincos(fma(ta, la[1], tw * lw[1]), &s, &c);
    const double mag = exp_small(fma(ta, la[0], tw * lw[0]));
    return cmake(mag * c, mag * s);
}

// W^(half * n^2), half = +-0.5
template <bool DFT> FA_DEV cplx chirp_w(const ChirpParams &C, long long n, double half)
{
    if (DFT) {
        const unsigned long long r = ((unsigned long long)n * (unsigned long long)n)
                                     % (2ull * (unsigned long long)C.dft_len);
        double s, c;
        fa_sincos(3.141592653589793238462643383279502884 * ((double)r / (double)C.dft_len), &s, &c);
        const double sg = (half > 0.0 ? 1.0 : -1.0) * (double)C.dft_sign;
        return cmake(c, sg * s);
    }
    const double dn = (double)n;
    return cpow_real(C.logW, half * dn * dn);
}

// sc: the signal's pending scale (C.scale[b], or formed by the caller from the unreduced maxima)
template <bool DFT> FA_DEV cplx chirp_poly_coef(const ChirpParams &C, int b, int slot, long long k, double sc)
{
    // coefficient k (highest power first) of polynomial `slot` of signal b
    if (DFT)
        return C.poly[((size_t)b * C.npoly + slot) * (size_t)(C.deg + 1) + (size_t)(C.deg - k)];
    if (C.poly && C.poly_tm)   // transfer matrices in the reference layout [r11|r12|r21|r22] per signal
        return C.poly[((size_t)b * 4 + C.entry[slot]) * (size_t)(C.deg + 1) + (size_t)k];
    if (C.poly) return C.poly[((size_t)b * C.npoly + slot) * (size_t)(C.deg + 1) + (size_t)k];
    if (C.real_layout)
        return stored_coef_real(C.body, C.tail, C.plane, C.deg_tot, C.batch, C.entry[slot], b, k) * sc;
    return stored_coef(C.body, C.tail, C.plane, C.deg_tot, C.deg, C.batch, C.ne, C.entry[slot], b, k) * sc;
}

// column step (forward) of the chirp-premultiplied polynomials and of the chirp filter
//   grid.x = N2/BC, grid.y = batch*npoly + 1 (the last one is the filter v)
// (Measured and dropped in round 3: one workgroup per signal that forms the chirp factor once for both polynomials --
//  35 -> 39.5 us at cfg 2, half the workgroups; and a 30-instruction sincos for |x| < 2^17 in place of the library's --
//  no change: the column kernels are not bound by their transcendentals.)
template <int N1, int R, int BC, bool DB, bool DFT> FA_DEV void body_chirp_col_fwd(const ChirpParams &C)
{
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    const int tid = FA_TID;
    const int c = tid % BC, v = tid / BC;
    const int n2 = FA_BID * BC + c;
    const int job = FA_BID_Y;
    const int njobs = C.batch * C.npoly;
    const long long Lc = (long long)N1 * C.N2;
    const long long Np = C.deg + 1;
    // pending scale of this job's signal: stored, or (root not finalized yet) from the 64 unreduced maxima -- every
    // lane of every wave is active here; the first workgroup of the signal's first polynomial stores the result
    double sc = 1.0;
    if (job < njobs && !DFT && C.poly == nullptr) {
        const int b = job / C.npoly;
        if (C.fin_max2 != nullptr) {
            const int a = exponent_of_max2(fa_slots_max_u32(C.fin_max2 + (size_t)b * kMax2Slots));
            sc = pow2i(-a);
            if (FA_BID == 0 && job % C.npoly == 0 && tid == 0) {
                C.fin_scale[b] = sc;
                C.fin_wexp[b] = C.fin_wexp[b] + a;
            }
        } else {
            sc = C.scale[b];
        }
    }
    cplx x[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int n1 = v + (N1 / R) * i;
        const long long n = (long long)n1 * C.N2 + n2;
        cplx val = cmake(0.0, 0.0);
        const double dn = (double)n;
        if (job < njobs) {
            if (n < Np) {  // :68-69  p[deg-n] * A^-n * W^(n^2/2)
                const cplx pc = chirp_poly_coef<DFT>(C, job / C.npoly, job % C.npoly, C.deg - n, sc);
                val = DFT ? pc * chirp_w<true>(C, n, 0.5)
                          : pc * cpow_real2(C.logA, -dn, C.logW, 0.5 * dn * dn);
            }
        } else {  // :76-82
            if (n < C.M) val = chirp_w<DFT>(C, n, -0.5);
            else if (n > Lc - Np) val = chirp_w<DFT>(C, Lc - n, -0.5);
        }
        x[i] = val;
    }
    int parity = 0;
    fft_wg<N1, R, BC, -1, DB, true>(x, lds, v, c, C.tw1, parity);
    // the twiddle w_L^{n2 k1} is applied by the row kernel (shared by all rows of one k1)
    cplx *dst = (job < njobs) ? C.Ybuf + (size_t)job * Lc : C.Vbuf;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int k1 = v + (N1 / R) * i;
        dst[yz_index(N1, C.N2, k1, n2)] = x[i];
    }
}

// row step: per k1: V row forward; for each job of the group: Y row forward, * V, inverse,
// store in place.   grid.x = N1, grid.y = job groups; one workgroup of N2/R lanes
template <int N2, int R, bool DB> FA_DEV void body_chirp_rows(const ChirpParams &C)
{
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    const int v = FA_TID;
    const int k1 = FA_BID;
    const long long Lc = (long long)C.N1 * N2;
    int parity = 0;
    // w_L^{k1 n2} for n2 = v + (N2/R) i: per-lane base times workgroup-uniform factors
    const cplx base = big_twiddle(C.btw, (unsigned)k1 * (unsigned)v);
    cplx tw[R];
    tw[0] = base;
#pragma unroll
    for (int i = 1; i < R; i++) tw[i] = base * big_twiddle(C.btw, (unsigned)k1 * (unsigned)((N2 / R) * i));
    cplx vv[R];
    if (C.v_mode == 2) {
        const cplx *vs = C.VS + (size_t)k1 * N2;
#pragma unroll
        for (int i = 0; i < R; i++) vv[i] = vs[v + (N2 / R) * i];
    } else {
#pragma unroll
        for (int i = 0; i < R; i++) vv[i] = C.Vbuf[yz_index(C.N1, N2, k1, v + (N2 / R) * i)] * tw[i];
        fft_wg<N2, R, 1, -1, DB, true>(vv, lds, v, 0, C.tw2, parity);
        if (C.v_mode == 1 && FA_BID_Y == 0) {
            cplx *vd = C.VS + (size_t)k1 * N2;
#pragma unroll
            for (int i = 0; i < R; i++) vd[v + (N2 / R) * i] = vv[i];
        }
    }
    const int njobs = C.batch * C.npoly;
    const double inv = 1.0 / (double)N2;
    const int job0 = FA_BID_Y * C.jobs_per_group;
    const int job1 = (job0 + C.jobs_per_group < njobs) ? job0 + C.jobs_per_group : njobs;
    for (int job = job0; job < job1; job++) {
        cplx *ys = C.Ybuf + (size_t)job * Lc;
        cplx y[R];
#pragma unroll
        for (int i = 0; i < R; i++) y[i] = ys[yz_index(C.N1, N2, k1, v + (N2 / R) * i)] * tw[i];
        fft_wg<N2, R, 1, -1, DB, true>(y, lds, v, 0, C.tw2, parity);
#pragma unroll
        for (int i = 0; i < R; i++) y[i] = y[i] * vv[i];
        fft_wg<N2, R, 1, +1, DB, true>(y, lds, v, 0, C.tw2, parity);
#pragma unroll
        for (int i = 0; i < R; i++) ys[yz_index(C.N1, N2, k1, v + (N2 / R) * i)] = (y[i] * inv) * cconj(tw[i]);
    }
}

// column step (inverse) + chirp post-multiplication + spectrum epilogue
//   grid.x = N2/BC, grid.y = batch
template <int N1, int R, int BC, bool DB, bool DFT, bool KDV> FA_DEV void body_chirp_col_inv(const ChirpParams &C)
{
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    const int tid = FA_TID;
    const int c = tid % BC, v = tid / BC;
    const int n2 = FA_BID * BC + c;
    const int b = FA_BID_Y;
    const long long Lc = (long long)N1 * C.N2;
    const double inv = 1.0 / (double)N1;
    cplx H[2][R];
    int parity = 0;
#pragma unroll
    for (int slot = 0; slot < 2; slot++) {
        if (slot >= C.npoly) {
#pragma unroll
            for (int i = 0; i < R; i++) H[slot][i] = cmake(0.0, 0.0);
            continue;
        }
        const cplx *src = C.Ybuf + ((size_t)b * C.npoly + slot) * Lc;
        cplx x[R];
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int k1 = v + (N1 / R) * i;
            x[i] = src[yz_index(N1, C.N2, k1, n2)];   // conj twiddle applied by the row kernel
        }
        fft_wg<N1, R, BC, +1, DB, true>(x, lds, v, c, C.tw1, parity);
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int n1 = v + (N1 / R) * i;
            const long long m = (long long)n1 * C.N2 + n2;
            const double dm = (double)m;
            // V[m] / L here (1/N2 was applied by the row step); W^(m^2/2) follows below, once
            (void)dm;
            H[slot][i] = (m < C.M) ? x[i] * inv : cmake(0.0, 0.0);
        }
    }
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int n1 = v + (N1 / R) * i;
        const long long m = (long long)n1 * C.N2 + n2;
        if (m >= C.M) continue;
        {   // :94-95  result[m] = W^(m^2/2) * V[m] / L
            const cplx cw = chirp_w<DFT>(C, m, 0.5);
            H[0][i] = H[0][i] * cw;
            H[1][i] = H[1][i] * cw;
        }
        if (C.cstype < 0) {  // raw chirp-z values
#pragma unroll
            for (int slot = 0; slot < 2; slot++)
                if (slot < C.npoly)
                    C.Hbuf[((size_t)b * C.npoly + slot) * (size_t)C.M + (size_t)m] = H[slot][i];
            continue;
        }
        const double xi = C.xi0 + C.eps_xi * (double)m;  // fnft_nsev.c:784-785
        if (KDV) {
            // KdV reflection coefficient, fnft_kdvv.c:186-203: slots hold H12 and H22, xi runs over
            // -(XI0 + m*eps_xi); pf_a = -eps_t/deg undoes the 2SPLIT2A base change (else 0),
            // pf_rho = 2*(T1 + eps_t/2).  No zero test: the reference divides unguarded.
            double s, co;
            cplx h12 = H[0][i];
            if (C.pf_a != 0.0) {
                fa_sincos(xi * C.pf_a, &s, &co);
                h12 = h12 * cmake(co, s);
            }
            fa_sincos(xi * C.pf_rho, &s, &co);
            const cplx h22 = H[1][i];
            const cplx den = cmake(-2.0 * xi * h22.y - h12.x, 2.0 * xi * h22.x - h12.y);  // 2 i xi H22 - H12
            C.contspec[(size_t)b * C.M + m] = c_div(h12 * cmake(co, s), den);
            continue;
        }
        const long long cs_len = C.M * (C.cstype == 0 ? 1 : (C.cstype == 1 ? 2 : 3));
        cplx *out = C.contspec + (size_t)b * cs_len;
        long long off = 0;
        const cplx h11 = H[0][i], h21 = H[1][i];
        if (C.cstype == 0 || C.cstype == 2) {  // :844-855
            if (h11.x == 0.0 && h11.y == 0.0) {
                fa_atomic_or_i32(C.status, 2);
            } else {
                double s, co;
                fa_sincos(xi * C.pf_rho, &s, &co);
                out[m] = c_div(h21 * cmake(co, s), h11);
            }
            off = C.M;
        }
        if (C.cstype == 1 || C.cstype == 2) {  // :861-876
            const double scale = C.use_W ? ldexp(1.0, C.W[b]) : 1.0;
            double s, co;
            fa_sincos(xi * C.pf_a, &s, &co);
            out[off + m] = (h11 * scale) * cmake(co, s);
            fa_sincos(xi * C.pf_b, &s, &co);
            out[off + C.M + m] = (h21 * scale) * cmake(co, s);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 4SPLIT4A/B front end (fnft__nse_discretization.c:474-503): two band-limited resamplings of the
// signal at -/+ delta (fnft__misc.c:326-407: DFT, phase ramp, inverse DFT -- the DFTs are chirp
// transforms in DFT mode, any length) and their weighted combination, two samples per step.
// ---------------------------------------------------------------------------------------------
struct ResampleParams {
    const cplx *X;     // batch * Din: spectrum of the signal
    cplx *X12;         // batch * 2 * Din: spectra of the two shifted copies
    const cplx *Q12;   // batch * 2 * Din: the two shifted copies times Din (unnormalised inverse DFT)
    cplx *qpre;        // batch * 2 * Dsub
    long long Din, Dsub, nskip;
    int batch;
    double delta_over_span;   // delta / (Din * eps_t)
    double w0, w1;            // 0.25 + sqrt(3)/6, 0.25 - sqrt(3)/6
    int *status;              // bit 2: "signal does not appear to be bandlimited" (body_band_check)
};

// fnft__misc.c:371-381: the resampler's own check that the spectrum has decayed -- the l2 norms (trapezoidal rule,
// fnft__misc.c:90-112) of the two 5 % bands next to the Nyquist bin against the norm of the whole spectrum; above
// sqrt(eps) the interpolation may be inaccurate and the reference warns.  One workgroup per signal; sets bit 2 of
// the status word (a warning, not an error).
FA_DEV void body_band_check(const ResampleParams &P)
{
    FA_LDS_DECL
    double *red = (double *)FA_LDS_PTR;   // 3 * blockDim partial sums
    const long long D = P.Din, Dlp = D / 20;
    const cplx *X = P.X + (size_t)FA_BID * D;
    double s_all = 0.0, s_lo = 0.0, s_hi = 0.0;
    for (long long i = FA_TID; i < D; i += FA_BDIM) {
        const double m = cnorm2(X[i]);
        s_all += ((i == 0 || i == D - 1) ? 0.5 : 1.0) * m;
        const long long a = i - (D / 2 - 1 - Dlp), b = i - (D / 2 + 1);   // position inside the two bands
        if (a >= 0 && a < Dlp) s_lo += ((a == 0 || a == Dlp - 1) ? 0.5 : 1.0) * m;
        if (b >= 0 && b < Dlp) s_hi += ((b == 0 || b == Dlp - 1) ? 0.5 : 1.0) * m;
    }
    red[FA_TID] = s_all; red[FA_BDIM + FA_TID] = s_lo; red[2 * FA_BDIM + FA_TID] = s_hi;
    FA_SYNC();
    if (FA_TID == 0) {
        double ta = 0.0, tl = 0.0, th = 0.0;
        for (int t = 0; t < FA_BDIM; t++) { ta += red[t]; tl += red[FA_BDIM + t]; th += red[2 * FA_BDIM + t]; }
        // tmp = sqrt(lo + hi)/sqrt(all) > sqrt(eps)  <=>  lo + hi > eps * all   (Dlp >= 2: the reference's norm of fewer
        // than two points is NaN and never warns)
        if (Dlp >= 2 && (tl + th) > 2.220446049250313e-16 * ta) fa_atomic_or_i32(P.status, 4);
    }
}

FA_DEV void body_resample_phase(const ResampleParams &P)
{
    const long long gid = (long long)FA_BID * FA_BDIM + FA_TID;
    if (gid >= (long long)P.batch * P.Din) return;
    const long long b = gid / P.Din, i = gid % P.Din;
    // freq[i]*delta, :383-390: i/(D eps) below D/2, (i - D)/(D eps) from D/2 on
    const double f = (i < P.Din / 2) ? (double)i : (double)i - (double)P.Din;
    double s, c;
    fa_sincos(2.0 * 3.141592653589793238462643383279502884 * (P.delta_over_span * f), &s, &c);
    const cplx x = P.X[gid];
    P.X12[(size_t)(2 * b) * P.Din + i] = x * cmake(c, -s);      // shift by -delta
    P.X12[(size_t)(2 * b + 1) * P.Din + i] = x * cmake(c, s);   // shift by +delta
}

FA_DEV void body_resample_combine(const ResampleParams &P)
{
    const long long gid = (long long)FA_BID * FA_BDIM + FA_TID;
    if (gid >= (long long)P.batch * P.Dsub) return;
    const long long b = gid / P.Dsub, is = gid % P.Dsub;
    const long long i = is * P.nskip;
    const double inv = 1.0 / (double)P.Din;
    const cplx q1 = P.Q12[(size_t)(2 * b) * P.Din + i] * inv;
    const cplx q2 = P.Q12[(size_t)(2 * b + 1) * P.Din + i] * inv;
    cplx *out = P.qpre + (size_t)b * 2 * P.Dsub + 2 * is;
    out[0] = q1 * P.w0 + q2 * P.w1;
    out[1] = q1 * P.w1 + q2 * P.w0;
}

// ---------------------------------------------------------------------------------------------
// Discrete spectrum.  The reference refines and characterises bound states with a
// sequential O(K*D) scatterer (fnft__nse_scatter_bound_states.c:40-668; Boffetta-Osborne for the
// 2SPLIT schemes, CF4_2 for 4SPLIT4A/B, fnft_nsev.c:669-676,937-942).  The step matrices are
// independent, so here the D steps are cut into chunks: one lane per (chunk, eigenvalue) forms the
// chunk's 2x2 matrix (and its lambda-derivative), one lane per eigenvalue strings the chunks
// together, and the per-point quantities needed for b are recomputed chunk-parallel from the
// chunk boundary vectors.
//   U(l) = [[ch - i l sh, q sh], [r sh, ch + i l sh]],  k^2 = q r - l^2, ch = cosh(k e), sh = sinh(k e)/k
//   V = dU/dl from dk/dl = -l/k;  r = -conj(q)  (kappa = +1 is the only case with bound states)
// ---------------------------------------------------------------------------------------------
struct BsParams {
    const cplx *q;       // D preprocessed samples of one signal
    long long D;
    int ups;             // samples per grid point: 1 BO, 2 CF4_2 and the half-step eigenfunctions of the inverse transform
    double lscale;       // factor on the spectral parameter (and on a'): 1/2 for CF4_2, 1 otherwise
    double T0, T1, eps;  // eps = (T1 - T0)/(D/ups - 1)
    int K;
    const cplx *lam;     // K eigenvalue candidates
    int L;               // samples per chunk, multiple of ups
    int nchunk;
    cplx *cm;            // K*nchunk*8: forward {M[4], M'[4]}; backward uses the first 4
    cplx *bnd;           // K*(nchunk+1)*2: phi at the chunk starts, [nchunk] = end of the grid
    cplx *bndp;          // K*(nchunk+1)*2: psi at the same points
    cplx *PHI;           // K*(D/ups+1)*2: phi at every grid point
    cplx *PSI;           // K*(D/ups+1)*2: psi at every grid point (body_bs_psi only)
    const cplx *r;       // NULL: r = -kappa conj(q) (forward chunk kernel; the others are focusing-only)
    int kappa;           // -1: defocusing; anything else: focusing
    cplx *smat;          // body_bs_matrix: K * (4 or 8) values [S11 S12 S21 S22 (S11' S12' S21' S22')]
    int with_deriv;
    cplx *best;          // K*nchunk*2: {metric, 0}, b of the best point of the chunk
    cplx *a, *aprime, *b;
};

FA_DEV void c_cosh_sinh(cplx w, cplx &ch, cplx &sh)
{
    double s, c;
    fa_sincos(w.y, &s, &c);
    const double chr = cosh(w.x), shr = sinh(w.x);
    ch = cmake(chr * c, shr * s);
    sh = cmake(shr * c, chr * s);
}

struct BsStep { cplx u00, u01, u10, u11; };
// step matrix for step size e (e < 0: inverse step); V (derivative) only when WITH_D
template <bool WITH_D> FA_DEV void bs_step_r(cplx q, cplx r, cplx l, double e, BsStep &U, BsStep &V);
template <bool WITH_D> FA_DEV void bs_step(cplx q, cplx l, double e, BsStep &U, BsStep &V)
{
    bs_step_r<WITH_D>(q, cmake(-q.x, q.y), l, e, U, V);   // r = -conj(q): focusing
}
template <bool WITH_D> FA_DEV void bs_step_r(cplx q, cplx r, cplx l, double e, BsStep &U, BsStep &V)
{
    const cplx ks = q * r - l * l;
    const cplx k = c_sqrt(ks);
    cplx ch, shk;
    c_cosh_sinh(k * e, ch, shk);
    const bool nz = (ks.x != 0.0 || ks.y != 0.0);
    const cplx sh = nz ? c_div(shk, k) : cmake(e, 0.0);
    const cplx il = cmake(-l.y, l.x);   // i*l
    const cplx ilsh = il * sh;
    U.u00 = ch - ilsh;
    U.u01 = q * sh;
    U.u10 = r * sh;
    U.u11 = ch + ilsh;
    if (WITH_D) {
        const cplx g = c_div(ch * e - sh, ks);           // -(d sh/dl)/l
        const cplx ish = cmake(-sh.y, sh.x);
        const cplx t = (l * sh) * e;                     // e l sh
        const cplx ill_g = (il * l) * g;                 // i l^2 g
        V.u00 = cmake(0.0, 0.0) - t - ish + ill_g;
        V.u11 = cmake(0.0, 0.0) - t + ish - ill_g;
        const cplx lg = l * g;
        V.u01 = cmake(0.0, 0.0) - q * lg;
        V.u10 = cmake(0.0, 0.0) - r * lg;
    }
}

// chunk matrices.  grid.x = nchunk lanes / THREADS, grid.y = K.  BACKWARD: inverse steps, last sample first
template <bool BACKWARD> FA_DEV void body_bs_chunk(const BsParams &P)
{
    const int c = FA_BID * FA_BDIM + FA_TID, e = FA_BID_Y;
    if (c >= P.nchunk) return;
    const cplx l = P.lam[e] * P.lscale;
    const long long n0 = (long long)c * P.L;
    const long long n1 = (n0 + P.L < P.D) ? n0 + P.L : P.D;
    cplx m00 = cmake(1.0, 0.0), m01 = cmake(0.0, 0.0), m10 = m01, m11 = m00;
    cplx d00 = m01, d01 = m01, d10 = m01, d11 = m01;
    BsStep U, V;
    if (!BACKWARD) {
        for (long long n = n0; n < n1; n++) {
            const cplx qn = P.q[n];
            const cplx rn = P.r ? P.r[n] : (P.kappa < 0 ? cmake(qn.x, -qn.y) : cmake(-qn.x, qn.y));
            bs_step_r<true>(qn, rn, l, P.eps, U, V);
            // M' <- V M + U M',  M <- U M
            const cplx e00 = V.u00 * m00 + V.u01 * m10 + U.u00 * d00 + U.u01 * d10;
            const cplx e01 = V.u00 * m01 + V.u01 * m11 + U.u00 * d01 + U.u01 * d11;
            const cplx e10 = V.u10 * m00 + V.u11 * m10 + U.u10 * d00 + U.u11 * d10;
            const cplx e11 = V.u10 * m01 + V.u11 * m11 + U.u10 * d01 + U.u11 * d11;
            d00 = e00; d01 = e01; d10 = e10; d11 = e11;
            const cplx f00 = U.u00 * m00 + U.u01 * m10, f01 = U.u00 * m01 + U.u01 * m11;
            const cplx f10 = U.u10 * m00 + U.u11 * m10, f11 = U.u10 * m01 + U.u11 * m11;
            m00 = f00; m01 = f01; m10 = f10; m11 = f11;
        }
    } else {
        for (long long n = n1; n-- > n0;) {
            bs_step<false>(P.q[n], l, -P.eps, U, V);
            const cplx f00 = U.u00 * m00 + U.u01 * m10, f01 = U.u00 * m01 + U.u01 * m11;
            const cplx f10 = U.u10 * m00 + U.u11 * m10, f11 = U.u10 * m01 + U.u11 * m11;
            m00 = f00; m01 = f01; m10 = f10; m11 = f11;
        }
    }
    cplx *o = P.cm + ((size_t)e * P.nchunk + c) * 8;
    o[0] = m00; o[1] = m01; o[2] = m10; o[3] = m11;
    if (!BACKWARD) { o[4] = d00; o[5] = d01; o[6] = d10; o[7] = d11; }
}

// one workgroup per eigenvalue: string the chunks together.  Forward: a, a' (:627-628) and phi at the
// chunk starts; backward: psi at the chunk ends.  Three stages instead of one walk over all chunks (a chain of
// nchunk dependent matrix-vector products, each behind a global load): (1) lane t composes the maps of its own
// run of G = ceil(nchunk/lanes) chunks into one map {M, M'}: (p, d) -> (M p, M' p + M d); (2) lane 0 carries the
// vector over the lanes' maps (in LDS) and leaves every run's start vector; (3) lane t walks its run again from
// that vector and writes the vectors at its chunk boundaries.
template <bool BACKWARD> FA_DEV void body_bs_combine(const BsParams &P)
{
    FA_LDS_DECL
    cplx *gm = (cplx *)FA_LDS_PTR;             // lanes x 8: the run's map {M[4], M'[4]}
    cplx *gv = gm + (size_t)FA_BDIM * 8;       // lanes x 4: the run's start vector (p1, p2, d1, d2)
    const int e = FA_BID, t = FA_TID, nl = FA_BDIM;
    const cplx lc = P.lam[e];
    const double bc = 0.5;
    const cplx *cmv = P.cm + (size_t)e * P.nchunk * 8;
    const int G = (P.nchunk + nl - 1) / nl;
    const int k0 = (t * G < P.nchunk) ? t * G : P.nchunk;
    const int k1 = (k0 + G < P.nchunk) ? k0 + G : P.nchunk;
    const cplx one = cmake(1.0, 0.0), zero = cmake(0.0, 0.0);
    {   // stage 1
        cplx m0 = one, m1 = zero, m2 = zero, m3 = one, d0 = zero, d1 = zero, d2 = zero, d3 = zero;
        if (!BACKWARD) {
            for (int k = k0; k < k1; k++) {
                const cplx *m = cmv + (size_t)k * 8;
                // D <- m' M + m D,  M <- m M
                const cplx e0 = m[4] * m0 + m[5] * m2 + m[0] * d0 + m[1] * d2;
                const cplx e1 = m[4] * m1 + m[5] * m3 + m[0] * d1 + m[1] * d3;
                const cplx e2 = m[6] * m0 + m[7] * m2 + m[2] * d0 + m[3] * d2;
                const cplx e3 = m[6] * m1 + m[7] * m3 + m[2] * d1 + m[3] * d3;
                d0 = e0; d1 = e1; d2 = e2; d3 = e3;
                const cplx f0 = m[0] * m0 + m[1] * m2, f1 = m[0] * m1 + m[1] * m3;
                const cplx f2 = m[2] * m0 + m[3] * m2, f3 = m[2] * m1 + m[3] * m3;
                m0 = f0; m1 = f1; m2 = f2; m3 = f3;
            }
        } else {
            for (int k = k1; k-- > k0;) {
                const cplx *m = cmv + (size_t)k * 8;
                const cplx f0 = m[0] * m0 + m[1] * m2, f1 = m[0] * m1 + m[1] * m3;
                const cplx f2 = m[2] * m0 + m[3] * m2, f3 = m[2] * m1 + m[3] * m3;
                m0 = f0; m1 = f1; m2 = f2; m3 = f3;
            }
        }
        cplx *o = gm + (size_t)t * 8;
        o[0] = m0; o[1] = m1; o[2] = m2; o[3] = m3; o[4] = d0; o[5] = d1; o[6] = d2; o[7] = d3;
    }
    FA_SYNC();
    double s, c;
    const double tb = P.T1 + P.eps * bc;            // e^{i lam tb}
    fa_sincos(lc.x * tb, &s, &c);
    const cplx ph = cmake(c, s) * exp(-lc.y * tb);
    if (t == 0) {   // stage 2
        if (!BACKWARD) {
            const double ta = P.T0 - P.eps * bc;        // e^{-i lam ta}
            fa_sincos(-lc.x * ta, &s, &c);
            cplx p1 = cmake(c, s) * exp(lc.y * ta), p2 = zero;
            cplx d1 = p1 * cmake(0.0, -ta), d2 = zero;
            for (int g = 0; g < nl; g++) {
                cplx *v = gv + (size_t)g * 4;
                v[0] = p1; v[1] = p2; v[2] = d1; v[3] = d2;
                const cplx *m = gm + (size_t)g * 8;
                const cplx n1 = m[4] * p1 + m[5] * p2 + m[0] * d1 + m[1] * d2;
                const cplx n2 = m[6] * p1 + m[7] * p2 + m[2] * d1 + m[3] * d2;
                d1 = n1; d2 = n2;
                const cplx t1 = m[0] * p1 + m[1] * p2, t2 = m[2] * p1 + m[3] * p2;
                p1 = t1; p2 = t2;
            }
            cplx *bnd = P.bnd + (size_t)e * (P.nchunk + 1) * 2;
            bnd[2 * P.nchunk] = p1; bnd[2 * P.nchunk + 1] = p2;
            const cplx av = p1 * ph;
            P.a[e] = av;
            P.aprime[e] = (d1 * ph + cmake(0.0, tb) * av) * P.lscale;
        } else {
            cplx s1 = zero, s2 = ph;
            for (int g = nl; g-- > 0;) {
                cplx *v = gv + (size_t)g * 4;
                v[0] = s1; v[1] = s2;
                const cplx *m = gm + (size_t)g * 8;
                const cplx t1 = m[0] * s1 + m[1] * s2, t2 = m[2] * s1 + m[3] * s2;
                s1 = t1; s2 = t2;
            }
            cplx *bnd = P.bndp + (size_t)e * (P.nchunk + 1) * 2;
            bnd[0] = s1; bnd[1] = s2;
        }
    }
    FA_SYNC();
    // stage 3
    const cplx *v = gv + (size_t)t * 4;
    if (!BACKWARD) {
        cplx p1 = v[0], p2 = v[1];
        cplx *bnd = P.bnd + (size_t)e * (P.nchunk + 1) * 2;
        for (int k = k0; k < k1; k++) {
            bnd[2 * k] = p1; bnd[2 * k + 1] = p2;
            const cplx *m = cmv + (size_t)k * 8;
            const cplx t1 = m[0] * p1 + m[1] * p2, t2 = m[2] * p1 + m[3] * p2;
            p1 = t1; p2 = t2;
        }
    } else {
        cplx s1 = v[0], s2 = v[1];
        cplx *bnd = P.bndp + (size_t)e * (P.nchunk + 1) * 2;
        for (int k = k1; k-- > k0;) {
            bnd[2 * (k + 1)] = s1; bnd[2 * (k + 1) + 1] = s2;
            const cplx *m = cmv + (size_t)k * 8;
            const cplx t1 = m[0] * s1 + m[1] * s2, t2 = m[2] * s1 + m[3] * s2;
            s1 = t1; s2 = t2;
        }
    }
}

// fnft__nse_scatter_matrix (src/private/fnft__akns_scatter_matrix.c, BO): the whole scattering matrix S = U_{D-1} ...
// U_0 and its derivative with respect to lambda from the chunk maps {M, M'}; one workgroup per lambda, every lane
// composes its run of chunks, lane 0 the 256 run maps.
FA_DEV void body_bs_matrix(const BsParams &P)
{
    FA_LDS_DECL
    cplx *gm = (cplx *)FA_LDS_PTR;             // lanes x 8
    const int e = FA_BID, t = FA_TID, nl = FA_BDIM;
    const cplx *cmv = P.cm + (size_t)e * P.nchunk * 8;
    const int G = (P.nchunk + nl - 1) / nl;
    const int k0 = (t * G < P.nchunk) ? t * G : P.nchunk;
    const int k1 = (k0 + G < P.nchunk) ? k0 + G : P.nchunk;
    const cplx one = cmake(1.0, 0.0), zero = cmake(0.0, 0.0);
    cplx m0 = one, m1 = zero, m2 = zero, m3 = one, d0 = zero, d1 = zero, d2 = zero, d3 = zero;
    auto apply = [&](const cplx *m) {   // (M, D) <- (m M, m' M + m D)
        const cplx e0 = m[4] * m0 + m[5] * m2 + m[0] * d0 + m[1] * d2;
        const cplx e1 = m[4] * m1 + m[5] * m3 + m[0] * d1 + m[1] * d3;
        const cplx e2 = m[6] * m0 + m[7] * m2 + m[2] * d0 + m[3] * d2;
        const cplx e3 = m[6] * m1 + m[7] * m3 + m[2] * d1 + m[3] * d3;
        d0 = e0; d1 = e1; d2 = e2; d3 = e3;
        const cplx f0 = m[0] * m0 + m[1] * m2, f1 = m[0] * m1 + m[1] * m3;
        const cplx f2 = m[2] * m0 + m[3] * m2, f3 = m[2] * m1 + m[3] * m3;
        m0 = f0; m1 = f1; m2 = f2; m3 = f3;
    };
    for (int k = k0; k < k1; k++) apply(cmv + (size_t)k * 8);
    cplx *o = gm + (size_t)t * 8;
    o[0] = m0; o[1] = m1; o[2] = m2; o[3] = m3; o[4] = d0; o[5] = d1; o[6] = d2; o[7] = d3;
    FA_SYNC();
    if (t != 0) return;
    m0 = one; m1 = zero; m2 = zero; m3 = one; d0 = zero; d1 = zero; d2 = zero; d3 = zero;
    for (int g = 0; g < nl; g++) apply(gm + (size_t)g * 8);
    const int w = P.with_deriv ? 8 : 4;
    cplx *res = P.smat + (size_t)e * w;
    res[0] = m0; res[1] = m1; res[2] = m2; res[3] = m3;
    if (P.with_deriv) {
        const double s = P.lscale;       // chain rule of the scaled spectral parameter (1 for BO)
        res[4] = d0 * s; res[5] = d1 * s; res[6] = d2 * s; res[7] = d3 * s;
    }
}

// phi at every grid point of the chunk (grid point g+1 follows sample ups*(g+1)-1)
FA_DEV void body_bs_phi(const BsParams &P)
{
    const int c = FA_BID * FA_BDIM + FA_TID, e = FA_BID_Y;
    if (c >= P.nchunk) return;
    const cplx l = P.lam[e] * P.lscale;
    const long long n0 = (long long)c * P.L;
    const long long n1 = (n0 + P.L < P.D) ? n0 + P.L : P.D;
    const long long Dg = P.D / P.ups;
    const cplx *bnd = P.bnd + ((size_t)e * (P.nchunk + 1) + c) * 2;
    cplx p1 = bnd[0], p2 = bnd[1];
    cplx *PHI = P.PHI + (size_t)e * (Dg + 1) * 2;
    if (c == 0) { PHI[0] = p1; PHI[1] = p2; }
    BsStep U, V;
    for (long long n = n0; n < n1; n++) {
        bs_step<false>(P.q[n], l, P.eps, U, V);
        const cplx t1 = U.u00 * p1 + U.u01 * p2, t2 = U.u10 * p1 + U.u11 * p2;
        p1 = t1; p2 = t2;
        if ((n + 1) % P.ups == 0) {
            const long long g = (n + 1) / P.ups;
            PHI[2 * g] = p1; PHI[2 * g + 1] = p2;
        }
    }
}

// psi at every grid point of the chunk, backwards from the chunk's end vector (eigenfunctions of the inverse
// transform's Darboux step, src/fnft_nsev_inverse.c:963-1004); grid point g precedes sample ups*g
FA_DEV void body_bs_psi(const BsParams &P)
{
    const int c = FA_BID * FA_BDIM + FA_TID, e = FA_BID_Y;
    if (c >= P.nchunk) return;
    const cplx l = P.lam[e] * P.lscale;
    const long long n0 = (long long)c * P.L;
    const long long n1 = (n0 + P.L < P.D) ? n0 + P.L : P.D;
    const long long Dg = P.D / P.ups;
    const cplx *bnd = P.bndp + ((size_t)e * (P.nchunk + 1) + (c + 1)) * 2;
    cplx s1 = bnd[0], s2 = bnd[1];
    cplx *PSI = P.PSI + (size_t)e * (Dg + 1) * 2;
    if (n1 == P.D) { PSI[2 * Dg] = s1; PSI[2 * Dg + 1] = s2; }
    BsStep U, V;
    for (long long n = n1; n-- > n0;) {
        bs_step<false>(P.q[n], l, -P.eps, U, V);
        const cplx t1 = U.u00 * s1 + U.u01 * s2, t2 = U.u10 * s1 + U.u11 * s2;
        s1 = t1; s2 = t2;
        if (n % P.ups == 0) {
            const long long g = n / P.ups;
            PSI[2 * g] = s1; PSI[2 * g + 1] = s2;
        }
    }
}

FA_DEV double bs_metric(cplx p1, cplx p2, cplx s1, cplx s2)
{   // |0.5 log |(phi2/psi2)/(phi1/psi1)||, :644
    const cplx r = c_div(c_div(p2, s2), c_div(p1, s1));
    return fabs(0.5 * log(sqrt(cnorm2(r))));
}

// psi backwards through the chunk; the chunk owns its grid points except its first one (chunk 0
// owns point 0 too); best point of the chunk by the metric, first one on ties in ascending order
FA_DEV void body_bs_metric(const BsParams &P)
{
    const int c = FA_BID * FA_BDIM + FA_TID, e = FA_BID_Y;
    if (c >= P.nchunk) return;
    const cplx l = P.lam[e] * P.lscale;
    const long long n0 = (long long)c * P.L;
    const long long n1 = (n0 + P.L < P.D) ? n0 + P.L : P.D;
    const long long Dg = P.D / P.ups;
    const cplx *bnd = P.bndp + ((size_t)e * (P.nchunk + 1) + (c + 1)) * 2;
    cplx s1 = bnd[0], s2 = bnd[1];
    const cplx *PHI = P.PHI + (size_t)e * (Dg + 1) * 2;
    double best = 1.0e308 * 10.0;   // +inf
    cplx bval = cmake(0.0, 0.0);
    {
        const long long g = n1 / P.ups;
        const double m = bs_metric(PHI[2 * g], PHI[2 * g + 1], s1, s2);
        if (m <= best) { best = m; bval = c_div(PHI[2 * g], s1); }
    }
    BsStep U, V;
    for (long long n = n1; n-- > n0;) {
        bs_step<false>(P.q[n], l, -P.eps, U, V);
        const cplx t1 = U.u00 * s1 + U.u01 * s2, t2 = U.u10 * s1 + U.u11 * s2;
        s1 = t1; s2 = t2;
        if (n % P.ups == 0) {
            const long long g = n / P.ups;
            if (n > n0 || c == 0) {
                const double m = bs_metric(PHI[2 * g], PHI[2 * g + 1], s1, s2);
                if (m <= best) { best = m; bval = c_div(PHI[2 * g], s1); }
            }
        }
    }
    cplx *o = P.best + ((size_t)e * P.nchunk + c) * 2;
    o[0] = cmake(best, 0.0);
    o[1] = bval;
}

// one workgroup per eigenvalue: the chunk with the smallest metric, the first one on ties
FA_DEV void body_bs_pick(const BsParams &P)
{
    FA_LDS_DECL
    double *lm = (double *)FA_LDS_PTR;          // lanes: best metric
    int *li = (int *)(lm + FA_BDIM);            // lanes: its chunk
    const int e = FA_BID, t = FA_TID, nl = FA_BDIM;
    const cplx *bv = P.best + (size_t)e * P.nchunk * 2;
    double best = 1.0e308 * 10.0;
    int bi = 0x7fffffff;
    for (int c = t; c < P.nchunk; c += nl)
        if (bv[2 * c].x < best) { best = bv[2 * c].x; bi = c; }
    lm[t] = best;
    li[t] = bi;
    FA_SYNC();
    for (int h = nl / 2; h >= 1; h >>= 1) {
        if (t < h) {
            const double mo = lm[t + h];
            const int io = li[t + h];
            if (mo < lm[t] || (mo == lm[t] && io < li[t])) { lm[t] = mo; li[t] = io; }
        }
        FA_SYNC();
    }
    if (t == 0) P.b[e] = (li[0] != 0x7fffffff) ? bv[2 * li[0] + 1] : cmake(0.0, 0.0);
}

// ---------------------------------------------------------------------------------------------
// Roots of a polynomial on an arc of the unit circle by grid search (src/private/fnft__poly_roots_fftgridsearch.c):
// the polynomial is evaluated by chirp z-transforms (three rings, or one for the para-Hermitian form); these
// kernels mark the grid points that hold a root estimate and then compact the estimates IN GRID ORDER (block counts,
// host prefix over the blocks, ordered scatter) -- the reference returns them in that order.
// ---------------------------------------------------------------------------------------------
struct GridSearchParams {
    const cplx *vals;      // 3*M (rings k = -1, 0, 1) or M (para-Hermitian)
    long long M;
    double phi0, eps;
    long long N1;          // para-Hermitian: N - 1 = deg/2 (phase factor exp(-i phi (N-1)))
    int *keep;             // M flags
    cplx *cand;            // M candidates
    int *blockcnt;         // ceil(M/256) counts
    const int *blockoff;   // exclusive prefix of blockcnt
    cplx *out;             // compacted estimates
    int *status;           // bit 1: division by zero (:117-120)
};
// :77-146: nine-point minimum test, least-squares linear fit, root of the fit
FA_DEV void body_gridsearch_mark(const GridSearchParams &P)
{
    const long long i = (long long)FA_BID * FA_BDIM + FA_TID;
    if (i >= P.M) return;
    int keep = 0;
    cplx zr = cmake(0.0, 0.0);
    const long long M = P.M;
    if (i >= 1 && i < M - 1) {
        const double c0 = cnorm2(P.vals[M + i]);
        bool is_min = true;
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int dj = -1; dj <= 1; dj++)
                if (!(k == 1 && dj == 0) && c0 > cnorm2(P.vals[(long long)k * M + i + dj])) is_min = false;
        if (is_min) {
            double s0, c0s;
            fa_sincos(P.phi0 + (double)i * P.eps, &s0, &c0s);
            const cplx z0 = cmake(c0s, s0);
            const cplx y0 = P.vals[M + i];
            cplx c = cmake(0.0, 0.0);
            double den = 0.0;
            for (long long j = i - 1; j < i + 2; j++)
                for (int k = -1; k < 2; k++) {
                    if (j == 0 && k == 0) continue;   // as in the reference (:99-100)
                    double sj, cj;
                    fa_sincos(P.phi0 + (double)j * P.eps, &sj, &cj);
                    const cplx zi = cmake(cj, sj) * (1.0 - (double)k * P.eps);
                    const cplx yi = P.vals[(long long)(k + 1) * M + j];
                    c = c + cconj(zi - z0) * (yi - y0);
                    den += cnorm2(zi - z0);
                }
            if (den == 0.0) fa_atomic_or_i32(P.status, 2);
            else {
                c = c * (1.0 / den);
                if (c.x == 0.0 && c.y == 0.0) {
                    if (y0.x == 0.0 && y0.y == 0.0) { keep = 1; zr = z0; }
                } else {
                    zr = z0 - c_div(y0, c);
                    if (!(sqrt(cnorm2(zr - z0)) > P.eps)) keep = 1;
                }
            }
        }
    }
    P.keep[i] = keep;
    P.cand[i] = zr;
}
// para-Hermitian form (:187-214): sign change of the real part between consecutive grid points
FA_DEV void body_gridsearch_mark_ph(const GridSearchParams &P)
{
    const long long i = (long long)FA_BID * FA_BDIM + FA_TID;
    if (i >= P.M) return;
    int keep = 0;
    cplx zr = cmake(0.0, 0.0);
    if (i >= 1) {
        auto val = [&](long long m) {
            const double phi = P.phi0 + P.eps * (double)m;
            double sn, cs;
            fa_sincos(-phi * (double)P.N1, &sn, &cs);
            return P.vals[m] * cmake(cs, sn);
        };
        const cplx r0 = val(i - 1), r1 = val(i);
        if (r0.x * r1.x <= 0.0) {
            const double phi1 = P.phi0 + P.eps * (double)(i - 1), phi2 = phi1 + P.eps;
            double phi;
            if (r0.x != r1.x || r0.y != r1.y) phi = phi1 - c_div(r0 * (phi2 - phi1), r1 - r0).x;
            else phi = 0.5 * (phi1 + phi2);
            double sn, cs;
            fa_sincos(phi, &sn, &cs);
            zr = cmake(cs, sn);
            keep = 1;
        }
    }
    P.keep[i] = keep;
    P.cand[i] = zr;
}
FA_DEV void body_compact_count(const GridSearchParams &P)
{
    FA_LDS_DECL
    int *cnt = (int *)FA_LDS_PTR;
    const long long i = (long long)FA_BID * FA_BDIM + FA_TID;
    cnt[FA_TID] = (i < P.M) ? P.keep[i] : 0;
    FA_SYNC();
    for (int h = FA_BDIM / 2; h >= 1; h >>= 1) {
        if (FA_TID < h) cnt[FA_TID] += cnt[FA_TID + h];
        FA_SYNC();
    }
    if (FA_TID == 0) P.blockcnt[FA_BID] = cnt[0];
}
FA_DEV void body_compact_scatter(const GridSearchParams &P)
{
    FA_LDS_DECL
    int *rank = (int *)FA_LDS_PTR;
    const long long i = (long long)FA_BID * FA_BDIM + FA_TID;
    const int k = (i < P.M) ? P.keep[i] : 0;
    rank[FA_TID] = k;
    FA_SYNC();
    for (int off = 1; off < FA_BDIM; off <<= 1) {    // inclusive scan
        const int add = (FA_TID >= off) ? rank[FA_TID - off] : 0;
        FA_SYNC();
        rank[FA_TID] += add;
        FA_SYNC();
    }
    if (k) P.out[(long long)P.blockoff[FA_BID] + rank[FA_TID] - 1] = P.cand[i];
}

// ---------------------------------------------------------------------------------------------
// fnft_nsev_inverse (src/fnft_nsev_inverse.c): element-wise stages between the DFTs (chirp kernels in DFT mode) of
// the continuous part, and the Darboux steps of the discrete part.  One kernel, selected by `op`.
// ---------------------------------------------------------------------------------------------
enum InvOpCode {
    INV_PREP = 0,      // :1013-1033 + :251-296  contspec *= Blaschke factors * exp(-i xi pf) (in place) -> FFT order
    INV_PAD,           // out[i] = i <= i0 ? a[i] : 0
    INV_SPEC_X,        // fnft__poly_specfact.c:75-108  x = log|P|, 0.5 log(1 + |P|^2), 0.5 log(1 - |P|^2)
    INV_HILBERT,       // :116-121
    INV_SPEC_RESP,     // :129-130  exp(x - i y)/M
    INV_REV_CONJ,      // :135-136  out[i] = conj(a[i0 - i]), i <= i0
    INV_ITER_FIN,      // :434-438  q / sqrt(1 + kappa |q|^2) / D
    INV_REVERSE,       // out[i] = a[n-1-i]
    INV_ITER_PHASE,    // :462-469  phase = arg(a[i]); out[i] = reordered(b)[i] * exp(i phase); block sums of |phase|
    INV_BTAU,          // :654-657  b = 2 eps contspec / degree1step, end points halved
    INV_DOUBLE_Q,      // q'[2m] = q[m], q'[2m+1] = q[m+1]: the half steps of compute_eigenfunctions as a signal
};
struct InvOpParams {
    int op;
    long long n;             // elements
    const cplx *a, *b;
    cplx *out, *out2;
    double s0, s1, s2;       // op-specific scalars
    long long i0;            // op-specific index
    int kappa;
    int K;
    const cplx *bs;          // K bound states (INV_PREP)
    int *status;             // bit 3: ill-posed spectral factorization (a warning)
    double *accum;           // INV_ITER_PHASE: one partial sum per workgroup
};
FA_DEV void body_inv_op(const InvOpParams &P)
{
    FA_LDS_DECL
    double *red = (double *)FA_LDS_PTR;
    const long long i = (long long)FA_BID * FA_BDIM + FA_TID;
    const bool act = i < P.n;
    double mine = 0.0;
    if (act) {
        switch (P.op) {
        case INV_PREP: {
            const double xi = P.s0 + (double)i * P.s1;
            cplx c = P.a[i];
            for (int k = 0; k < P.K; k++) {
                const cplx bk = P.bs[k];
                c = c * c_div(cmake(xi - bk.x, -bk.y), cmake(xi - bk.x, bk.y));
            }
            double sn, cs;
            fa_sincos(-xi * P.s2, &sn, &cs);
            c = c * cmake(cs, sn);
            P.out[i] = c;
            const long long h = P.n / 2;
            P.out2[(i >= h - 1) ? i - (h - 1) : i + (h + 1)] = c;
        } break;
        case INV_PAD: P.out[i] = (i <= P.i0) ? P.a[i] : cmake(0.0, 0.0); break;
        case INV_SPEC_X: {
            const double tol = 1.4901161193847656e-08;   // sqrt(eps)
            const double a2 = cnorm2(P.a[i]);
            cplx x;
            if (P.kappa == 0) {
                const double ab = sqrt(a2);
                if (ab < tol) fa_atomic_or_i32(P.status, 8);
                x = cmake(log(ab), 0.0);
            } else if (P.kappa == -1) {
                x = cmake(0.5 * log(1.0 + a2), 0.0);
            } else {
                if (a2 > 1.0 - tol) fa_atomic_or_i32(P.status, 8);
                const double v = 1.0 - a2;   // 0.5*clog(v): log|v| + i*pi for v < 0
                x = cmake(0.5 * log(fabs(v)), v < 0.0 ? 0.5 * 3.14159265358979323846 : 0.0);
            }
            P.out[i] = x;
        } break;
        case INV_HILBERT: {
            const long long h = P.n / 2;
            const cplx v = P.a[i];
            const double m = 1.0 / (double)P.n;
            cplx r;
            if (i == 0 || i == h - 1) r = cmake(0.0, 0.0);
            else if (i < h - 1) r = cmake(v.y * m, -v.x * m);      // * (-i/M)
            else r = cmake(-v.y * m, v.x * m);                     // * (+i/M)
            P.out[i] = r;
        } break;
        case INV_SPEC_RESP: {
            const cplx x = P.a[i], y = P.b[i];
            // exp(x - i*y) / M
            const double re = x.x + y.y, im = x.y - y.x;
            double sn, cs;
            fa_sincos(im, &sn, &cs);
            const double m = exp(re) / (double)P.n;
            P.out[i] = cmake(m * cs, m * sn);
        } break;
        case INV_REV_CONJ: P.out[i] = cconj(P.a[P.i0 - i]); break;
        case INV_ITER_FIN: {
            const cplx q = P.a[i];
            const double d = 1.0 / (sqrt(1.0 + (double)P.kappa * cnorm2(q)) * (double)P.n);
            P.out[i] = q * d;
        } break;
        case INV_REVERSE: P.out[i] = P.a[P.n - 1 - i]; break;
        case INV_ITER_PHASE: {
            const cplx v = P.a[i];
            const double ph = atan2(v.y, v.x);
            mine = fabs(ph);
            const long long h = P.n / 2;
            const cplx c = P.b[(i <= h) ? i + (h - 1) : i - (h + 1)];
            double sn, cs;
            fa_sincos(ph, &sn, &cs);
            P.out[i] = c * cmake(cs, sn);
        } break;
        case INV_BTAU: {
            const double f = (i == 0 || i == P.n - 1) ? P.s0 : 2.0 * P.s0;
            P.out[i] = P.a[i] * f;
        } break;
        case INV_DOUBLE_Q: P.out[i] = P.a[(i + 1) / 2]; break;
        default: break;
        }
    }
    if (P.op == INV_ITER_PHASE) {   // deterministic block sums (the host adds them in order)
        red[FA_TID] = mine;
        FA_SYNC();
        for (int h = FA_BDIM / 2; h >= 1; h >>= 1) {
            if (FA_TID < h) red[FA_TID] += red[FA_TID + h];
            FA_SYNC();
        }
        if (FA_TID == 0) P.accum[FA_BID] = red[0];
    }
}

// ---- layer peeling on the device (src/private/fnft__nse_finvscatter.c:66-232) ------------------------------------
// Two factors of a 2x2 polynomial product from arbitrary strided device arrays into level 0 of a plan with n = 2
// matrices of degree d (body/tail layout); the product comes back out through body_export_tm (strided, times 2^W).
struct PeelIoParams {
    const cplx *A, *B;        // four entries of d+1 coefficients each, highest power first
    long long As, Bs;         // entry strides
    long long d;
    cplx *body, *tail;        // import: level 0 of the plan (plane = 2*d)
    double *scale;
    int *wexp;
};
FA_DEV void body_peel_import(const PeelIoParams &P)
{
    const long long i = (long long)FA_BID * FA_BDIM + FA_TID;
    const long long w = P.d + 1;
    if (i >= 8 * w) return;
    const int e = (int)(i / (2 * w));
    const long long r = i % (2 * w);
    const int m = (int)(r / w);
    const long long k = r % w;
    const cplx v = m == 0 ? P.A[(long long)e * P.As + k] : P.B[(long long)e * P.Bs + k];
    if (k < P.d) P.body[(long long)e * 2 * P.d + (long long)m * P.d + k] = v;
    else P.tail[e * 2 + m] = v;
    if (i < 2) { P.scale[i] = 1.0; P.wexp[i] = 0; }
}
// One 2x2 polynomial product of the layer peeling in ONE launch, for the degrees where a product is a single
// workgroup's work anyway (deg = N/2 = 256, 512, 1024): the four entries of a factor are transformed CONCURRENTLY
// (four groups of N/R lanes, fft_wg with B = 4) instead of one after the other, factors are read from and the result
// is written to the caller's strided arrays (no import / export launches), cyclic length N = 2 deg with the aliased
// top coefficient formed from the constant terms (as the tree does).  C = A * B, entries [11|12|21|22].
struct PeelProdParams {
    const cplx *A, *B;        // four entries of deg+1 coefficients, highest power first
    long long As, Bs;
    cplx *C;                  // four entries of 2 deg + 1 coefficients
    long long Cs;
    const cplx *tw;           // exp(-2 pi i j / N), j < N
};
template <int N, int R> FA_DEV void body_peel_product(const PeelProdParams &P)
{
    constexpr int DEG = N / 2;
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;     // N*4: transform buffer, then the spectra of one factor [bin][entry]
    const int tid = FA_TID;
    const int g = tid % 4, v = tid / 4; // entry of this lane's group, lane inside the group
    const int r = g >> 1, cc = g & 1;   // row and column of the result entry this group forms
    cplx a[R], b[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int n = v + (N / R) * i;
        a[i] = (n <= DEG) ? P.A[(long long)g * P.As + n] : cmake(0.0, 0.0);   // deg + 1 coefficients, zero-padded to N
        b[i] = (n <= DEG) ? P.B[(long long)g * P.Bs + n] : cmake(0.0, 0.0);
    }
    // constant terms of the row of A and the column of B this group needs: the aliased coefficient
    const cplx tp = P.A[(long long)(2 * r) * P.As + DEG] * P.B[(long long)cc * P.Bs + DEG]
                  + P.A[(long long)(2 * r + 1) * P.As + DEG] * P.B[(long long)(2 + cc) * P.Bs + DEG];
    int parity = 0;
    fft_wg<N, R, 4, -1, false, true>(a, lds, v, g, P.tw, parity);
    fft_wg<N, R, 4, -1, false, true>(b, lds, v, g, P.tw, parity);
    // spectra of A through LDS: every group picks up its row
    FA_SYNC_LDS();
#pragma unroll
    for (int i = 0; i < R; i++) lds[(size_t)(v + (N / R) * i) * 4 + g] = a[i];
    FA_SYNC_LDS();
    cplx p[R], q[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
        p[i] = lds[(size_t)(v + (N / R) * i) * 4 + 2 * r];
        q[i] = lds[(size_t)(v + (N / R) * i) * 4 + 2 * r + 1];
    }
    FA_SYNC_LDS();
#pragma unroll
    for (int i = 0; i < R; i++) lds[(size_t)(v + (N / R) * i) * 4 + g] = b[i];
    FA_SYNC_LDS();
#pragma unroll
    for (int i = 0; i < R; i++) {
        const cplx b1 = lds[(size_t)(v + (N / R) * i) * 4 + cc], b2 = lds[(size_t)(v + (N / R) * i) * 4 + 2 + cc];
        a[i] = p[i] * b1 + q[i] * b2;
    }
    FA_SYNC_LDS();
    fft_wg<N, R, 4, +1, false, true>(a, lds, v, g, P.tw, parity);
    const double inv = 1.0 / (double)N;
    cplx *dst = P.C + (long long)g * P.Cs;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int n = v + (N / R) * i;
        cplx val = a[i] * inv;
        if (n == 0) {
            val = val - tp;          // coefficient 2 deg folded onto coefficient 0
            dst[N] = tp;
        }
        dst[n] = val;
    }
}

// One block of d <= 256 samples peeled off by one workgroup of three waves, coefficient arrays in registers (4
// consecutive indices per lane), one sample per step: with Q = -kappa conj(T21(0)/T11(0)) (:158-176) the last step
// matrix is divided out of the first column,
//   T11 <- scl (T11 - Q T21),   T21 <- scl (kappa Q* T11 + T21) / z,          scl = 1/sqrt(1 + kappa |Q|^2)
// (wave 0; the second column of T never influences the samples), and the same matrices are multiplied onto the
// block's inverse (up to a power of z), which the caller one level up needs (:144-156): waves 1 and 2, one column
// each, one step behind wave 0 (they take every Q from LDS as soon as wave 0 has published it; the scalar factors scl
// of all steps are multiplied on at the end).  In exact arithmetic this is what the reference's recursion computes
// for the block.
struct PeelLeafParams {
    const cplx *T;            // four entries of d+1 coefficients at stride Ts (highest power first)
    long long Ts;
    int d;
    cplx *Ti;                 // NULL or four entries of d+1 coefficients at stride Tis
    long long Tis;
    cplx *q;                  // d samples
    double eps_t;
    int kappa, modal;
    int *status;              // bit 4: a reconstructed sample violates 1 + kappa |eps q|^2 > 0 (:173-176); bit 5: internal
};
FA_DEV cplx fa_shfl_c(cplx v, int src) { return cmake(fa_shfl(v.x, src), fa_shfl(v.y, src)); }
FA_DEV cplx fa_readlane_c(cplx v, int src) { return cmake(fa_readlane(v.x, src), fa_readlane(v.y, src)); }
FA_DEV cplx fa_shfl_up_c(cplx v) { return cmake(fa_shfl_up1(v.x), fa_shfl_up1(v.y)); }
FA_DEV cplx fa_shfl_down_c(cplx v) { return cmake(fa_shfl_down1(v.x), fa_shfl_down1(v.y)); }
FA_DEV cplx fa_shfl_down_cz(cplx v) { return cmake(fa_shfl_down1_z(v.x), fa_shfl_down1_z(v.y)); }
// diagnostic builds (tests/gpu_debug/build_variant.py): 1 the inverse's waves skip their steps, 2 wave 0 skips its steps
// (results are wrong in both; they time one side of the kernel alone)
#ifndef FA_PEEL_DIAG
#define FA_PEEL_DIAG 0
#endif
FA_DEV void body_peel_leaf(const PeelLeafParams &P)
{
    constexpr int R = 4;
    FA_LDS_DECL
    cplx *Qs = (cplx *)FA_LDS_PTR;               // 256 step parameters Q
    double *Stot = (double *)(Qs + 256);         // product of the 256 scale factors
    int *ready = (int *)(Stot + 1);              // number of steps wave 0 has published
    const int wave = FA_TID / 64, lane = FA_TID % 64;
    const int d = P.d;
    const cplx zero = cmake(0.0, 0.0);
    if (FA_TID == 0) fa_lds_publish(ready, 0);
    FA_SYNC();
    // element K (1 <= K <= d) of an array lives in lane (K-1)/R, slot (K-1)%R; element 0 in a register of its own
    if (wave == 0) {
        cplx t1[R], t2[R];
        const cplx t10 = P.T[0], t20 = P.T[2 * P.Ts];
#pragma unroll
        for (int s = 0; s < R; s++) {
            const int K = lane * R + s + 1;
            t1[s] = (K <= d) ? P.T[K] : zero;
            t2[s] = (K <= d) ? P.T[2 * P.Ts + K] : zero;
        }
        const int lastLane = (d - 1) / R, lastSlot = (d - 1) % R;
        // Q = -kappa conj(c21 / c11) = -kappa c11 conj(c21) / |c11|^2 from the constant terms (reciprocal by v_rcp_f64 +
        // two Newton steps); every lane forms the same value
        auto q_of = [&](cplx c11, cplx c21) -> cplx {
            return (c11 * cconj(c21)) * ((double)(-P.kappa) * aberth_rcp(cnorm2(c11)));
        };
        auto consts_of = [&](const cplx (&x1)[R], const cplx (&x2)[R], bool slot3, cplx &c11, cplx &c21) {
            c11 = x1[R - 1]; c21 = x2[R - 1];                // constant terms T11[d], T21[d]
            if (!slot3) {
#pragma unroll
                for (int s = 0; s < R; s++)
                    if (s == lastSlot) { c11 = x1[s]; c21 = x2[s]; }
            }
            c11 = fa_readlane_c(c11, lastLane);
            c21 = fa_readlane_c(c21, lastLane);
        };
        // One step with its Q already known.  The dependent chain of the block runs Q_k -> constant terms of the
        // updated arrays (one slot: two fused multiply-adds deep) -> v_readlane -> reciprocal -> Q_{k+1}; the other slots
        // of step k are independent of it and fill its latencies: the step is written in that order (SLOT3: d is a
        // multiple of 4, the constant terms sit in slot 3, which needs no neighbour lane; FIRST: element 0 (t10, t20) is
        // the left neighbour of lane 0, only step 0 ever reads it).  Without the factor scl: Q only sees the ratio
        // T21/T11, and the scale factors are multiplied onto the inverse at the end.
        auto step_fn = [&](int step, cplx &Q, bool slot3, bool first) {
            if (lane == 0) {
                Qs[step] = Q;
                fa_lds_publish_inorder(ready, step + 1);    // the inverse's waves follow behind
            }
            const cplx mQ = cmake(-Q.x, -Q.y), kQc = cconj(Q) * (double)P.kappa;
            cplx n1[R], n2[R];
            if (slot3) {
                // the chain, one link per line, each followed by one independent complex multiply-add of the other
                // slots; the fences keep that order in the instruction stream
                static_assert(R == 4, "step_fn: written for four coefficients per lane");
                n1[3] = cfma(mQ, t2[3], t1[3]);
                n2[3] = cfma(kQc, t1[2], t2[2]);
                cplx l1 = fa_shfl_up_c(t1[R - 1]), l2 = fa_shfl_up_c(t2[R - 1]);
                if (first && lane == 0) { l1 = t10; l2 = t20; }
                fa_sched_fence();
                const cplx c11 = fa_readlane_c(n1[3], lastLane), c21 = fa_readlane_c(n2[3], lastLane);
                fa_sched_fence();
                double m = c11.y * c11.y;                    n1[2] = cfma(mQ, t2[2], t1[2]);
                fa_sched_fence();
                m = fma(c11.x, c11.x, m);                    n2[2] = cfma(kQc, t1[1], t2[1]);
                fa_sched_fence();
                double r = fa_rcp_approx(m);                 n1[1] = cfma(mQ, t2[1], t1[1]);
                fa_sched_fence();
                double e = fma(-m, r, 1.0);                  n2[1] = cfma(kQc, t1[0], t2[0]);
                fa_sched_fence();
                r = fma(e, r, r);                            const cplx num = c11 * cconj(c21);
                fa_sched_fence();
                e = fma(-m, r, 1.0);                         n1[0] = cfma(mQ, t2[0], t1[0]);
                fa_sched_fence();
                r = fma(e, r, r);                            n2[0] = cfma(kQc, l1, l2);
                fa_sched_fence();
                Q = num * ((double)(-P.kappa) * r);          // garbage after the last step (never published)
            } else {
                // row 1 element-wise, row 2 takes the left neighbour (division by z)
#pragma unroll
                for (int s = R - 1; s >= 1; s--) {
                    n1[s] = cfma(mQ, t2[s], t1[s]);
                    n2[s] = cfma(kQc, t1[s - 1], t2[s - 1]);
                }
                cplx l1 = fa_shfl_up_c(t1[R - 1]), l2 = fa_shfl_up_c(t2[R - 1]);
                if (first && lane == 0) { l1 = t10; l2 = t20; }
                n1[0] = cfma(mQ, t2[0], t1[0]);
                n2[0] = cfma(kQc, l1, l2);
                cplx c11, c21;
                consts_of(n1, n2, false, c11, c21);
                Q = q_of(c11, c21);
            }
#pragma unroll
            for (int s = 0; s < R; s++) { t1[s] = n1[s]; t2[s] = n2[s]; }
        };
        {
            cplx c11, c21;
            consts_of(t1, t2, lastSlot == R - 1, c11, c21);
            cplx Q = q_of(c11, c21);
            if (FA_PEEL_DIAG == 2) {
                if (lane == 0) fa_lds_publish_inorder(ready, d);
            } else if (lastSlot == R - 1) {
                // two steps per trip: the arrays change registers from step to step, the second step hands them back
                step_fn(0, Q, true, true);
                int step = 1;
                for (; step + 1 < d; step += 2) {
                    step_fn(step, Q, true, false);
                    step_fn(step + 1, Q, true, false);
                }
                if (step < d) step_fn(step, Q, true, false);
            } else {
                step_fn(0, Q, false, true);
                for (int step = 1; step < d; step++) step_fn(step, Q, false, false);
            }
        }
        // (lane 0's Qs are read by every lane of this wave below: LDS instructions of one wave execute in order)
        // samples and the product of the scale factors, all steps at once (off the dependent chain), :158-196
        double prod = 1.0;
        for (int step = lane; step < d; step += 64) {
            const cplx Q = Qs[step];
            const double aQ2 = cnorm2(Q);
            const double den = 1.0 + (double)P.kappa * aQ2;
            if (!(den > 0.0)) fa_atomic_or_i32(P.status, 16);
            prod *= 1.0 / sqrt(den);
            cplx qv;
            if (P.modal) qv = Q * (1.0 / P.eps_t);
            else {
                const double aQ = sqrt(aQ2);
                const double f = (aQ > 0.0) ? atan(aQ) / (aQ * P.eps_t) : 1.0 / P.eps_t;   // atan|Q| e^{i arg Q} / eps
                qv = Q * f;
            }
            P.q[d - 1 - step] = qv;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) prod *= fa_shfl(prod, lane ^ off);
        if (lane == 0) Stot[0] = prod;
        FA_SYNC();
        return;
    }
    // inverse, column j = wave - 1: entries (0, 2) or (1, 3); starts as the identity (constant terms); every step
    // matrix without its factor scl, as soon as wave 0 has published its Q
    const int j = wave - 1;
    cplx a1[R], a2[R], a10 = zero, a20 = zero;
#pragma unroll
    for (int s = 0; s < R; s++) {
        const int K = lane * R + s + 1;
        a1[s] = (K == d && j == 0) ? cmake(1.0, 0.0) : zero;
        a2[s] = (K == d && j == 1) ? cmake(1.0, 0.0) : zero;
    }
    const bool want = P.Ti != nullptr;
    int have = 0;
    // wait until wave 0 has published step `upto` (bounded: wave 0 never waits for anybody, so `ready` reaches d; the
    // bound only guards the exit)
    auto wait_for = [&](int upto) -> bool {
        for (int spin = 0; have <= upto && spin < (1 << 20); spin++) {
            have = fa_lds_observe(ready);
            if (have <= upto) fa_nap();
        }
        return have > upto;
    };
    // four steps per visit of LDS: the counter is polled once and four Q are fetched at once, so the LDS round trips
    // are off the per-step path (the waves run up to four steps behind wave 0, which never waits for them)
    auto one_step = [&](cplx Q) {
        const cplx mQ = cmake(-Q.x, -Q.y), kQc = cconj(Q) * (double)P.kappa;
        // row 1 takes the right neighbour (multiplication by z), row 2 element-wise
        const cplx r1 = fa_shfl_down_cz(a1[0]), r2 = fa_shfl_down_cz(a2[0]);   // zero behind the last lane
        // element 0 is kept by lane 0 alone (the other lanes' copies are never read): new element 0 from element 1,
        // which is lane 0's slot 0
        const cplx n10 = cfma(mQ, a2[0], a1[0]);
        const cplx n20 = cfma(kQc, a10, a20);
        cplx m1[R], m2[R];
#pragma unroll
        for (int s = 0; s < R; s++) {
            const cplx x1 = (s == R - 1) ? r1 : a1[s + 1], x2 = (s == R - 1) ? r2 : a2[s + 1];
            m1[s] = cfma(mQ, x2, x1);
            m2[s] = cfma(kQc, a1[s], a2[s]);
        }
#pragma unroll
        for (int s = 0; s < R; s++) { a1[s] = m1[s]; a2[s] = m2[s]; }
        a10 = n10;
        a20 = n20;
    };
    bool ok = want && FA_PEEL_DIAG != 1;
    for (int step = 0; ok && step < d; step += 4) {
        const int last = (step + 3 < d) ? step + 3 : d - 1;
        if (!wait_for(last)) { if (lane == 0) fa_atomic_or_i32(P.status, 32); ok = false; break; }
        cplx Qb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) Qb[u] = Qs[(step + u < d) ? step + u : d - 1];
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (step + u < d) one_step(Qb[u]);
    }
    FA_SYNC();
    if (!want) return;
    const double S = Stot[0];
    cplx *o1 = P.Ti + (long long)j * P.Tis, *o2 = P.Ti + (long long)(2 + j) * P.Tis;
    if (lane == 0) { o1[0] = a10 * S; o2[0] = a20 * S; }
#pragma unroll
    for (int s = 0; s < R; s++) {
        const int K = lane * R + s + 1;
        if (K <= d) { o1[K] = a1[s] * S; o2[K] = a2[s] * S; }
    }
}

// Discrete part, src/fnft_nsev_inverse.c:680-903.  One lane per sample n.
struct InvDsParams {
    long long D;
    int K;
    const cplx *bs;          // K bound states, sorted by descending imaginary part
    const cplx *nc;          // K norming constants
    double T0, eps_t;
    long long zc;            // first sample with t >= 0
    cplx *q;                 // D samples: seed in (CDT), result out
    cplx *work;              // K*D (pure solitons: rho_k) or 2*K*D (CDT: S1, S2)
    const cplx *PHI, *PSI;   // K*D*2 each (CDT)
};
// pure multi-soliton, :797-842: q(t) by the recursive Darboux formula in rho_k = b_k exp(2 i lam_k t); samples
// before the zero crossing use the mirrored recursion (1/b_k, -t) and the conjugate
FA_DEV void body_inv_solitons(const InvDsParams &P)
{
    const long long n = (long long)FA_BID * FA_BDIM + FA_TID;
    if (n >= P.D) return;
    const double t = P.T0 + P.eps_t * (double)n;
    const bool right = n >= P.zc;
    const double sg = right ? 1.0 : -1.0;
    cplx *rho = P.work + n;   // rho_k at work[k*D + n]
    for (int k = 0; k < P.K; k++) {
        const cplx l = P.bs[k];
        const cplx b = right ? P.nc[k] : c_div(cmake(1.0, 0.0), P.nc[k]);
        // exp(sg * 2 i l t)
        double sn, cs;
        fa_sincos(sg * 2.0 * l.x * t, &sn, &cs);
        const double m = exp(-sg * 2.0 * l.y * t);
        rho[(size_t)k * P.D] = b * cmake(m * cs, m * sn);
    }
    cplx qt = cmake(0.0, 0.0);
    for (int i = 0; i < P.K; i++) {
        const cplx li = P.bs[i];
        const cplx r = rho[(size_t)i * P.D], rc = cconj(r);
        const cplx f = cmake(0.0, 2.0 * li.y) * (1.0 / (1.0 + cnorm2(r)));
        qt = qt + (rc * f) * cmake(0.0, 2.0);
        for (int j = i + 1; j < P.K; j++) {
            const cplx lj = P.bs[j];
            const cplx rj = rho[(size_t)j * P.D];
            const cplx num = (lj - li) * rj + (rj - r) * f;
            const cplx den = lj - cconj(li) - (cmake(1.0, 0.0) + rc * rj) * f;
            rho[(size_t)j * P.D] = c_div(num, den);
        }
    }
    P.q[n] = right ? qt : cconj(qt);
}
// Darboux steps on top of a seed potential, :864-889, from the eigenfunctions phi, psi of the seed at every sample
FA_DEV void body_inv_cdt(const InvDsParams &P)
{
    const long long n = (long long)FA_BID * FA_BDIM + FA_TID;
    if (n >= P.D) return;
    cplx *S1 = P.work + n, *S2 = P.work + (size_t)P.K * P.D + n;
    cplx qn = P.q[n];
    for (int i = 0; i < P.K; i++) {
        const cplx li = P.bs[i];
        const cplx *ph = P.PHI + ((size_t)i * P.D + n) * 2, *ps = P.PSI + ((size_t)i * P.D + n) * 2;
        cplx p1 = ph[0], p2 = ph[1], s1 = ps[0], s2 = ps[1];
        for (int j = 0; j < i; j++) {
            const cplx a = li - S1[(size_t)j * P.D], b = S2[(size_t)j * P.D], ac = li - cconj(S1[(size_t)j * P.D]);
            const cplx t1 = a * p1 - b * p2;
            p2 = cconj(b) * p1 + ac * p2;
            p1 = t1;
            const cplx t2 = a * s1 - b * s2;
            s2 = cconj(b) * s1 + ac * s2;
            s1 = t2;
        }
        const cplx nci = P.nc[i];
        const cplx beta = c_div(p1 - nci * s1, p2 - nci * s2);
        const double ab = cnorm2(beta);
        const double inv = 1.0 / (1.0 + ab);
        const cplx s1v = (li * ab + cconj(li)) * inv;
        const cplx s2v = (cmake(0.0, 2.0 * li.y) * beta) * inv;
        S1[(size_t)i * P.D] = s1v;
        S2[(size_t)i * P.D] = s2v;
        qn = qn - cmake(0.0, 2.0) * s2v;
    }
    P.q[n] = qn;
}

// ---------------------------------------------------------------------------------------------
// All roots of the a-polynomial: Ehrlich-Aberth iteration in place of the reference's structured
// QR (eiscor through fnft__poly_roots_fasteigen.c:29-48, Fortran).  Every root estimate is a lane:
// Newton correction p/p' by Horner in z (|z| <= 1) or in 1/z, then the Aberth sum over all others.
// ---------------------------------------------------------------------------------------------
struct AberthParams {
    const cplx *coef;    // n+1 coefficients, highest power first
    long long n;
    const cplx *z;       // n estimates of this sweep (read by every workgroup)
    cplx *z_out;         // the next sweep's estimates (double buffer: no workgroup reads what another one writes)
    int fast;            // sum kernel: single-precision Aberth sum (sweeps far from convergence)
    unsigned long long *maxcorr;   // bits of max |corr|/|z| of the sweep
    // Both O(n^2) kernels are cut into S segments (second grid dimension) so that a polynomial of a few 10^4
    // roots fills the chip: segment s of the Newton kernel evaluates the powers [s*L, (s+1)*L) of every estimate,
    // segment s of the sum kernel adds the terms of the estimates [s*J, (s+1)*J); the apply kernel joins them.
    int S;
    long long L;         // powers per Newton segment (a multiple of the chain count)
    long long J;         // estimates per segment of the Aberth sum
    cplx *pp, *pd;       // S*na: value and derivative of segment s's polynomial (in x, powers 0 .. L-1) at estimate idx[i]
    double *pe;          // S*na: its error scale sum |c_j| |x|^j
    cplx *ps;            // S*na: partial Aberth sums
    int *hit;            // n: estimate coincides with another one
    // Only the estimates that still move are worked on: an estimate whose Newton correction is zero (|p| at the
    // level of its evaluation error) stays where it is, so p need not be evaluated there again.  idx lists the
    // na estimates of this sweep (lane i of every kernel works on estimate idx[i]; the segment arrays above are
    // indexed [s*na + i]); the apply kernel appends those that moved to idx_out and counts them in *cnt.
    const int *idx;
    int *idx_out;
    int *cnt;
    long long na;
    // polish != 0 (after the iteration has converged): one plain Newton step per estimate -- no repulsion sum (ps is not
    // read), no freezing at the evaluation-noise bound, steps limited to 1e-3 (1 + |z|).  The bound of the sweeps is a
    // worst-case one; inside a cluster of roots it stops the estimates a factor ~10 further out than the polynomial's
    // conditioning allows, and Newton's method converges quadratically from there (the neighbours are 100x further away).
    int polish = 0;
};

// p(x) = sum_r x^r P_r(x^NCH): NCH independent Horner chains (plus their derivatives and the |coefficient| chains
// of the error scale) per lane, with the coefficients staged through LDS in tiles that the workgroup loads
// together.  asc(k) is the coefficient of x^k: x = z reads coef[n-k], x = 1/z (|z| > 1) reads coef[k].
// blockIdx.y = segment of the coefficient range (AberthParams).
template <int TILE> FA_DEV void body_aberth_newton(const AberthParams &P)
{
    constexpr int NCH = 8;
    FA_LDS_DECL
    cplx *tile = (cplx *)FA_LDS_PTR;
    double *tabs = (double *)(tile + 2 * TILE);   // |coefficient| of the same tile, both orientations
    const long long li = (long long)FA_BID * FA_BDIM + FA_TID;
    const long long n = P.n;
    const long long base = (long long)FA_BID_Y * P.L;   // lowest power of this segment
    const bool act = li < P.na;
    const cplx z = act ? P.z[P.idx[li]] : cmake(0.5, 0.0);
    const bool inside = cnorm2(z) <= 1.0;
    const cplx x = inside ? z : c_div(cmake(1.0, 0.0), z);
    cplx xp[NCH];               // x^0 .. x^(NCH-1)
    xp[0] = cmake(1.0, 0.0);
#pragma unroll
    for (int c = 1; c < NCH; c++) xp[c] = xp[c - 1] * x;
    const cplx w = xp[NCH - 1] * x;   // x^NCH
    const double ax = sqrt(cnorm2(x));
    double aw = 1.0;
#pragma unroll
    for (int c = 0; c < NCH; c++) aw *= ax;
    cplx pc[NCH], dc[NCH];
    double ec[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) { pc[c] = cmake(0.0, 0.0); dc[c] = cmake(0.0, 0.0); ec[c] = 0.0; }
    // local powers run from the top block (m = Mtop) down to m = 0; block m holds powers NCH*m .. NCH*m + NCH-1
    const long long Mtop = P.L / NCH - 1;
    for (long long mhi = Mtop; mhi >= 0; mhi -= TILE / NCH) {
        const long long mlo = (mhi - TILE / NCH + 1 > 0) ? mhi - TILE / NCH + 1 : 0;
        const long long kbase = base + NCH * mlo;     // lowest (global) power held by this tile
        if (kbase > n) continue;                      // the whole tile lies above the leading coefficient (uniform)
        FA_SYNC();
        for (int t = FA_TID; t < TILE; t += FA_BDIM) {
            const long long kk = kbase + t;
            // both orientations are needed inside one workgroup: interleave them
            const cplx ca = (kk <= n) ? P.coef[n - kk] : cmake(0.0, 0.0);
            const cplx cb = (kk <= n) ? P.coef[kk] : cmake(0.0, 0.0);
            tile[2 * t] = ca;
            tile[2 * t + 1] = cb;
            tabs[2 * t] = sqrt(cnorm2(ca));
            tabs[2 * t + 1] = sqrt(cnorm2(cb));
        }
        FA_SYNC();
        const int off = inside ? 0 : 1;
        for (long long m = mhi; m >= mlo; m--) {
            const int t = (int)(NCH * (m - mlo));
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                dc[c] = dc[c] * w + pc[c];
                pc[c] = pc[c] * w + tile[2 * (t + c) + off];
                ec[c] = fma(ec[c], aw, tabs[2 * (t + c) + off]);
            }
        }
    }
    if (!act) return;
    // p = sum_c x^c P_c;  p' = sum_c c x^(c-1) P_c + NCH x^(NCH-1) sum_c x^c P_c'
    cplx p = pc[0], dsum = dc[0], dp = cmake(0.0, 0.0);
    double escale = ec[0], axp = 1.0;
#pragma unroll
    for (int c = 1; c < NCH; c++) {
        p = p + xp[c] * pc[c];
        dsum = dsum + xp[c] * dc[c];
        dp = dp + (xp[c - 1] * pc[c]) * (double)c;
        axp *= ax;
        escale = fma(axp, ec[c], escale);
    }
    dp = dp + (xp[NCH - 1] * dsum) * (double)NCH;
    const long long o = (long long)FA_BID_Y * P.na + li;
    P.pp[o] = p;
    P.pd[o] = dp;
    P.pe[o] = escale;
}

// x^e, e >= 0, by repeated squaring (|x| <= 1: underflow to zero is the right answer)
FA_DEV cplx c_powi(cplx x, long long e)
{
    cplx r = cmake(1.0, 0.0);
    while (e > 0) {
        if (e & 1) r = r * x;
        x = x * x;
        e >>= 1;
    }
    return r;
}

// One term 1/(zk - zj) = conj(d)/|d|^2 of the Aberth sum, branch-free: the own term counts as 1/(1 + 0i) (taken
// out by the caller); an estimate that COINCIDES with another one contributes nothing here (instead of 1/0, after
// which both would freeze) and raises `hit`, and the caller then pushes this estimate a tiny, index-dependent step
// away so that the pair separates.
FA_DEV cplx aberth_term(cplx zk, cplx zj, bool own, bool &hit)
{
    cplx d = zk - zj;
    d = own ? cmake(1.0, 0.0) : d;
    double n2 = cnorm2(d);
    const bool zero = (n2 == 0.0);
    hit = hit || zero;
    n2 = zero ? 1.0 : n2;
    const double inv = aberth_rcp(n2);
    return cmake(d.x * inv, -d.y * inv);
}
// Partial Aberth sums: blockIdx.y = segment of the estimates z_j, lane = estimate k.
template <int TILE> FA_DEV void body_aberth_sum(const AberthParams &P)
{
    FA_LDS_DECL
    cplx *tile = (cplx *)FA_LDS_PTR;
    const long long li = (long long)FA_BID * FA_BDIM + FA_TID;
    const bool act = li < P.na;
    const long long k = act ? (long long)P.idx[li] : -1;
    const cplx zk = act ? P.z[k] : cmake(0.0, 0.0);
    const long long jlo = (long long)FA_BID_Y * P.J;
    const long long jhi = (jlo + P.J < P.n) ? jlo + P.J : P.n;
    cplx s = cmake(0.0, 0.0);
    bool hit = false;   // this estimate coincides with another one
    if (P.fast) {
        // far from convergence (corrections of 1e-3 and more) the repulsion sum only steers the estimates apart:
        // single precision is plenty, and it runs at several times the rate of the double-precision reciprocals
        float sx = 0.f, sy = 0.f, tx = 0.f, ty = 0.f;
        const float zx = (float)zk.x, zy = (float)zk.y;
        float *tf = (float *)tile;
        for (long long j0 = jlo; j0 < jhi; j0 += TILE) {
            FA_SYNC();
            if (j0 + FA_TID < jhi) {
                const cplx zj = P.z[j0 + FA_TID];
                tf[2 * FA_TID] = (float)zj.x;
                tf[2 * FA_TID + 1] = (float)zj.y;
            }
            FA_SYNC();
            const int lim = (int)((jhi - j0 < TILE) ? jhi - j0 : TILE);
            if (act) {
                int j = 0;
                for (; j + 1 < lim; j += 2) {
                    float ax = zx - tf[2 * j], ay = zy - tf[2 * j + 1];
                    float bx = zx - tf[2 * j + 2], by = zy - tf[2 * j + 3];
                    float na = ax * ax + ay * ay, nb = bx * bx + by * by;
                    // own term and exact coincidences contribute nothing (the double-precision sweeps separate them)
                    const float ia = (na > 0.f) ? fa_rcp_approx_f32(na) : 0.f, ib = (nb > 0.f) ? fa_rcp_approx_f32(nb) : 0.f;
                    sx += ax * ia; sy -= ay * ia;
                    tx += bx * ib; ty -= by * ib;
                }
                if (j < lim) {
                    const float ax = zx - tf[2 * j], ay = zy - tf[2 * j + 1];
                    const float na = ax * ax + ay * ay;
                    const float ia = (na > 0.f) ? fa_rcp_approx_f32(na) : 0.f;
                    sx += ax * ia; sy -= ay * ia;
                }
            }
        }
        s = cmake((double)sx + (double)tx, (double)sy + (double)ty);
    } else {
        for (long long j0 = jlo; j0 < jhi; j0 += TILE) {
            FA_SYNC();
            if (j0 + FA_TID < jhi) tile[FA_TID] = P.z[j0 + FA_TID];
            FA_SYNC();
            const int lim = (int)((jhi - j0 < TILE) ? jhi - j0 : TILE);
            if (act) {
                cplx s1 = cmake(0.0, 0.0);
                int j = 0;
                for (; j + 1 < lim; j += 2) {
                    s = s + aberth_term(zk, tile[j], j0 + j == k, hit);
                    s1 = s1 + aberth_term(zk, tile[j + 1], j0 + j + 1 == k, hit);
                }
                if (j < lim) s = s + aberth_term(zk, tile[j], j0 + j == k, hit);
                s = s + s1;
            }
        }
        if (k >= jlo && k < jhi) s = s - cmake(1.0, 0.0);   // the own term was counted as 1/(1 + 0i)
    }
    if (act) {
        P.ps[(long long)FA_BID_Y * P.na + li] = s;
        if (hit) P.hit[k] = 1;
    }
}

// Joins the segments: p(z), p'(z) and the error scale from the segment polynomials (Horner in x^L), the Newton
// correction w = p/p', the Aberth sum, and the new estimate z - w/(1 - w*sum).
FA_DEV void body_aberth_apply(const AberthParams &P)
{
    const long long li = (long long)FA_BID * FA_BDIM + FA_TID;
    const long long n = P.n;
    const bool act = li < P.na;
    double rel = 0.0;
    bool moved = false;
    long long k = 0;
    if (act) {
        k = P.idx[li];
        const cplx zk = P.z[k];
        const bool inside = cnorm2(zk) <= 1.0;
        const cplx x = inside ? zk : c_div(cmake(1.0, 0.0), zk);
        const cplx xL1 = c_powi(x, P.L - 1), xL = xL1 * x;
        const double axL = sqrt(cnorm2(xL));
        // Q(y) = sum_s p_s y^s at y = x^L:  p = Q,  p' = sum_s y^s p_s' + L x^(L-1) Q'(y)
        cplx q = cmake(0.0, 0.0), dq = cmake(0.0, 0.0), d = cmake(0.0, 0.0), s = cmake(0.0, 0.0);
        double escale = 0.0;
        for (int g = P.S - 1; g >= 0; g--) {
            const long long o = (long long)g * P.na + li;
            dq = dq * xL + q;
            q = q * xL + P.pp[o];
            d = d * xL + P.pd[o];
            escale = fma(escale, axL, P.pe[o]);
            if (!P.polish) s = s + P.ps[o];
        }
        const cplx p = q;
        const cplx dp = d + (xL1 * dq) * (double)P.L;
        // |p(x)| at the level of its own evaluation error (running error bound of Horner's scheme, statistical
        // sqrt(n) growth): the estimate is a root to working accuracy and is left alone -- without this, roots of
        // an ill-conditioned polynomial keep receiving corrections of the size of the noise and never "converge"
        const double noise = P.polish ? 0.0 : 2.220446049250313e-16 * (2.0 * sqrt((double)n) + 2.0) * escale;
        cplx w;
        if (cnorm2(p) <= noise * noise) {
            w = cmake(0.0, 0.0);
        } else if (inside) {
            w = c_div(p, dp);
        } else {
            // p(z) = z^n q(y), y = 1/z:  p'/p = n/z - y^2 q'(y)/q(y)
            const cplx t = x * (double)n - (x * x) * c_div(dp, p);
            w = c_div(cmake(1.0, 0.0), t);
        }
        if (!(w.x == w.x) || !(w.y == w.y) || fabs(w.x) > 1.0e300 || fabs(w.y) > 1.0e300) w = cmake(0.0, 0.0);
        const bool hit = !P.polish && P.hit[k] != 0;
        if (!P.polish) P.hit[k] = 0;
        cplx corr = c_div(w, cmake(1.0, 0.0) - w * s);
        const double az0 = sqrt(cnorm2(zk));
        if (P.polish && !(cnorm2(corr) <= 1.0e-6 * (1.0 + az0) * (1.0 + az0))) corr = cmake(0.0, 0.0);   // NaN or not small
        if (hit) {   // coincident estimates: separate them (different steps for different k)
            const double h = 1.0e-8 * (1.0 + az0);
            corr = corr + cmake(h * (1.0 + (double)(k & 7)), -0.5 * h * (1.0 + (double)((k >> 3) & 7)));
        }
        if (!(corr.x == corr.x) || !(corr.y == corr.y) || fabs(corr.x) > 1.0e300 || fabs(corr.y) > 1.0e300) {
            // degenerate step (0/0, overflow): move the estimate a little instead of freezing it; the sweep
            // then does not count as converged
            const double h = 1.0e-6 * (1.0 + az0);
            corr = cmake(h * (1.0 + (double)(k & 3)), -h * (1.0 + (double)((k >> 2) & 3)));
        }
        const cplx zn = zk - corr;
        P.z_out[k] = zn;   // an estimate that stays writes its value into the other buffer too: both agree from now on
        moved = (corr.x != 0.0 || corr.y != 0.0);
        // polish: only estimates that still moved by more than rounding take part in the second polish sweep
        if (P.polish) moved = cnorm2(corr) > 1.0e-26 * (1.0 + az0) * (1.0 + az0);
        const double az = sqrt(cnorm2(zn));
        rel = sqrt(cnorm2(corr)) / (az > 1.0e-300 ? az : 1.0e-300);
    }
    const int slot = fa_wave_append_slot(P.cnt, moved);
    if (moved) P.idx_out[slot] = (int)k;
    fa_wave_atomic_max_f64bits(P.maxcorr, rel);
}
