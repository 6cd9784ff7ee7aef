// nft_real.h -- product tree for transfer matrices with REAL polynomial coefficients (general 4-entry form).
//
// Where it applies: fnft_kdvv / fnft__kdv_fscatter (src/fnft_kdvv.c:126-209, src/private/fnft__kdv_fscatter.c:45-83:
// r = -1, and a real potential u) -- every step matrix of fnft__akns_fscatter.c:116-917 is then a real polynomial
// matrix, and so is every product of fnft__poly_fmult.c:381-546.  The complex path of nft_kernels.h transforms
// four complex polynomials of length N >= 2d+1 per matrix; here a real polynomial a[0..2M) is FOLDED into M complex
// numbers z[n] = a[n] + i a[n+M], twisted by zeta^n (zeta = exp(+2 pi i/(4M))) and transformed with length M:
//     X[k] = sum_n z[n] zeta^n exp(-2 pi i n k/M) = a(x_k),   x_k = zeta exp(-2 pi i k/M),   x_k^M = i,
// i.e. the polynomial is evaluated at the M roots of x^M = i.  x^(2M) + 1 = (x^M - i)(x^M + i), and a real polynomial's
// residue modulo the second factor is the conjugate of the first, so these M values carry the product modulo
// x^(2M) + 1 (a negacyclic product of length 2M): pointwise products of spectra, NO mirror bins, half the points and
// half the bytes of the complex form.  Inverse: z_c[n] = zeta^-n IDFT_M(X_c)[n] / M, c[n] = Re z_c[n],
// c[n+M] = Im z_c[n].  M = d when d is a power of two (the single wrapped coefficient, index 2d = 2M, comes back on
// index 0 with a minus sign and is the product of the constant terms: added back), otherwise the power of two above d.
//
// Layout: the body/tail arrays of nft_kernels.h reinterpreted as arrays of double: entry e of matrix j holds powers
// d..1 at rbody[e*plane + j*d + k] and its constant term at rtail[e*n + j]; scale / wexp / max2 as in the complex form.
#pragma once

FA_HD size_t nft_real_len(size_t d)
{
    size_t p = 1;
    while (p < d) p *= 2;
    return p;   // smallest power of two >= d: equal to d ("exact"), or above it ("loose": >= d + 1)
}

// constant term of entry e = 2*row + col of a product from the factors' constant terms (general form, real):
// C[row][col] = A[row][0] B[0][col] + A[row][1] B[1][col]; rt: the level's tails, mA / mB the factors, sA / sB their scales
FA_DEV double rtail_product_at(const double *rt, size_t n_in, size_t mA, size_t mB, int e, double sA, double sB)
{
    const size_t row = (size_t)(e >> 1), col = (size_t)(e & 1);
    const double a0 = rt[(2 * row) * n_in + mA] * sA, a1 = rt[(2 * row + 1) * n_in + mA] * sA;
    const double b0 = rt[col * n_in + mB] * sB, b1 = rt[(2 + col) * n_in + mB] * sB;
    return fma(a1, b1, a0 * b0);
}
FA_DEV double rsplit_tail_product(const TreeLevel &L, int P, int e, double sA, double sB)
{
    return rtail_product_at((const double *)L.tail_in, (size_t)L.n_in, (size_t)(2 * P), (size_t)(2 * P + 1), e, sA, sB);
}

// ---------------------------------------------------------------------------------------------
// direct pair product for tiny degrees, one lane per pair (fnft__poly_fmult.c:239-328)
// ---------------------------------------------------------------------------------------------
template <int DEG> FA_DEV void body_rpair_school(const TreeLevel &L)
{
    const long long P = (long long)FA_BID * FA_BDIM + FA_TID;
    const int n_out = L.n_in / 2;
    if (P >= n_out) return;
    constexpr int d = DEG;
    const double *rb = (const double *)L.body_in, *rt = (const double *)L.tail_in;
    double *ob = (double *)L.body_out, *ot = (double *)L.tail_out;
    double A[4][d + 1], Bm[4][d + 1];
    const double sA = L.scale_in[2 * P], sB = L.scale_in[2 * P + 1];
#pragma unroll
    for (int e = 0; e < 4; e++) {
#pragma unroll
        for (int k = 0; k < d; k++) {
            A[e][k] = rb[(size_t)e * L.plane + (size_t)(2 * P) * d + k] * sA;
            Bm[e][k] = rb[(size_t)e * L.plane + (size_t)(2 * P + 1) * d + k] * sB;
        }
        A[e][d] = rt[(size_t)e * L.n_in + 2 * P] * sA;
        Bm[e][d] = rt[(size_t)e * L.n_in + 2 * P + 1] * sB;
    }
    double C[4][2 * d + 1];
    double m2 = 0.0;
#pragma unroll
    for (int row = 0; row < 2; row++)
#pragma unroll
        for (int col = 0; col < 2; col++) {
            const int e = 2 * row + col;
#pragma unroll
            for (int k = 0; k <= 2 * d; k++) C[e][k] = 0.0;
#pragma unroll
            for (int i = 0; i <= d; i++)
#pragma unroll
                for (int j = 0; j <= d; j++) {
                    C[e][i + j] = fma(A[2 * row][i], Bm[col][j], C[e][i + j]);
                    C[e][i + j] = fma(A[2 * row + 1][i], Bm[2 + col][j], C[e][i + j]);
                }
#pragma unroll
            for (int k = 0; k <= 2 * d; k++) m2 = fmax(m2, C[e][k] * C[e][k]);
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
        for (int k = 0; k < 2 * d; k++) ob[(size_t)e * L.plane + (size_t)P * (2 * d) + k] = C[e][k] * sc;
        ot[(size_t)e * n_out + P] = C[e][2 * d] * sc;
    }
    L.scale_out[P] = 1.0;
    L.wexp_out[P] = L.wexp_in[2 * P] + L.wexp_in[2 * P + 1] + a;
}

// ---------------------------------------------------------------------------------------------
// pair product in one workgroup: folded transforms of length M, M/R lanes per pair, B pairs per workgroup
// (the IO object of pair_product_core, nft_kernels.h)
// ---------------------------------------------------------------------------------------------
template <int M, int R, int B> struct RTreeIO {
    const TreeLevel &L;
    long long P;
    bool active, exact;
    double sc[2];
    double m2;
    cplx tws[R];   // w_{4M}^{idx} of this lane's elements: conjugated on load (twist), plain on store (untwist)

    FA_DEV RTreeIO(const TreeLevel &L_, int c, int v) : L(L_)
    {
        P = (long long)FA_BID * B + c;
        active = P < L.n_in / 2;
        exact = (L.d == M);
        m2 = 0.0;
        sc[0] = active ? L.scale_in[2 * P] : 0.0;
        sc[1] = active ? L.scale_in[2 * P + 1] : 0.0;
#pragma unroll
        for (int i = 0; i < R; i++) tws[i] = L.twx[v + (M / R) * i];
    }
    FA_DEV double tail(int which, int e) const
    {
        return ((const double *)L.tail_in)[(size_t)e * L.n_in + 2 * P + which] * sc[which];
    }
    FA_DEV void load(int which, int e, cplx (&x)[R], int v, int)
    {
        const int d = L.d;
        const double *src = (const double *)L.body_in + (size_t)e * L.plane + (size_t)(2 * P + which) * d;
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int idx = v + (M / R) * i;
            double re = 0.0, im = 0.0;
            if (active) {
                if (idx < d) re = src[idx] * sc[which];
                else if (idx == d) re = tail(which, e);
                if (exact && idx == 0) im = tail(which, e);   // index d = M folds onto 0
            }
            // (re + i im) * conj(w)
            x[i] = cmake(fma(re, tws[i].x, im * tws[i].y), fma(im, tws[i].x, -(re * tws[i].y)));
        }
    }
    FA_DEV double tail_product(int e) const
    {
        return rtail_product_at((const double *)L.tail_in, (size_t)L.n_in, (size_t)(2 * P), (size_t)(2 * P + 1), e, sc[0], sc[1]);
    }
    // results leave through the idle transform buffer (as doubles) when a workgroup holds several pairs, so that
    // lanes write consecutive elements (see TreeIO::store)
    FA_DEV void store(int e, cplx (&x)[R], int v, int c, cplx *lds, int &parity)
    {
        const int d2 = 2 * L.d;
        const int n_out = L.n_in / 2;
        const double inv = 1.0 / (double)M;
        double *stage = (double *)(lds + ((M > R) ? (size_t)parity * (size_t)(M * B) : 0));
        double *dst = (double *)L.body_out + (size_t)e * L.plane + (size_t)P * d2;
        if (B > 1 && M == R) FA_SYNC_LDS();   // no transform barrier separates consecutive stores
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int idx = v + (M / R) * i;
            const cplx val = (x[i] * inv) * tws[i];
            double re = val.x;
            const double im = val.y;
            if (active && idx == 0) {
                const double tp = tail_product(e);
                if (exact) re += tp;   // coefficient 2d = 2M came back on 0 with a minus sign
                ((double *)L.tail_out)[(size_t)e * n_out + P] = tp;
                m2 = fmax(m2, tp * tp);
            }
            if (!active) continue;
            const int hi = idx + M;
            if (idx < d2) m2 = fmax(m2, re * re);
            if (hi < d2) m2 = fmax(m2, im * im);
            if (B > 1) {
                if (idx < d2) {
                    int rot = idx + c;
                    rot = rot >= d2 ? rot - d2 * (rot / d2) : rot;
                    stage[(size_t)c * d2 + rot] = re;
                }
                if (hi < d2) {
                    int rot = hi + c;
                    rot = rot >= d2 ? rot - d2 * (rot / d2) : rot;
                    stage[(size_t)c * d2 + rot] = im;
                }
            } else {
                if (idx < d2) dst[idx] = re;
                if (hi < d2) dst[hi] = im;
            }
        }
        if (B > 1) {
            FA_SYNC_LDS();
            const long long P0 = (long long)FA_BID * B;
            const long long nvalid = (n_out - P0 < B) ? n_out - P0 : B;
            const int total = (int)nvalid * d2;
            double *out0 = (double *)L.body_out + (size_t)e * L.plane + (size_t)P0 * d2;
            for (int m = FA_TID; m < total; m += B * (M / R)) {
                const int c2 = m / d2, i2 = m - c2 * d2;
                int rot = i2 + c2;
                rot = rot >= d2 ? rot - d2 * (rot / d2) : rot;
                out0[m] = stage[(size_t)c2 * d2 + rot];
            }
            if (M > R) parity ^= 1;
        }
    }
};

// LDS: as body_pair_fft (transform buffers, twiddle table when small, B maxima)
template <int M, int R, int B, bool DB> FA_DEV void body_rpair(const TreeLevel &L)
{
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    constexpr size_t kBuf = (M > R) ? (size_t)(DB ? 2 : 1) * M * B : (size_t)M * B;
    constexpr bool kTwLds = (M > R) && (M <= 512);
    cplx *twl = lds + kBuf;
    unsigned long long *mx = (unsigned long long *)(twl + (kTwLds ? M : 0));
    const int tid = FA_TID;
    const int c = tid % B, v = tid / B;
    if (v == 0) mx[c] = 0ull;
    const cplx *tw = L.tw;
    if (kTwLds) tw = stage_twiddles<M, B *(M / R)>(twl, L.tw);
    RTreeIO<M, R, B> io(L, c, v);
    pair_product_core<M, R, B, DB, (M > R) && !kTwLds>(io, lds, tw);
    if (M > R) {
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
// split transforms, M = N1*N2 (element n = n1*N2 + n2, bin k = k1 + N1*k2).  The twist factors as
// zeta^n = exp(2 pi i n1/(4 N1)) * exp(2 pi i n2/(4M)): the column kernels apply the first factor, the row kernel
// (body_mid with BigLevel::rtwist) the second together with its own twiddle.  Y / Z scratch as in nft_kernels.h.
// ---------------------------------------------------------------------------------------------
// first split level: real coefficients -> column transform of the folded, twisted sequence
//   grid.x = N2/BC tiles, grid.y = 4*n_in polynomials
template <int N1, int R, int BC, bool DB> FA_DEV void body_rcol_fwd(const BigLevel &G)
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
    const bool exact = ((long long)N1 * N2 == (long long)d);
    const double sc = level_in_scale(L, mat);
    const double *src = (const double *)L.body_in + (size_t)e * L.plane + (size_t)mat * d;
    const double tl = ((const double *)L.tail_in)[(size_t)e * L.n_in + mat] * sc;
    cplx x[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int n1 = v + (N1 / R) * i;
        const long long idx = (long long)n1 * N2 + n2;
        double re = 0.0, im = 0.0;
        if (idx < d) re = src[idx] * sc;
        else if (idx == d) re = tl;
        if (exact && idx == 0) im = tl;
        const cplx w = G.twq[n1];   // exp(-2 pi i n1/(4 N1)); the twist is its conjugate
        x[i] = cmake(fma(re, w.x, im * w.y), fma(im, w.x, -(re * w.y)));
    }
    int parity = 0;
    fft_wg<N1, R, BC, -1, DB, true>(x, lds, v, c, G.tw1, parity);
    cplx *dst = G.Y + (size_t)poly * N1 * N2;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int k1 = v + (N1 / R) * i;
        dst[yz_index(N1, N2, k1, n2)] = x[i];
    }
}

// last split level: inverse column transform, untwist, unfold -> real coefficients
//   grid.x = N2/BC tiles, grid.y = 4*n_out polynomials
template <int N1, int R, int BC, bool DB> FA_DEV void body_rcol_inv(const BigLevel &G)
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
    const long long M = (long long)N1 * N2;
    const long long d2 = 2 * (long long)L.d;
    const bool exact = (M == (long long)L.d);
    const cplx *src = G.Z + (size_t)poly * N1 * N2;
    const double sA = level_in_scale(L, 2 * P), sB = level_in_scale(L, 2 * P + 1);   // wave-uniform
    cplx x[R];
#pragma unroll
    for (int i = 0; i < R; i++) x[i] = src[yz_index(N1, N2, v + (N1 / R) * i, n2)];
    int parity = 0;
    fft_wg<N1, R, BC, +1, DB, true>(x, lds, v, c, G.tw1, parity);
    double *dst = (double *)L.body_out + (size_t)e * L.plane + (size_t)P * d2;
    const double inv = 1.0 / (double)N1;
    double m2 = 0.0;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int n1 = v + (N1 / R) * i;
        const long long idx = (long long)n1 * N2 + n2;
        const cplx val = (x[i] * inv) * G.twq[n1];
        double re = val.x;
        const double im = val.y;
        if (idx == 0) {
            const double tp = rsplit_tail_product(L, P, e, sA, sB);
            if (exact) re += tp;
            ((double *)L.tail_out)[(size_t)e * n_out + P] = tp;
            m2 = fmax(m2, tp * tp);
        }
        if (idx < d2) {
            dst[idx] = re;
            m2 = fmax(m2, re * re);
        }
        if (idx + M < d2) {
            dst[idx + M] = im;
            m2 = fmax(m2, im * im);
        }
    }
    fa_wave_atomic_max_hi32(&L.max2_out[(size_t)P * kMax2Slots + max2_slot()], m2);   // P is uniform in the workgroup
}

// between two split levels: inverse column transform of level l (length N1), untwist, unfold -- the 2*N1 real
// coefficients of column n2 of the product, in registers only (maximum, constant term) -- then fold / twist / forward
// column transform (length 2*N1) of level l+1.  Y' is written unscaled (the pending scale is not known yet).
//   grid.x = N2/BC tiles, grid.y = 4*n_out polynomials
template <int N1, int R, int BC, bool DB> FA_DEV void body_rbridge(const BigLevel &G)
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
    const long long M = (long long)N1 * N2;
    const long long d2 = 2 * (long long)L.d;
    const bool exact = (M == (long long)L.d);
    const cplx *src = G.Z + (size_t)poly * N1 * N2;
    const double sA = level_in_scale(L, 2 * P), sB = level_in_scale(L, 2 * P + 1);   // wave-uniform
    cplx x[2 * R];
    int parity = 0;
    {
        cplx lo[R];
#pragma unroll
        for (int i = 0; i < R; i++) lo[i] = src[yz_index(N1, N2, v + (N1 / R) * i, n2)];
        fft_wg<N1, R, BC, +1, DB, true>(lo, lds, v, c, G.tw1, parity);
#pragma unroll
        for (int i = 0; i < R; i++) x[i] = lo[i];
    }
    const double inv = 1.0 / (double)N1;
    double m2 = 0.0;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int n1 = v + (N1 / R) * i;
        const long long idx = (long long)n1 * N2 + n2;
        const cplx val = (x[i] * inv) * G.twq[n1];
        double re = val.x, im = val.y;   // coefficients idx and idx + M of the product
        double im0 = 0.0;                // imaginary part of the NEXT level's folded element idx
        if (idx == 0) {
            const double tp = rsplit_tail_product(L, P, e, sA, sB);
            if (exact) {
                re += tp;     // coefficient 2d = 2M came back on 0 with a minus sign
                im0 = tp;     // next level (M' = 2M = d'): its constant term, index d' = M', folds onto 0
            }
            ((double *)L.tail_out)[(size_t)e * n_out + P] = tp;
            m2 = fmax(m2, tp * tp);
        }
        m2 = fmax(m2, re * re);   // idx < M <= 2d always
        if (idx + M < d2) m2 = fmax(m2, im * im);
        else if (idx + M == d2) im = rsplit_tail_product(L, P, e, sA, sB);   // loose: the constant term, index 2d < 2M
        else im = 0.0;                                                       // above the degree: rounding noise
        const cplx w0 = G.twq2[n1], w1 = G.twq2[n1 + N1];   // exp(-2 pi i n1'/(8 N1)); the twist is the conjugate
        x[i] = cmake(fma(re, w0.x, im0 * w0.y), fma(im0, w0.x, -(re * w0.y)));
        x[R + i] = cmake(im * w1.x, -(im * w1.y));
    }
    fa_wave_atomic_max_hi32(&L.max2_out[(size_t)P * kMax2Slots + max2_slot()], m2);   // P is uniform in the workgroup
    fft_wg<2 * N1, 2 * R, BC, -1, DB>(x, lds, v, c, G.tw1x2, parity);
    cplx *dst = G.Y + (size_t)poly * (2 * N1) * N2;
#pragma unroll
    for (int i = 0; i < 2 * R; i++) {
        const int k1 = v + (N1 / R) * i;   // (2 N1)/(2 R) = N1/R
        dst[yz_index(2 * N1, N2, k1, n2)] = x[i];
    }
}

// is the potential real?  (fnft_amd_kdvv_contspec_device in its default mode asks before it chooses the path)
struct RealCheckParams {
    const cplx *q;
    long long n;
    int *flag;   // set to 1 when a sample has a non-zero imaginary part
};
FA_DEV void body_real_check(const RealCheckParams &P)
{
    bool bad = false;
    for (long long i = (long long)FA_BID * FA_BDIM + FA_TID; i < P.n; i += (long long)FA_GDIM * FA_BDIM)
        bad = bad || (P.q[i].y != 0.0);
    if (bad) fa_atomic_or_i32(P.flag, 1);
}
