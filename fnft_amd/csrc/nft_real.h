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
// The same pair product with the FOUR entries of a factor transformed at once: a workgroup holds BP pairs, each by
// four groups of M/R lanes (entry g = 2*row + col); batch index of the cooperative transform = 4*pair + entry.  Three
// rounds of transforms (A, B, the product) instead of twelve -- a quarter of the barrier-separated exchange passes
// of body_rpair -- and a third of the registers per lane (one entry of each factor instead of all of A and a column of
// B).  Between the forward and the inverse round the spectra cross through LDS: every group needs row `row` of A and
// column `col` of B at its own bins,  C[row][col] = A[row][0] B[0][col] + A[row][1] B[1][col].
// LDS: two transform buffers of 4*BP*M elements (spectra exchange and the staged stores reuse them), the twiddle table
// when small, BP maxima.
// ---------------------------------------------------------------------------------------------
// Measured (cfg 5, M = 128 / 256 / 512): 109 / 123 / 141 us against 138 / 132 / 144 of body_rpair; with the global loads
// replaced by constants 83 / 101 / 119, without the stores 102 / 129 / 142, with the factors fetched by whole-workgroup
// coalesced loads through LDS no change -- the kernel is bound by the instruction issue of its transform passes
// (4 points per lane: ~1300 instructions per lane for 12 points), not by its memory accesses.
template <int M, int R, int BP, bool TWLDS> FA_DEV void body_rpair4(const TreeLevel &L)
{
    constexpr int B4 = 4 * BP;                     // interleaved transforms
    constexpr int T = B4 * (M / R);
    constexpr size_t kBuf = (size_t)M * B4;        // one transform buffer
    static_assert(M > R, "body_rpair4: at least one exchange pass");
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    cplx *twl = lds + 2 * kBuf;
    unsigned long long *mx = (unsigned long long *)(twl + (TWLDS ? M : 0));
    const int tid = FA_TID;
    const int c = tid % B4, v = tid / B4;
    const int g = c % 4, pl = c / 4;
    const int row = g >> 1, col = g & 1;
    if (tid < BP) mx[tid] = 0ull;
    const cplx *tw = L.tw;
    if (TWLDS) tw = stage_twiddles<M, T>(twl, L.tw);
    RTreeIO<M, R, BP> io(L, pl, v);
    cplx a[R], b[R];
    io.load(0, g, a, v, pl);
    io.load(1, g, b, v, pl);
    int parity = 0;
    fft_wg<M, R, B4, -1, true, !TWLDS>(a, lds, v, c, tw, parity);
    fft_wg<M, R, B4, -1, true, !TWLDS>(b, lds, v, c, tw, parity);
    // spectra of A into the idle buffer, of B into the other one (still being read by lanes that have not left the
    // last exchange of B's transform: barrier first); element [pair][bin][entry]
    cplx *SA = lds + (size_t)parity * kBuf, *SB = lds + (size_t)(parity ^ 1) * kBuf;
#pragma unroll
    for (int i = 0; i < R; i++) SA[((size_t)pl * M + (size_t)(v + (M / R) * i)) * 4 + g] = a[i];
    FA_SYNC_LDS();
#pragma unroll
    for (int i = 0; i < R; i++) SB[((size_t)pl * M + (size_t)(v + (M / R) * i)) * 4 + g] = b[i];
    FA_SYNC_LDS();
#pragma unroll
    for (int i = 0; i < R; i++) {
        const size_t o = ((size_t)pl * M + (size_t)(v + (M / R) * i)) * 4;
        const cplx p = SA[o + 2 * row], q = SA[o + 2 * row + 1];
        const cplx b1 = SB[o + col], b2 = SB[o + 2 + col];
        a[i] = cfma(q, b2, p * b1);
    }
    FA_SYNC_LDS();   // the inverse transform writes into SA
    fft_wg<M, R, B4, +1, true, !TWLDS>(a, lds, v, c, tw, parity);
    // untwist, unfold (real part: coefficient idx, imaginary part: coefficient idx + M), alias fix, maximum; staged as
    // doubles in the idle buffer, region [pair][entry] of 2d elements rotated by the batch index against bank conflicts
    const int d2 = 2 * L.d;
    const int n_out = L.n_in / 2;
    const double inv = 1.0 / (double)M;
    double *stage = (double *)(lds + (size_t)parity * kBuf);
    double m2 = 0.0;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int idx = v + (M / R) * i;
        const cplx val = (a[i] * inv) * io.tws[i];
        double re = val.x;
        const double im = val.y;
        if (io.active && idx == 0) {
            const double tp = io.tail_product(g);
            if (io.exact) re += tp;   // coefficient 2d = 2M came back on 0 with a minus sign
            ((double *)L.tail_out)[(size_t)g * n_out + io.P] = tp;
            m2 = fmax(m2, tp * tp);
        }
        const int hi = idx + M;
        if (idx < d2) {
            if (io.active) m2 = fmax(m2, re * re);
            int rot = idx + c;
            rot = rot >= d2 ? rot - d2 : rot;
            stage[(size_t)c * d2 + rot] = re;
        }
        if (hi < d2) {
            if (io.active) m2 = fmax(m2, im * im);
            int rot = hi + c;
            rot = rot >= d2 ? rot - d2 : rot;
            stage[(size_t)c * d2 + rot] = im;
        }
    }
    if (io.active) fa_atomic_max_u64(&mx[pl], dbits(m2));
    FA_SYNC_LDS();
    const long long P0 = (long long)FA_BID * BP;
    const int total = B4 * d2;
    for (int m = tid; m < total; m += T) {
        const int c2 = m / d2, i2 = m - c2 * d2;
        const long long Pq = P0 + c2 / 4;
        if (Pq >= n_out) continue;
        int rot = i2 + c2;
        rot = rot >= d2 ? rot - d2 : rot;
        ((double *)L.body_out)[(size_t)(c2 % 4) * L.plane + (size_t)Pq * d2 + i2] = stage[(size_t)c2 * d2 + rot];
    }
    if (v == 0 && g == 0 && io.active) {
        const double mm = bitsd(mx[pl]);
        int e2 = 0;
        if (mm > 0.0 && mm < 1.0e300) e2 = half_exponent(mm);
        L.scale_out[io.P] = pow2i(-e2);
        L.wexp_out[io.P] = L.wexp_in[2 * io.P] + L.wexp_in[2 * io.P + 1] + e2;
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

// ---------------------------------------------------------------------------------------------
// Per-sample coefficients of the even-order splitting schemes 2SPLIT6A/6B/8A/8B for real q, r
// (fnft__akns_fscatter.c:567-641, 795-912), composed from their elementary factors instead of the reference's
// expanded coefficient formulas (or the monomial program of body_coeffs_prog): the scheme is
//     sum_{n=1..T} w_n Psi_n,   T = ORDER/2,   w_n = n^(2(T-1)) / prod_{m != n} (n^2 - m^2),
//     Psi_n = X(1/2n) [Y(1/n) X(1/n)]^(n-1) Y(1/n) X(1/2n)     (n Strang steps),
// X = B, Y = A for the "B" schemes and the other way round for the "A" schemes, with A(a) = diag(1, z^(a*deg)) and
// B(b) = expm([[0,q],[r,0]] b eps_t) = [[c, q s],[r s, c]] (:46-59).  Psi_n is a polynomial in u = z^(deg/n) (B first) or
// z^(deg/2n) (A first); a factor is one shift of the second column or one 2x2 column mix.  One lane per sample, 64-lane
// workgroups, coefficients staged through LDS so that lanes write consecutive doubles.
// ---------------------------------------------------------------------------------------------
struct RStep { double c, qs, rs; };
// expm([[0,q],[r,0]] h) for real q, r (zero_freq_step restricted to the real axis: -q r >= 0 gives cos / sinc,
// -q r < 0 the hyperbolic pair; fnft__misc.c:306-314 for the sinc)
FA_DEV RStep rzero_freq_step(double h, double q, double r)
{
    const double s2 = -(q * r);
    const double y = h * sqrt(fabs(s2));
    double c, snc;
    if (s2 >= 0.0) {
        double sn;
        fa_sincos(y, &sn, &c);
        if (y >= 1.0e-8) snc = sn / y;
        else fa_sincos(y * 0.57735026918962576451, &sn, &snc);
    } else {
        const double ep = exp(y), em = 1.0 / ep;
        c = 0.5 * (ep + em);
        if (y >= 1.0e-8) snc = 0.5 * (ep - em) / y;
        else { const double e2 = exp(y * 0.57735026918962576451); snc = 0.5 * (e2 + 1.0 / e2); }
    }
    RStep e;
    e.c = c;
    e.qs = q * (h * snc);
    e.rs = r * (h * snc);
    return e;
}
template <int NU> struct RPolyMat {
    double m[2][2][NU + 1];   // ascending powers of u
    FA_DEV void set(const RStep &b)
    {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int k = 0; k <= NU; k++) m[i][j][k] = 0.0;
        m[0][0][0] = b.c; m[0][1][0] = b.qs; m[1][0][0] = b.rs; m[1][1][0] = b.c;
    }
    FA_DEV void identity()
    {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int k = 0; k <= NU; k++) m[i][j][k] = 0.0;
        m[0][0][0] = 1.0; m[1][1][0] = 1.0;
    }
    // * diag(1, u^K): the second column moves up K powers
    template <int K> FA_DEV void shift()
    {
#pragma unroll
        for (int i = 0; i < 2; i++) {
#pragma unroll
            for (int k = NU; k >= K; k--) m[i][1][k] = m[i][1][k - K];
#pragma unroll
            for (int k = 0; k < K; k++) m[i][1][k] = 0.0;
        }
    }
    // * [[c, qs],[rs, c]]
    FA_DEV void mix(const RStep &b)
    {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int k = 0; k <= NU; k++) {
                const double x0 = m[i][0][k], x1 = m[i][1][k];
                m[i][0][k] = fma(x1, b.rs, x0 * b.c);
                m[i][1][k] = fma(x0, b.qs, x1 * b.c);
            }
    }
};
// Psi_n accumulated into P[e][k] (k: highest power of z first, as stored), weight w
template <int DEG, int NS, bool BFIRST> FA_DEV void rstrang_term(double eps_t, double q, double r, double w, double (&P)[4][DEG + 1])
{
    constexpr int NU = BFIRST ? NS : 2 * NS;
    constexpr int G = DEG / NU;   // powers of z per power of u
    const RStep full = rzero_freq_step(eps_t / (double)NS, q, r);
    RPolyMat<NU> M;
    if constexpr (BFIRST) {
        const RStep half = rzero_freq_step(eps_t / (double)(2 * NS), q, r);
        M.set(half);
#pragma unroll
        for (int i = 1; i <= NS; i++) {
            M.template shift<1>();
            M.mix(i < NS ? full : half);
        }
    } else {
        M.identity();
        M.template shift<1>();
#pragma unroll
        for (int i = 1; i <= NS; i++) {
            M.mix(full);
            if (i < NS) M.template shift<2>();
            else M.template shift<1>();
        }
    }
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
        for (int k = 0; k <= NU; k++) P[e][DEG - k * G] = fma(w, M.m[e >> 1][e & 1][k], P[e][DEG - k * G]);
}
template <int ORDER, bool BFIRST> struct RStrangCfg {
    static constexpr int T = ORDER / 2;
    static constexpr int DEG = (T == 3 ? 6 : 12) * (BFIRST ? 1 : 2);
};
// coefficient matrix of slot j of signal b (sample D-1-j, or the identity pad z^deg * I of fnft__poly_fmult.c:422-438
// beyond the signal; all zero for lanes outside the grid), highest power first
template <int ORDER, bool BFIRST>
FA_DEV void rstrang_sample(const CoeffParams &P, bool act, int b, long long j, double (&C)[4][RStrangCfg<ORDER, BFIRST>::DEG + 1])
{
    constexpr int T = RStrangCfg<ORDER, BFIRST>::T, DEG = RStrangCfg<ORDER, BFIRST>::DEG;
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
        for (int k = 0; k <= DEG; k++) C[e][k] = 0.0;
    if (act && j < P.D) {
        const size_t src = (size_t)b * P.D + (size_t)(P.D - 1 - j);
        const cplx qc = P.q[src];
        const cplx rc = P.r ? P.r[src] : cmake(-qc.x * (double)P.kappa, 0.0);
        if (qc.y != 0.0 || rc.y != 0.0) fa_atomic_or_i32(P.status, 8);
        const double q = qc.x, r = rc.x;
        // w_n = n^(2(T-1)) / prod_{m != n} (n^2 - m^2)
        if constexpr (T == 3) {
            rstrang_term<DEG, 1, BFIRST>(P.eps_t, q, r, 1.0 / 24.0, C);
            rstrang_term<DEG, 2, BFIRST>(P.eps_t, q, r, -16.0 / 15.0, C);
            rstrang_term<DEG, 3, BFIRST>(P.eps_t, q, r, 81.0 / 40.0, C);
        } else {
            rstrang_term<DEG, 1, BFIRST>(P.eps_t, q, r, -1.0 / 360.0, C);
            rstrang_term<DEG, 2, BFIRST>(P.eps_t, q, r, 16.0 / 45.0, C);
            rstrang_term<DEG, 3, BFIRST>(P.eps_t, q, r, -729.0 / 280.0, C);
            rstrang_term<DEG, 4, BFIRST>(P.eps_t, q, r, 1024.0 / 315.0, C);
        }
    } else if (act) {
        C[0][0] = 1.0;
        C[3][0] = 1.0;
    }
}

template <int ORDER, bool BFIRST> FA_DEV void body_rcoeffs_strang(const CoeffParams &P)
{
    constexpr int T = RStrangCfg<ORDER, BFIRST>::T, DEG = RStrangCfg<ORDER, BFIRST>::DEG;
    FA_LDS_DECL
    double *stage = (double *)FA_LDS_PTR;   // 64 x DEG
    const int lane = FA_TID;
    const long long gid0 = (long long)FA_BID * FA_BDIM, gid = gid0 + lane;
    const long long n = (long long)P.batch * P.Dpad;
    const bool act = gid < n;
    const int b = act ? (int)(gid / P.Dpad) : 0, j = act ? (int)(gid % P.Dpad) : 0;
    double C[4][DEG + 1];
    rstrang_sample<ORDER, BFIRST>(P, act, b, j, C);
    const long long nact = (n - gid0 < 64) ? n - gid0 : 64;
    double *rb = (double *)P.body, *rt = (double *)P.tail;
#pragma unroll
    for (int e = 0; e < 4; e++) {
#pragma unroll
        for (int k = 0; k < DEG; k++) stage[lane * DEG + ((k + lane) % DEG)] = C[e][k];
        if (act) rt[(size_t)e * n + gid] = C[e][DEG];
        FA_SYNC();
        double *out0 = rb + (size_t)e * P.plane + (size_t)gid0 * DEG;
        for (int m = lane; m < (int)nact * DEG; m += 64) {
            const int t2 = m / DEG, k2 = m - t2 * DEG;
            out0[m] = stage[t2 * DEG + ((k2 + t2) % DEG)];
        }
        FA_SYNC();
    }
    if (act) {
        P.scale[gid] = 1.0;
        P.wexp[gid] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Row step of a split level in the general 4-entry form, written like body_mid_sym (nft_kernels.h) for memory-level
// parallelism: the four rows of the left factor are requested back to back, the rows of the right factor's two columns
// from inside the left factor's transforms (fft_wg2 hooks), the transforms run in pairs
//     (A00, A01) (A10, A11) (B00, B10) -> C00 = A00 B00 + A01 B10, C10 = A10 B00 + A11 B10 -> inverse pair, store,
//     (B01, B11) -> C01, C11 -> inverse pair, store,
// and every barrier is an LDS-only one.  Serves the complex general form (Y rows; with spectral doubling the even rows
// come from the previous level's Z) and the real-coefficient path (BigLevel::rtwist).  8 + 4 transforms as
// fnft__poly_fmult.c:239-328.
// ---------------------------------------------------------------------------------------------
template <int N2, int R> struct MidColLoader {
    const cplx *p0, *p1;
    int rows, row, v;
    cplx (&x0)[R];
    cplx (&x1)[R];
    FA_DEV void operator()(int k) const
    {
#pragma unroll
        for (int i = 0; i < R; i++)
            if (i / 2 == k) {
                const size_t o = yz_index(rows, N2, row, v + (N2 / R) * i);
                x0[i] = p0[o];
                x1[i] = p1[o];
            }
    }
};
template <int N2, int R> FA_DEV void body_mid_gen(const BigLevel &G)
{
    FA_LDS_DECL
    cplx *lds = (cplx *)FA_LDS_PTR;
    const TreeLevel &L = G.L;
    const int v = FA_TID;
    const int n_in = L.n_in, n_out = n_in / 2;
    const long long P = FA_BID / G.N1;
    const int k1 = FA_BID % G.N1;
    const long long mA = 2 * P, mB = 2 * P + 1;
    const bool split = G.y_split != 0;
    const int rows = split ? G.N1 / 2 : G.N1;
    const int row = split ? (k1 >> 1) : k1;
    const cplx *base = split ? ((k1 & 1) ? G.Y : G.Zprev) : G.Y;
    const size_t blk = (size_t)rows * N2;
    // entry e of matrix m: polynomial e*n_in + m
    const cplx *pA = base + (size_t)mA * blk, *pB = base + (size_t)mB * blk;
    const size_t estr = (size_t)n_in * blk;
    cplx a00[R], a01[R], a10[R], a11[R], b0[R], b1[R], c0[R], c1[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
        const size_t o = yz_index(rows, N2, row, v + (N2 / R) * i);
        a00[i] = pA[o]; a01[i] = pA[estr + o]; a10[i] = pA[2 * estr + o]; a11[i] = pA[3 * estr + o];
    }
    MidColLoader<N2, R> ld0{pB, pB + 2 * estr, rows, row, v, b0, b1};              // column 0 of B: B00, B10
    MidColLoader<N2, R> ld1{pB + estr, pB + 3 * estr, rows, row, v, c0, c1};      // column 1 of B: B01, B11
    // ---- bookkeeping of the level (once per pair) and the factors every polynomial shares ----------------
    const int pendA = level_in_pending_exp(L, mA), pendB = level_in_pending_exp(L, mB);   // workgroup-uniform
    if (k1 == 0 && v == 0) L.wexp_out[P] = L.wexp_in[mA] + pendA + L.wexp_in[mB] + pendB;
    if (k1 == 0 && v < kMax2Slots) L.max2_out[(size_t)P * kMax2Slots + v] = 0u;
    const bool rescale = G.y_unscaled != 0;
    const double scA = !rescale ? 1.0 : (L.in_pending ? pow2i(-pendA) : L.scale_in[mA]);
    const double scB = !rescale ? 1.0 : (L.in_pending ? pow2i(-pendB) : L.scale_in[mB]);
    const cplx wbase = big_twiddle(G.btw, row_tw_index(G, k1, v));
    cplx wu[R];
    wu[0] = cmake(1.0, 0.0);
#pragma unroll
    for (int i = 1; i < R; i++) {
        const cplx w = big_twiddle(G.btw, row_tw_index(G, k1, (N2 / R) * i));
        wu[i] = cmake(fa_uniform(w.x), fa_uniform(w.y));
    }
    const cplx *tw = G.tw2;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const cplx wA = ((i == 0) ? wbase : wbase * wu[i]) * scA;
        a00[i] = a00[i] * wA; a01[i] = a01[i] * wA;
    }
    fft_wg2<N2, R, 1, -1, true>(a00, a01, lds, v, 0, tw, ld0);
#pragma unroll
    for (int i = 0; i < R; i++) {
        const cplx wA = ((i == 0) ? wbase : wbase * wu[i]) * scA;
        a10[i] = a10[i] * wA; a11[i] = a11[i] * wA;
    }
    fft_wg2<N2, R, 1, -1, true>(a10, a11, lds, v, 0, tw, ld1);
    cplx *dZ = G.Z + (size_t)P * G.N1 * N2;
    const size_t zstr = (size_t)n_out * G.N1 * N2;
    const cplx wb = wbase * (1.0 / (double)N2);
    // ---- column 0 ------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < R; i++) {
        const cplx wB = ((i == 0) ? wbase : wbase * wu[i]) * scB;
        b0[i] = b0[i] * wB; b1[i] = b1[i] * wB;
    }
    fft_wg2<N2, R, 1, -1, true>(b0, b1, lds, v, 0, tw);
#pragma unroll
    for (int i = 0; i < R; i++) {
        const cplx x0 = b0[i], x1 = b1[i];
        b0[i] = cfma(a01[i], x1, a00[i] * x0);
        b1[i] = cfma(a11[i], x1, a10[i] * x0);
    }
    fft_wg2<N2, R, 1, +1, true>(b0, b1, lds, v, 0, tw);
#pragma unroll
    for (int i = 0; i < R; i++) {
        const cplx w = cconj((i == 0) ? wb : wb * wu[i]);
        const size_t o = yz_index(G.N1, N2, k1, v + (N2 / R) * i);
        dZ[o] = b0[i] * w;                // C00
        dZ[2 * zstr + o] = b1[i] * w;     // C10
    }
    // ---- column 1 ------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < R; i++) {
        const cplx wB = ((i == 0) ? wbase : wbase * wu[i]) * scB;
        c0[i] = c0[i] * wB; c1[i] = c1[i] * wB;
    }
    fft_wg2<N2, R, 1, -1, true>(c0, c1, lds, v, 0, tw);
#pragma unroll
    for (int i = 0; i < R; i++) {
        const cplx x0 = c0[i], x1 = c1[i];
        c0[i] = cfma(a01[i], x1, a00[i] * x0);
        c1[i] = cfma(a11[i], x1, a10[i] * x0);
    }
    fft_wg2<N2, R, 1, +1, true>(c0, c1, lds, v, 0, tw);
#pragma unroll
    for (int i = 0; i < R; i++) {
        const cplx w = cconj((i == 0) ? wb : wb * wu[i]);
        const size_t o = yz_index(G.N1, N2, k1, v + (N2 / R) * i);
        dZ[zstr + o] = c0[i] * w;         // C01
        dZ[3 * zstr + o] = c1[i] * w;     // C11
    }
}

// ---------------------------------------------------------------------------------------------
// Split levels with a column length N1 = 3*K, K a power of two.  The degrees of fnft_kdvv's default scheme are
// d = 12*2^l = 3*2^(l+2): with power-of-two transforms the folded length is the power of two ABOVE d (a third of every
// transform is padding); with N1 = 3*K columns of N2 = 1024-point rows the length is M = d exactly.  Only the column
// kernels see the factor 3 (decimation in time: three transforms of length K on the elements n1 = 3 m + r, then one
// radix-3 butterfly per bin with the twiddles w_N1^(r kappa)); the row kernel is unchanged.
//   lane (v, c): elements n1 = 3 (v + (K/R) i) + r in x[r][i]; bins k1 = (v + (K/R) i) + K t in x[t][i].
// ---------------------------------------------------------------------------------------------
template <int SIGN> FA_DEV void dft3(cplx &t0, cplx &t1, cplx &t2)
{
    const double h = 0.86602540378443864676;   // sqrt(3)/2
    const cplx s = t1 + t2, dl = t1 - t2;
    const cplx m = cmake(fma(-0.5, s.x, t0.x), fma(-0.5, s.y, t0.y));
    const cplx q = cmake(-dl.y * (h * (double)SIGN), dl.x * (h * (double)SIGN));   // SIGN * i * h * (t1 - t2)
    t0 = t0 + s;
    t1 = m + q;
    t2 = m - q;
}
// tw: table of exp(-2 pi i j/K); tw3: table of exp(-2 pi i j/(3K))
template <int K, int R, int BC> FA_DEV void fft3_fwd(cplx (&x)[3][R], cplx *lds, int v, int c, const cplx *tw, const cplx *tw3, int &parity)
{
#pragma unroll
    for (int r = 0; r < 3; r++) fft_wg<K, R, BC, -1, false, true>(x[r], lds, v, c, tw, parity);
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int kap = v + (K / R) * i;
        cplx t1 = x[1][i] * tw3[kap], t2 = x[2][i] * tw3[2 * kap];
        dft3<-1>(x[0][i], t1, t2);
        x[1][i] = t1;
        x[2][i] = t2;
    }
}
template <int K, int R, int BC> FA_DEV void fft3_inv(cplx (&x)[3][R], cplx *lds, int v, int c, const cplx *tw, const cplx *tw3, int &parity)
{
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int kap = v + (K / R) * i;
        dft3<+1>(x[0][i], x[1][i], x[2][i]);
        x[1][i] = x[1][i] * cconj(tw3[kap]);
        x[2][i] = x[2][i] * cconj(tw3[2 * kap]);
    }
#pragma unroll
    for (int r = 0; r < 3; r++) fft_wg<K, R, BC, +1, false, true>(x[r], lds, v, c, tw, parity);
}

// real coefficients -> column transform (first split level); M = N1*N2 = d exactly
//   grid.x = N2/BC tiles, grid.y = 4*n_in polynomials
template <int K, int R, int BC> FA_DEV void body_r3col_fwd(const BigLevel &G)
{
    constexpr int N1 = 3 * K;
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
    const double *src = (const double *)L.body_in + (size_t)e * L.plane + (size_t)mat * d;
    const double tl = ((const double *)L.tail_in)[(size_t)e * L.n_in + mat] * sc;
    cplx x[3][R];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int n1 = 3 * (v + (K / R) * i) + r;
            const long long idx = (long long)n1 * N2 + n2;
            const double re = src[idx] * sc;
            const double im = (idx == 0) ? tl : 0.0;   // the constant term, index d = M, folds onto 0
            const cplx w = G.twq[n1];
            x[r][i] = cmake(fma(re, w.x, im * w.y), fma(im, w.x, -(re * w.y)));
        }
    int parity = 0;
    fft3_fwd<K, R, BC>(x, lds, v, c, G.tw1, G.tw3, parity);
    cplx *dst = G.Y + (size_t)poly * N1 * N2;
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int i = 0; i < R; i++) dst[yz_index(N1, N2, v + (K / R) * i + K * t, n2)] = x[t][i];
}

// last split level: inverse column transform, untwist, unfold -> real coefficients
template <int K, int R, int BC> FA_DEV void body_r3col_inv(const BigLevel &G)
{
    constexpr int N1 = 3 * K;
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
    const cplx *src = G.Z + (size_t)poly * N1 * N2;
    const double sA = level_in_scale(L, 2 * P), sB = level_in_scale(L, 2 * P + 1);   // wave-uniform
    cplx x[3][R];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int i = 0; i < R; i++) x[t][i] = src[yz_index(N1, N2, v + (K / R) * i + K * t, n2)];
    int parity = 0;
    fft3_inv<K, R, BC>(x, lds, v, c, G.tw1, G.tw3, parity);
    double *dst = (double *)L.body_out + (size_t)e * L.plane + (size_t)P * (2 * M);
    const double inv = 1.0 / (double)N1;
    double m2 = 0.0;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int n1 = 3 * (v + (K / R) * i) + r;
            const long long idx = (long long)n1 * N2 + n2;
            const cplx val = (x[r][i] * inv) * G.twq[n1];
            double re = val.x;
            if (idx == 0) {
                const double tp = rsplit_tail_product(L, P, e, sA, sB);
                re += tp;   // coefficient 2d = 2M came back on 0 with a minus sign
                ((double *)L.tail_out)[(size_t)e * n_out + P] = tp;
                m2 = fmax(m2, tp * tp);
            }
            dst[idx] = re;
            dst[idx + M] = val.y;
            m2 = fmax(m2, fmax(re * re, val.y * val.y));
        }
    fa_wave_atomic_max_hi32(&L.max2_out[(size_t)P * kMax2Slots + max2_slot()], m2);   // P is uniform in the workgroup
}

// between two split levels: inverse column transform (3K), untwist, unfold; fold, twist, forward column transform (3*2K)
// of the next level.  The next level's sub-sequence r (elements n1' = 3 m' + r, m' < 2K) is [Re z_r[m] | Im z_r[m]]:
// the same lane holds both halves.
template <int K, int R, int BC> FA_DEV void body_r3bridge(const BigLevel &G)
{
    constexpr int N1 = 3 * K;
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
    const cplx *src = G.Z + (size_t)poly * N1 * N2;
    const double sA = level_in_scale(L, 2 * P), sB = level_in_scale(L, 2 * P + 1);   // wave-uniform
    cplx x[3][R];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int i = 0; i < R; i++) x[t][i] = src[yz_index(N1, N2, v + (K / R) * i + K * t, n2)];
    int parity = 0;
    fft3_inv<K, R, BC>(x, lds, v, c, G.tw1, G.tw3, parity);
    const double inv = 1.0 / (double)N1;
    double m2 = 0.0;
    cplx y[3][2 * R];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int n1 = 3 * (v + (K / R) * i) + r;
            const long long idx = (long long)n1 * N2 + n2;
            const cplx val = (x[r][i] * inv) * G.twq[n1];
            double re = val.x;
            const double im = val.y;   // coefficients idx and idx + M of the product
            double im0 = 0.0;
            if (idx == 0) {
                const double tp = rsplit_tail_product(L, P, e, sA, sB);
                re += tp;
                im0 = tp;   // the next level's constant term folds onto its element 0
                ((double *)L.tail_out)[(size_t)e * n_out + P] = tp;
                m2 = fmax(m2, tp * tp);
            }
            m2 = fmax(m2, fmax(re * re, im * im));
            const cplx w0 = G.twq2[n1], w1 = G.twq2[n1 + N1];   // exp(-2 pi i n1'/(8 N1)); the twist is the conjugate
            y[r][i] = cmake(fma(re, w0.x, im0 * w0.y), fma(im0, w0.x, -(re * w0.y)));
            y[r][R + i] = cmake(im * w1.x, -(im * w1.y));
        }
    fa_wave_atomic_max_hi32(&L.max2_out[(size_t)P * kMax2Slots + max2_slot()], m2);   // P is uniform in the workgroup
    fft3_fwd<2 * K, 2 * R, BC>(y, lds, v, c, G.tw1x2, G.tw3x2, parity);
    cplx *dst = G.Y + (size_t)poly * (2 * N1) * N2;
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int i = 0; i < 2 * R; i++) dst[yz_index(2 * N1, N2, v + (K / R) * i + 2 * K * t, n2)] = y[t][i];
}

// ---------------------------------------------------------------------------------------------
// Leaf of the real path for the even-order schemes: the coefficient matrices of S = DL/DEG consecutive samples AND
// their ordered product, by direct multiplication in registers -- levels 0 .. log2(S)-1 of the tree never touch HBM
// and need no transforms (for fnft_kdvv's default 2SPLIT8B: three launches of 100 + 100 MB each).  A step matrix of
// these schemes is sparse in z -- only the powers that are multiples of deg/n (B first) or deg/2n (A first) for some
// n <= T occur, 7 of 13 for 2SPLIT8B -- so the running product acc <- acc * U costs (support size) multiply-adds per
// coefficient and entry pair.  The rows of a 2x2 product are independent (new row r = old row r * U): two lanes per
// leaf matrix, one per row, each with its row's two polynomials (2 x (DL+1) doubles) in registers; every index is a
// compile-time constant after unrolling.  Rescaled like fnft__poly_fmult.c:330-374 (scale 1, exponent in wexp).
// ---------------------------------------------------------------------------------------------
template <int ORDER, bool BFIRST> constexpr bool rstrang_tap(int k)
{
    constexpr int T = RStrangCfg<ORDER, BFIRST>::T, DEG = RStrangCfg<ORDER, BFIRST>::DEG;
    for (int n = 1; n <= T; n++) {
        const int g = DEG / (BFIRST ? n : 2 * n);
        if (k % g == 0) return true;
    }
    return false;
}
template <int ORDER, bool BFIRST, int DL, int SIDX> struct RLeafStep {
    static constexpr int DEG = RStrangCfg<ORDER, BFIRST>::DEG;
    // acc (degree DEG*SIDX, row r: a0 = entry (r,0), a1 = entry (r,1)) <- acc * U, in place, highest power first:
    // new[k] = sum_t old[k-t] U[t]; descending k reads only indices that have not been overwritten
    static FA_DEV void mul(double (&a0)[DL + 1], double (&a1)[DL + 1], const double (&U)[4][DEG + 1])
    {
        constexpr int dold = DEG * SIDX, dnew = dold + DEG;
#pragma unroll
        for (int k = dnew; k >= 0; k--) {
            double n0 = 0.0, n1 = 0.0;
#pragma unroll
            for (int t = 0; t <= DEG; t++) {
                if (!rstrang_tap<ORDER, BFIRST>(t)) continue;
                const int kk = k - t;
                if (kk < 0 || kk > dold) continue;
                n0 = fma(a0[kk], U[0][t], fma(a1[kk], U[2][t], n0));
                n1 = fma(a0[kk], U[1][t], fma(a1[kk], U[3][t], n1));
            }
            a0[k] = n0;
            a1[k] = n1;
        }
    }
};
// The two lanes of a leaf matrix share the step matrices: in round k the lane of row r forms U of sample 2k + r and
// publishes it in LDS (ush: [matrix][2][4*(DEG+1)]), then both lanes multiply their row by the two matrices of the round.
template <int ORDER, bool BFIRST, int DL, int K> struct RLeafLoop {
    static constexpr int DEG = RStrangCfg<ORDER, BFIRST>::DEG, S = DL / DEG, W = 4 * (DEG + 1);
    static FA_DEV void load_u(const double *src, double (&U)[4][DEG + 1])
    {
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int t = 0; t <= DEG; t++) U[e][t] = src[e * (DEG + 1) + t];
    }
    static FA_DEV void run(const CoeffParams &P, bool act, int b, long long j0, int row, double *ush_m, double (&a0)[DL + 1],
                           double (&a1)[DL + 1])
    {
        if constexpr (2 * K < S) {
            {
                double U[4][DEG + 1];
                rstrang_sample<ORDER, BFIRST>(P, act, b, j0 + 2 * K + row, U);
                if (K > 0) FA_SYNC();   // the previous round's matrices have been read
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int t = 0; t <= DEG; t++) ush_m[row * W + e * (DEG + 1) + t] = U[e][t];
            }
            FA_SYNC();
            double U[4][DEG + 1];
            load_u(ush_m, U);
            if constexpr (K == 0) {
#pragma unroll
                for (int k = 0; k <= DL; k++) {
                    a0[k] = (k <= DEG) ? (row ? U[2][k <= DEG ? k : 0] : U[0][k <= DEG ? k : 0]) : 0.0;
                    a1[k] = (k <= DEG) ? (row ? U[3][k <= DEG ? k : 0] : U[1][k <= DEG ? k : 0]) : 0.0;
                }
            } else {
                RLeafStep<ORDER, BFIRST, DL, 2 * K>::mul(a0, a1, U);
            }
            load_u(ush_m + W, U);
            RLeafStep<ORDER, BFIRST, DL, 2 * K + 1>::mul(a0, a1, U);
            RLeafLoop<ORDER, BFIRST, DL, K + 1>::run(P, act, b, j0, row, ush_m, a0, a1);
        }
    }
};
// THREADS lanes = THREADS/2 leaf matrices per workgroup.  LDS: THREADS doubles (maxima) + THREADS * 2 * CH doubles (staging)
template <int ORDER, bool BFIRST, int DL, int THREADS> FA_DEV void body_rleaf_strang(const LeafParams &LP)
{
    constexpr int DEG = RStrangCfg<ORDER, BFIRST>::DEG, S = DL / DEG, CH = 16;
    static_assert(DL % DEG == 0 && DL % CH == 0 && S % 2 == 0, "leaf degree");
    FA_LDS_DECL
    double *mx = (double *)FA_LDS_PTR;        // THREADS
    double *stage = mx + THREADS;             // [lane][2][CH]; the same space holds the shared step matrices before
    double *ush = stage;                      // [matrix][2][4*(DEG+1)]
    const CoeffParams &P = LP.c;
    const int tid = FA_TID;
    const long long per = P.Dpad / S;                    // leaf matrices per signal
    const long long n_out = (long long)P.batch * per;
    const long long g0 = (long long)FA_BID * (THREADS / 2);
    const long long g = g0 + (tid >> 1);
    const int row = tid & 1;
    const bool act = g < n_out;
    const int b = act ? (int)(g / per) : 0;
    const long long j0 = act ? (g % per) * S : 0;
    double a0[DL + 1], a1[DL + 1];
    RLeafLoop<ORDER, BFIRST, DL, 0>::run(P, act, b, j0, row, ush + (size_t)(tid >> 1) * 8 * (DEG + 1), a0, a1);
    FA_SYNC();   // the shared matrices have been read: their space becomes the staging area
    // rescale: maximum over both rows (the two lanes of the matrix meet in LDS)
    double m2 = 0.0;
#pragma unroll
    for (int k = 0; k <= DL; k++) m2 = fmax(m2, fmax(a0[k] * a0[k], a1[k] * a1[k]));
    mx[tid] = m2;
    FA_SYNC();
    m2 = fmax(m2, mx[tid ^ 1]);
    int a = 0;
    double sc = 1.0;
    if (m2 > 0.0 && m2 < 1.0e300) {
        a = half_exponent(m2);
        sc = pow2i(-a);
    }
    if (act && row == 0) {
        P.scale[g] = 1.0;
        P.wexp[g] = a;
    }
    double *rb = (double *)P.body, *rt = (double *)P.tail;
    if (act) {
        rt[(size_t)(2 * row) * n_out + g] = a0[DL] * sc;
        rt[(size_t)(2 * row + 1) * n_out + g] = a1[DL] * sc;
    }
    const long long nvalid = (n_out - g0 < THREADS / 2) ? n_out - g0 : THREADS / 2;
    // bodies leave through LDS, CH coefficients of every entry at a time, so that lanes write runs of CH doubles
#pragma unroll
    for (int c0 = 0; c0 < DL; c0 += CH) {
        FA_SYNC();
#pragma unroll
        for (int kk = 0; kk < CH; kk++) {
            stage[(size_t)tid * 2 * CH + kk] = a0[c0 + kk] * sc;
            stage[(size_t)tid * 2 * CH + CH + kk] = a1[c0 + kk] * sc;
        }
        FA_SYNC();
        const int total = (int)nvalid * 4 * CH;
        for (int i = tid; i < total; i += THREADS) {
            const int kk = i % CH, m = (i / CH) % (int)nvalid, e = i / (CH * (int)nvalid);
            const int lane = 2 * m + (e >> 1);
            rb[(size_t)e * P.plane + (size_t)(g0 + m) * DL + c0 + kk] = stage[(size_t)lane * 2 * CH + (e & 1) * CH + kk];
        }
    }
}
