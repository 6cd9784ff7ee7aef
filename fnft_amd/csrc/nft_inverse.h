// nft_inverse.h -- fast inverse scattering by layer peeling, the other consumer of the pair-product kernels
// (first slice).
//
// Restates src/private/fnft__nse_finvscatter.c:66-232 (recursion) and :234-366 (driver): given the transfer
// matrix T(z) of D = 2^k samples (discretizations with one sample per step and degree 1: 2SPLIT2_MODAL, 2SPLIT2A),
// recover the samples.  With T = T2 * T1 (T1 from samples 0..D/2-1, T2 from D/2..D-1):
//   1. the LOWER half of T's coefficients determines T2's inverse (up to a power of z) and samples D/2..D-1,
//   2. T1 = T2i * T  (one 2x2 polynomial product of degree deg),
//   3. the middle coefficients of T1 determine T1i and samples 0..D/2-1,
//   4. Ti = T1i * T2i when the caller one level up needs it.
// The reference runs this with an explicit stack; plain recursion (depth log2 D) is used here.  Products go
// through `Product` -- the GPU tree (api_poly_fmult2x2, n = 2) above a degree threshold, direct convolution on
// the host below it, where a device round trip costs more than the product.
#pragma once
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

template <class Product> class NftLayerPeeling {
public:
    typedef std::complex<double> cd;
    Product &prod;
    double eps_t;
    int kappa;
    int modal;        // 1: 2SPLIT2_MODAL, 0: 2SPLIT2A
    int rc = 0;       // first error (5 = FNFT_EC_OTHER: a reconstructed sample violates |q| < 1, :173-176)

    NftLayerPeeling(Product &p, double eps, int kap, int is_modal) : prod(p), eps_t(eps), kappa(kap), modal(is_modal) {}

    // T: four entries of deg+1 coefficients (highest power first) at stride T_stride; Ti (may be NULL): receives
    // the inverse up to a power of z, four entries of deg+1 at stride Ti_stride; q: deg samples out
    void peel(size_t deg, const cd *T, size_t T_stride, cd *Ti, size_t Ti_stride, cd *q)
    {
        if (rc) return;
        if (deg == 1) { base(T, T_stride, Ti, Ti_stride, q); return; }
        const size_t h = deg / 2;
        std::vector<cd> T2i(4 * (deg + 1), cd(0, 0)), T1(4 * (2 * deg + 1)), T1i(4 * (h + 1));
        // step 1 (:107-116): lower half of T -> T2i (stored in the lower half of a degree-deg array), q[h..]
        peel(h, T + h, T_stride, T2i.data() + h, deg + 1, q + h);
        if (rc) return;
        // step 2 (:120-127): T1 = T2i * T
        rc = prod(deg, T2i.data(), deg + 1, T, T_stride, T1.data(), 2 * deg + 1);
        if (rc) return;
        // step 3 (:131-140): coefficients deg .. deg+h of T1 -> T1i, q[0..h)
        peel(h, T1.data() + deg, 2 * deg + 1, T1i.data(), h + 1, q);
        if (rc) return;
        // step 4 (:144-156): Ti = T1i * T2i
        if (Ti) rc = prod(h, T1i.data(), h + 1, T2i.data() + h, deg + 1, Ti, Ti_stride);
    }

private:
    // one sample, :158-218
    void base(const cd *T, size_t T_stride, cd *Ti, size_t Ti_stride, cd *q)
    {
        const cd *T11 = T, *T21 = T + 2 * T_stride;
        const cd Q = -(double)kappa * std::conj(T21[1] / T11[1]);
        const double absQ = std::abs(Q);
        const double den = 1.0 + (double)kappa * absQ * absQ;
        if (den <= 0.0) { rc = 5; return; }
        const double scl = 1.0 / std::sqrt(den);
        if (modal) *q = Q / eps_t;                                                 // :177
        else *q = std::atan(absQ) * std::exp(cd(0.0, std::arg(Q))) / eps_t;        // :196
        if (!Ti) return;
        cd *Ti11 = Ti, *Ti12 = Ti + Ti_stride, *Ti21 = Ti + 2 * Ti_stride, *Ti22 = Ti + 3 * Ti_stride;
        Ti11[0] = scl;                 Ti11[1] = 0.0;
        Ti12[0] = -scl * Q;            Ti12[1] = 0.0;
        Ti21[0] = 0.0;                 Ti21[1] = scl * (double)kappa * std::conj(Q);
        Ti22[0] = 0.0;                 Ti22[1] = scl;
    }
};

// direct 2x2 polynomial product on the host (small degrees)
inline void nft_host_product2x2(size_t deg, const std::complex<double> *A, size_t As, const std::complex<double> *B,
                                size_t Bs, std::complex<double> *C, size_t Cs)
{
    typedef std::complex<double> cd;
    for (int e = 0; e < 4; e++) {
        const cd *a0 = A + (size_t)(e / 2) * 2 * As, *a1 = a0 + As;       // row of A: entries (r,0), (r,1)
        const cd *b0 = B + (size_t)(e % 2) * Bs, *b1 = b0 + 2 * Bs;       // column of B: entries (0,c), (1,c)
        cd *c = C + (size_t)e * Cs;
        for (size_t k = 0; k <= 2 * deg; k++) c[k] = cd(0, 0);
        for (size_t i = 0; i <= deg; i++) {
            const cd x0 = a0[i], x1 = a1[i];
            for (size_t j = 0; j <= deg; j++) c[i + j] += x0 * b0[j] + x1 * b1[j];
        }
    }
}
