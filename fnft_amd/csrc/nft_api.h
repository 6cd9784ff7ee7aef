// nft_api.h -- host-pointer entry points (the reference's private-layer signatures) expressed
// over NftPlan<BE>.  Instantiated with the HIP back end in hip_backend.hip (product) and with
// the lane emulator in tests/emu (tests only).
#pragma once
#include "nft_plan.h"

// fnft__poly_fmult2x2, src/private/fnft__poly_fmult.c:381-546 (host buffers in and out)
template <class BE>
int api_poly_fmult2x2(BE &be, size_t *d, size_t n, std::complex<double> *p,
                      std::complex<double> *result, int32_t *W_ptr)
{
    if (!d || !p || !result || n == 0 || *d == 0) return NFT_EC_INVALID_ARGUMENT;
    const size_t deg = *d;
    if (deg > 0x7fffffff) return NFT_EC_INVALID_ARGUMENT;
    NftPlan<BE> pl(be, n, 0, 1, -1, (int)deg);
    int rc = pl.init();
    if (rc == NFT_SUCCESS) rc = pl.load_level0_from_host(p);
    if (rc == NFT_SUCCESS) rc = pl.run_tree();
    if (rc == NFT_SUCCESS) {
        pl.export_tm();
        const size_t cnt = 4 * (pl.res_deg + 1);
        be.d2h(result, pl.tm_out, cnt * sizeof(cplx));
        int W = 0;
        be.d2h(&W, pl.wexp[pl.cur], sizeof(int));
        rc = be.sync();
        if (rc == NFT_SUCCESS) {
            *d = pl.res_deg;
            if (W_ptr) {
                *W_ptr = W;
            } else {  // caller asked for un-normalised coefficients
                const double s = std::ldexp(1.0, W);
                for (size_t i = 0; i < cnt; i++) result[i] *= s;
            }
        }
    }
    pl.destroy();
    return rc;
}

// fnft__akns_fscatter, src/private/fnft__akns_fscatter.c:64-925 (r may be NULL: r = -kappa q*)
template <class BE>
int api_akns_fscatter(BE &be, size_t D, const std::complex<double> *q, const std::complex<double> *r,
                      double eps_t, int kappa, std::complex<double> *result, size_t *deg_ptr,
                      int32_t *W_ptr, int akns_disc)
{
    const int deg0 = nft_akns_degree(akns_disc);
    if (deg0 == 0) return NFT_EC_INVALID_ARGUMENT;
    NftPlan<BE> pl(be, D, 0, 1, akns_disc, deg0);
    int rc = pl.init();
    cplx *dq = nullptr, *dr = nullptr;
    if (rc == NFT_SUCCESS) {
        if (!pl.alloc(dq, D) || (r && !pl.alloc(dr, D))) rc = NFT_EC_NOMEM;
    }
    if (rc == NFT_SUCCESS) {
        be.h2d(dq, q, D * sizeof(cplx));
        if (r) be.h2d(dr, r, D * sizeof(cplx));
        rc = pl.run_coeffs(dq, dr, eps_t, kappa);
    }
    if (rc == NFT_SUCCESS) rc = pl.run_tree();
    if (rc == NFT_SUCCESS) {
        pl.export_tm();
        const size_t cnt = 4 * (pl.res_deg + 1);
        be.d2h(result, pl.tm_out, cnt * sizeof(cplx));
        int W = 0;
        be.d2h(&W, pl.wexp[pl.cur], sizeof(int));
        rc = pl.read_status();
        if (rc == -NFT_EC_OTHER) rc = NFT_EC_OTHER;  // raised by akns_fscatter itself
        if (rc == NFT_SUCCESS) {
            *deg_ptr = pl.res_deg;
            if (W_ptr) {
                *W_ptr = W;
            } else {
                const double s = std::ldexp(1.0, W);
                for (size_t i = 0; i < cnt; i++) result[i] *= s;
            }
        }
    }
    be.free(dq);
    be.free(dr);
    pl.destroy();
    return rc;
}

// fnft__kdv_fscatter, src/private/fnft__kdv_fscatter.c:45-83: r = -1 for every sample; a real potential takes the
// real-coefficient tree (nft_real.h)
template <class BE>
int api_kdv_fscatter(BE &be, size_t D, const std::complex<double> *u, double eps_t, std::complex<double> *result,
                     size_t *deg_ptr, int32_t *W_ptr, int akns_disc)
{
    const int deg0 = nft_akns_degree(akns_disc);
    if (deg0 == 0) return NFT_EC_INVALID_ARGUMENT;
    bool real = true;
    for (size_t i = 0; i < D && real; i++) real = (u[i].imag() == 0.0);
    NftPlan<BE> pl(be, D, 0, 1, akns_disc, deg0);
    pl.kdv = true;
    pl.want_real = real;
    int rc = pl.init();
    cplx *dq = nullptr;
    if (rc == NFT_SUCCESS && !pl.alloc(dq, D)) rc = NFT_EC_NOMEM;
    if (rc == NFT_SUCCESS) {
        be.h2d(dq, u, D * sizeof(cplx));
        rc = pl.run_coeffs(dq, pl.rneg, eps_t, 1);
    }
    if (rc == NFT_SUCCESS) rc = pl.run_tree();
    if (rc == NFT_SUCCESS) {
        pl.export_tm();
        const size_t cnt = 4 * (pl.res_deg + 1);
        be.d2h(result, pl.tm_out, cnt * sizeof(cplx));
        int W = 0;
        be.d2h(&W, pl.wexp[pl.cur], sizeof(int));
        rc = pl.read_status();
        if (rc == -NFT_EC_OTHER) rc = NFT_EC_OTHER;
        if (rc == NFT_SUCCESS) {
            *deg_ptr = pl.res_deg;
            if (W_ptr) {
                *W_ptr = W;
            } else {
                const double s = std::ldexp(1.0, W);
                for (size_t i = 0; i < cnt; i++) result[i] *= s;
            }
        }
    }
    be.free(dq);
    pl.destroy();
    return rc;
}
