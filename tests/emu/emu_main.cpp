// emu_main.cpp -- builds tests/emu/libfnft_emu.so: the kernel bodies of fnft_amd/csrc compiled
// for the CPU lane emulator (TEST INFRASTRUCTURE ONLY; see emu_backend.h).
#include "emu_backend.h"

thread_local fa_emu_ctx *fa_emu = nullptr;

#include "../../fnft_amd/csrc/nft_api.h"
#include "../../fnft_amd/csrc/nft_discspec.h"

// ---- stand-alone checks of fft_wg in every tiling the kernels use -----------------------------
struct FftTestParams {
    const cplx *in;   // B sequences of N, sequence c at in[c*N + idx]
    cplx *out;
    const cplx *tw;
    int sign;
};
template <int N, int R, int B, bool DB, int T> struct KFftTest {
    using Params = FftTestParams;
    static constexpr int THREADS = T;
    static constexpr size_t lds_bytes() { return (N > R) ? (size_t)(DB ? 2 : 1) * N * B * sizeof(cplx) : 0; }
    static void body(const Params &p)
    {
        cplx *lds = (cplx *)FA_LDS_PTR;
        const int tid = FA_TID;
        const int c = tid % B, v = tid / B;
        cplx x[R];
        for (int i = 0; i < R; i++) x[i] = p.in[(size_t)c * N + v + (N / R) * i];
        int parity = 0;
        // two transforms back to back exercise the buffer-parity hand-over
        if (p.sign < 0) {
            fft_wg<N, R, B, -1, DB>(x, lds, v, c, p.tw, parity);
        } else {
            fft_wg<N, R, B, -1, DB>(x, lds, v, c, p.tw, parity);
            fft_wg<N, R, B, +1, DB>(x, lds, v, c, p.tw, parity);
        }
        for (int i = 0; i < R; i++) p.out[(size_t)c * N + v + (N / R) * i] = x[i];
    }
};

static std::vector<cplx> host_tw(size_t N)
{
    std::vector<cplx> t(N);
    const long double tau = 6.283185307179586476925286766559005768L;
    for (size_t j = 0; j < N; j++) {
        const long double a = -tau * (long double)j / (long double)N;
        t[j] = cmake((double)cosl(a), (double)sinl(a));
    }
    return t;
}

template <int N> static int fft_pair_cfg(int sign, const cplx *in, cplx *out)
{
    EmuBackend be;
    using C = PairCfg<N>;
    auto tw = host_tw(N);
    FftTestParams p{in, out, tw.data(), sign};
    be.run<KFftTest<N, C::R, C::B, true, C::THREADS>>(1, 1, p);
    return C::B;
}
template <int N1> static int fft_col_cfg(int sign, const cplx *in, cplx *out)
{
    EmuBackend be;
    using C = ColCfg<N1>;
    auto tw = host_tw(N1);
    FftTestParams p{in, out, tw.data(), sign};
    be.run<KFftTest<N1, C::R, C::BC, C::DB, C::THREADS>>(1, 1, p);
    return C::BC;
}

extern "C" {

// returns the batch size B of the configuration (0: not instantiated); if in == NULL only reports B
int emu_fft_pair_cfg(int N, int sign, const cplx *in, cplx *out)
{
    switch (N) {
#define X(n) case n: return in ? fft_pair_cfg<n>(sign, in, out) : PairCfg<n>::B;
        X(8) X(16) X(32) X(64) X(128) X(256) X(512) X(1024) X(2048) X(4096)
#undef X
    default: return 0;
    }
}
int emu_fft_col_cfg(int N1, int sign, const cplx *in, cplx *out)
{
    switch (N1) {
#define X(n) case n: return in ? fft_col_cfg<n>(sign, in, out) : ColCfg<n>::BC;
        FA_FOR_EACH_N1(X)
#undef X
    default: return 0;
    }
}

int emu_poly_fmult2x2(size_t *d, size_t n, std::complex<double> *p, std::complex<double> *result,
                      int32_t *W_ptr)
{
    EmuBackend be;
    return api_poly_fmult2x2(be, d, n, p, result, W_ptr);
}

int emu_akns_fscatter(size_t D, const std::complex<double> *q, const std::complex<double> *r,
                      double eps_t, int kappa, std::complex<double> *result, size_t *deg_ptr,
                      int32_t *W_ptr, int akns_disc)
{
    EmuBackend be;
    return api_akns_fscatter(be, D, q, r, eps_t, kappa, result, deg_ptr, W_ptr, akns_disc);
}

int emu_poly_chirpz(size_t deg, const std::complex<double> *p, const double *A, const double *W,
                    size_t M, std::complex<double> *result)
{
    EmuBackend be;
    return NftPlan<EmuBackend>::chirpz_host(be, deg, p, {A[0], A[1]}, {W[0], W[1]}, M, result);
}

// fnft_nsev continuous spectrum, host buffers
int emu_nsev_contspec(size_t D, const std::complex<double> *q, const double *T, size_t M,
                      std::complex<double> *contspec, const double *XI, int kappa, int nse_disc,
                      int cstype, int normalization_flag)
{
    EmuBackend be;
    const int akns = nft_nse_to_akns(nse_disc);
    if (akns < 0) return NFT_EC_INVALID_ARGUMENT;
    const int ups = nft_nse_upsampling(nse_disc);
    NftPlan<EmuBackend> pl(be, (size_t)ups * D, M, 1, akns, nft_akns_degree(akns));
    pl.set_front(D, 1, ups);
    int rc = pl.init();
    if (rc != NFT_SUCCESS) { pl.destroy(); return rc; }
    const size_t cs_len = M * (cstype == 0 ? 1 : (cstype == 1 ? 2 : 3));
    cplx *dq = (cplx *)be.alloc(D * sizeof(cplx));
    cplx *dcs = (cplx *)be.alloc(cs_len * sizeof(cplx));
    be.h2d(dq, q, D * sizeof(cplx));
    double Tsub[2];
    rc = pl.run_front(dq, T, kappa, Tsub);
    if (rc == NFT_SUCCESS) rc = pl.run_tree();
    if (rc == NFT_SUCCESS) {
        NftPlan<EmuBackend>::Contspec cs;
        cs.T[0] = Tsub[0]; cs.T[1] = Tsub[1]; cs.XI[0] = XI[0]; cs.XI[1] = XI[1];
        cs.nse_disc = nse_disc; cs.cstype = cstype; cs.normalization_flag = normalization_flag;
        rc = pl.run_contspec(dcs, cs);
    }
    if (rc == NFT_SUCCESS) rc = pl.read_status();
    if (rc == NFT_SUCCESS) be.d2h(contspec, dcs, cs_len * sizeof(cplx));
    be.free(dq);
    be.free(dcs);
    pl.destroy();
    return rc;
}

// fnft_kdvv reflection coefficient, host buffers (kdv_disc 0..17)
// real != 0: the real-coefficient path (nft_real.h; u must be real)
int emu_kdvv_contspec(size_t D, const std::complex<double> *u, const double *T, size_t M,
                      std::complex<double> *contspec, const double *XI, int kdv_disc, int real)
{
    EmuBackend be;
    const int akns = kdv_disc + 1;
    NftPlan<EmuBackend> pl(be, D, M, 1, akns, nft_akns_degree(akns));
    pl.kdv = true;
    pl.want_real = real != 0;
    int rc = pl.init();
    if (rc != NFT_SUCCESS) { pl.destroy(); return rc; }
    cplx *du = (cplx *)be.alloc(D * sizeof(cplx));
    cplx *dcs = (cplx *)be.alloc(M * sizeof(cplx));
    be.h2d(du, u, D * sizeof(cplx));
    double Tsub[2];
    rc = pl.run_front(du, T, 1, Tsub);
    if (rc == NFT_SUCCESS) rc = pl.run_tree();
    if (rc == NFT_SUCCESS) rc = pl.run_contspec_kdv(dcs, T, XI, kdv_disc == 2);
    if (rc == NFT_SUCCESS) rc = pl.read_status();
    if (rc == NFT_SUCCESS) be.d2h(contspec, dcs, M * sizeof(cplx));
    be.free(du);
    be.free(dcs);
    pl.destroy();
    return rc;
}

// discrete spectrum of fnft_nsev, host buffers (capacity *K_ptr in, count out)
int emu_nsev_discspec(size_t D, const std::complex<double> *q, const double *T, int bsfilt, int bsloc,
                      size_t niter, size_t Dsub, int dstype, int nse_disc, int richardson, size_t *K_ptr,
                      std::complex<double> *bound_states, std::complex<double> *nc)
{
    EmuBackend be;
    NftDiscSpec<EmuBackend> ds(be);
    NftDsOpts o;
    o.bsfilt = bsfilt; o.bsloc = bsloc; o.niter = niter; o.Dsub = Dsub; o.dstype = dstype;
    o.nse_disc = nse_disc; o.richardson = richardson;
    return ds.run(D, q, T, o, K_ptr, bound_states, nc);
}

// all roots of a polynomial (fnft__poly_roots_fasteigen seam): the Ehrlich-Aberth kernels in the emulator.
// *last_corr: largest relative correction of the last sweep
int emu_poly_roots(size_t deg, const std::complex<double> *p, std::complex<double> *roots, double *last_corr)
{
    EmuBackend be;
    NftDiscSpec<EmuBackend> ds(be);
    cplx *d_coef = (cplx *)be.alloc((deg + 1) * sizeof(cplx));
    be.h2d(d_coef, p, (deg + 1) * sizeof(cplx));
    std::vector<std::complex<double>> z;
    const int rc = ds.roots(d_coef, deg, z);
    be.free(d_coef);
    if (last_corr) *last_corr = ds.last_root_corr;
    for (size_t i = 0; i < z.size() && i < deg; i++) roots[i] = z[i];
    return rc;
}

}  // extern "C"
