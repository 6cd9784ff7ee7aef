// nft_nsev_inverse.h -- device side of fnft_nsev_inverse (src/fnft_nsev_inverse.c:121-1033) and of
// fnft__poly_specfact (src/private/fnft__poly_specfact.c:25-140).  The host keeps the reference's control flow;
// what runs on the GPU: every DFT (chirp kernels in DFT mode: any length, so the reference's 2-3-5-smooth lengths are
// kept), every element-wise stage with transcendentals (body_inv_op), the layer peeling's polynomial products
// (fnft__nse_finvscatter), the multi-soliton recursion, the eigenfunctions of the seed potential (the chunk-parallel
// scatterer of the discrete spectrum on the half-step signal) and the Darboux steps.
#pragma once
#include <algorithm>
#include <complex>
#include <cstring>
#include <map>
#include <new>
#include <vector>

#include "nft_plan.h"
#include "nft_discspec.h"

// kiss_fft_next_fast_size (fft_wrapper_next_fft_length): smallest m >= n with only the factors 2, 3, 5
inline size_t nft_next_fast_size(size_t n)
{
    for (;; n++) {
        size_t m = n;
        while (m % 2 == 0) m /= 2;
        while (m % 3 == 0) m /= 3;
        while (m % 5 == 0) m /= 5;
        if (m <= 1) return n;
    }
}

// Layer peeling with every array on the device (src/private/fnft__nse_finvscatter.c:66-232): the recursion of
// (steps 1-4 below) issued from the host without a single synchronisation -- blocks of up to kLeaf samples by the
// one-wave leaf kernel (body_peel_leaf), the 2x2 polynomial products above that by resident plans (one per degree,
// n = 2 matrices: the tree's general pair kernels), factors and results moved between the caller's strided arrays
// and the plans' layout by two small kernels.  Work arrays: one set per recursion depth (depth-first order on one
// stream, so siblings reuse them).
template <class BE> class NftLayerPeelingDev {
public:
    static constexpr size_t kLeaf = 256;
    static constexpr size_t kOneLaunchMaxDeg = 1024;   // products up to this degree: KPeelProduct
    BE &be;
    double eps_t;
    int kappa, modal;
    int rc = NFT_SUCCESS;
    int *d_status = nullptr;
    std::map<size_t, NftPlan<BE> *> plans;
    struct Work { cplx *T2i = nullptr, *T1 = nullptr, *T1i = nullptr; };
    std::vector<Work> work;

    NftLayerPeelingDev(BE &b, double eps, int kap, int is_modal) : be(b), eps_t(eps), kappa(kap), modal(is_modal) {}
    ~NftLayerPeelingDev() { destroy(); }
    void destroy()
    {
        for (auto &kv : plans) { kv.second->destroy(); delete kv.second; }
        plans.clear();
        for (auto &w : work) { be.free(w.T2i); be.free(w.T1); be.free(w.T1i); }
        work.clear();
        work_deg = 0;
        be.free(d_status);
        d_status = nullptr;
    }
    // status word, work arrays for a transfer matrix of degree deg (kept and reused by later calls of the same or a
    // smaller size: depth 0 is always the largest)
    size_t work_deg = 0;
    int init(size_t deg)
    {
        rc = NFT_SUCCESS;
        if (!d_status) {
            d_status = (int *)be.alloc(4 * sizeof(int));
            if (!d_status) return NFT_EC_NOMEM;
        }
        be.memset0(d_status, 4 * sizeof(int));
        if (deg > work_deg) {
            for (auto &w : work) { be.free(w.T2i); be.free(w.T1); be.free(w.T1i); }
            work.clear();
            work_deg = 0;
            for (size_t d = deg; d > kLeaf; d /= 2) {
                Work w;
                w.T2i = (cplx *)be.alloc(4 * (d + 1) * sizeof(cplx));
                w.T1 = (cplx *)be.alloc(4 * (2 * d + 1) * sizeof(cplx));
                w.T1i = (cplx *)be.alloc(4 * (d / 2 + 1) * sizeof(cplx));
                work.push_back(w);
                if (!w.T2i || !w.T1 || !w.T1i) return NFT_EC_NOMEM;
                // the upper half of T2i (coefficients 0 .. d/2 - 1 of every entry) is never written: zero once
                be.memset0(w.T2i, 4 * (d + 1) * sizeof(cplx));
            }
            work_deg = deg;
        }
        return NFT_SUCCESS;
    }
    NftPlan<BE> *plan_for(size_t deg)
    {
        auto it = plans.find(deg);
        if (it != plans.end()) return it->second;
        NftPlan<BE> *pl = new (std::nothrow) NftPlan<BE>(be, 2, 0, 1, -1, (int)deg);
        if (!pl) { rc = NFT_EC_NOMEM; return nullptr; }
        const int r = pl->init();
        if (r != NFT_SUCCESS) { rc = r; pl->destroy(); delete pl; return nullptr; }
        plans[deg] = pl;
        return pl;
    }
    // C (four entries of 2 deg + 1 coefficients at stride Cs) = A * B (four entries of deg + 1 at strides As, Bs)
    void prod(size_t deg, const cplx *A, size_t As, const cplx *B, size_t Bs, cplx *C, size_t Cs)
    {
        if (rc != NFT_SUCCESS) return;
        if (deg == 256 || deg == 512 || deg == kOneLaunchMaxDeg) {   // one launch, entries transformed concurrently
            NftPlan<BE> *tp = plan_for(kLeaf * 2);                  // any resident plan: its twiddle tables
            if (!tp) return;
            PeelProdParams Q;
            std::memset(&Q, 0, sizeof(Q));
            Q.A = A; Q.B = B; Q.As = (long long)As; Q.Bs = (long long)Bs; Q.C = C; Q.Cs = (long long)Cs;
            Q.tw = tp->tw_table(2 * deg);
            if (deg == 256) be.template run<KPeelProduct<512>>(1, 1, Q);
            else if (deg == 512) be.template run<KPeelProduct<1024>>(1, 1, Q);
            else be.template run<KPeelProduct<2048>>(1, 1, Q);
            return;
        }
        NftPlan<BE> *pl = plan_for(deg);
        if (!pl) return;
        PeelIoParams P;
        std::memset(&P, 0, sizeof(P));
        P.A = A; P.B = B; P.As = (long long)As; P.Bs = (long long)Bs; P.d = (long long)deg;
        P.body = pl->body[0]; P.tail = pl->tail[0]; P.scale = pl->scale[0]; P.wexp = pl->wexp[0];
        be.template run<KPeelImport>((int)((8 * (deg + 1) + 255) / 256), 1, P);
        pl->cur = 0;
        pl->ne = 4;
        pl->start_n = pl->n0;
        pl->start_d = deg;
        const int r = pl->run_tree();
        if (r != NFT_SUCCESS) { rc = r; return; }
        pl->export_tm(C, Cs, true);     // result layout, un-normalised, straight into the caller's strided array
    }
    // T: four entries of deg+1 coefficients at stride Ts; Ti (may be NULL): the inverse up to a power of z, four
    // entries of deg+1 at stride Tis; q: deg samples.  All device pointers.
    void peel(size_t deg, const cplx *T, size_t Ts, cplx *Ti, size_t Tis, cplx *q, size_t depth = 0)
    {
        if (rc != NFT_SUCCESS) return;
        if (deg <= kLeaf) {
            PeelLeafParams L;
            std::memset(&L, 0, sizeof(L));
            L.T = T; L.Ts = (long long)Ts; L.d = (int)deg; L.Ti = Ti; L.Tis = (long long)Tis; L.q = q;
            L.eps_t = eps_t; L.kappa = kappa; L.modal = modal; L.status = d_status;
            be.template run<KPeelLeaf>(1, 1, L);
            return;
        }
        const size_t h = deg / 2;
        size_t slot = 0;                                                      // work[0] serves degree work_deg
        for (size_t d = work_deg; d > deg; d /= 2) slot++;
        (void)depth;
        Work &w = work[slot];
        // w.T2i: the child below fills coefficients h .. deg of every entry; 0 .. h-1 are zero since init()
        peel(h, T + h, Ts, w.T2i + h, deg + 1, q + h, depth + 1);             // step 1, :107-116
        prod(deg, w.T2i, deg + 1, T, Ts, w.T1, 2 * deg + 1);                  // step 2, :120-127
        peel(h, w.T1 + deg, 2 * deg + 1, w.T1i, h + 1, q, depth + 1);         // step 3, :131-140
        if (Ti) prod(h, w.T1i, h + 1, w.T2i + h, deg + 1, Ti, Tis);           // step 4, :144-156
    }
    // host-pointer driver: tm (4*(deg+1)) -> q[deg]
    int run_host(size_t deg, const std::complex<double> *tm, std::complex<double> *q)
    {
        int r = init(deg);
        cplx *dT = (cplx *)be.alloc(4 * (deg + 1) * sizeof(cplx)), *dq = (cplx *)be.alloc(deg * sizeof(cplx));
        if (r == NFT_SUCCESS && (!dT || !dq)) r = NFT_EC_NOMEM;
        if (r == NFT_SUCCESS) {
            be.h2d(dT, tm, 4 * (deg + 1) * sizeof(cplx));
            peel(deg, dT, deg + 1, nullptr, 0, dq);
            r = rc;
        }
        if (r == NFT_SUCCESS) {
            int hst[4] = {0, 0, 0, 0};
            be.d2h(q, dq, deg * sizeof(cplx));
            be.d2h(hst, d_status, sizeof(hst));
            r = be.sync();
            if (r == NFT_SUCCESS && (hst[0] & 48)) r = NFT_EC_OTHER;      // :173-176 (bit 5: a leaf gave up waiting)
        }
        be.free(dT); be.free(dq);
        return r;      // plans and work arrays stay for the next call (destroy() releases them)
    }
};

template <class BE> class NftInverseDev {
public:
    typedef std::complex<double> cd;
    BE &be;
    NftPlan<BE> pl;
    size_t Lcap = 0;
    cplx *dY = nullptr, *dV = nullptr;
    int *dstatus = nullptr;
    double *dacc = nullptr;
    size_t acc_cap = 0;

    explicit NftInverseDev(BE &b) : be(b), pl(b, 2, 0, 1, 0, 1) {}   // only the twiddle tables of the plan are used
    ~NftInverseDev() { destroy(); }

    static size_t dft_L(size_t n)
    {
        size_t L = nft_nextpow2(2 * n - 1);
        if (L < 2 * (size_t)kRowChirp) L = 2 * (size_t)kRowChirp;
        return L;
    }
    // workspace for DFTs of up to nmax points
    int init(size_t nmax)
    {
        const size_t L = dft_L(nmax);
        if (L > kMaxSplitChirp) return NFT_EC_NOT_YET_IMPLEMENTED;
        bool ok = pl.alloc(pl.twtab, (size_t)2 * kMaxTwTable) && pl.alloc(pl.twlo, kTwLoEntries);
        ok = ok && pl.alloc(dY, L) && pl.alloc(dV, L) && pl.alloc(dstatus, 4);
        acc_cap = (nmax + 255) / 256;
        ok = ok && pl.alloc(dacc, acc_cap);
        if (!ok) return NFT_EC_NOMEM;
        Lcap = L;
        pl.upload_twiddles();
        be.memset0(dstatus, 4 * sizeof(int));
        return NFT_SUCCESS;
    }
    void destroy()
    {
        be.free(dY); be.free(dV); be.free(dstatus); be.free(dacc);
        be.free(pl.twtab); be.free(pl.twlo);
        dY = dV = nullptr; dstatus = nullptr; dacc = nullptr;
        pl.twtab = nullptr; pl.twlo = nullptr;
    }
    // out[k] = sum_n in[n] exp(sign * 2 pi i n k / n), un-normalised (fft_wrapper_execute_plan)
    int dft(const cplx *d_in, cplx *d_out, size_t n, int sign)
    {
        const size_t L = dft_L(n);
        if (L > Lcap) return NFT_EC_OTHER;
        ChirpParams C;
        std::memset(&C, 0, sizeof(C));
        C.poly = d_in;
        C.deg = (long long)n - 1;
        C.batch = 1;
        C.npoly = 1;
        C.M = (long long)n;
        C.Ybuf = dY; C.Vbuf = dV; C.Hbuf = d_out;
        pl.fill_chirp_geometry(C, L);
        C.status = dstatus;
        C.cstype = -1;
        C.dft_len = (long long)n;
        C.dft_sign = sign;
        return pl.run_chirp(C);
    }
    void op(int code, size_t n, const cplx *a, const cplx *b, cplx *out, cplx *out2 = nullptr, double s0 = 0, double s1 = 0,
            double s2 = 0, long long i0 = 0, int kappa = 0, int K = 0, const cplx *bs = nullptr)
    {
        InvOpParams P;
        std::memset(&P, 0, sizeof(P));
        P.op = code; P.n = (long long)n; P.a = a; P.b = b; P.out = out; P.out2 = out2;
        P.s0 = s0; P.s1 = s1; P.s2 = s2; P.i0 = i0; P.kappa = kappa; P.K = K; P.bs = bs;
        P.status = dstatus; P.accum = dacc;
        be.template run<KInvOp>((int)((n + 255) / 256), 1, P);
    }

    // fnft__poly_specfact.c:25-140 on device arrays: d_poly deg+1 coefficients -> d_result deg+1 coefficients;
    // w3: three work arrays of next_fast_size((deg+1)*oversampling) elements.  *warn: ill-posed problem (:109-110)
    static size_t specfact_len(size_t deg, size_t oversampling) { return nft_next_fast_size((deg + 1) * oversampling); }
    int specfact(size_t deg, const cplx *d_poly, cplx *d_result, size_t oversampling, int kappa, cplx *w_in, cplx *w_out,
                 cplx *w_x, int *warn)
    {
        const size_t Mf = specfact_len(deg, oversampling);
        be.memset0(dstatus, 4 * sizeof(int));
        op(INV_PAD, Mf, d_poly, nullptr, w_in, nullptr, 0, 0, 0, (long long)deg);
        int rc = dft(w_in, w_out, Mf, -1);
        if (rc != NFT_SUCCESS) return rc;
        op(INV_SPEC_X, Mf, w_out, nullptr, w_x, nullptr, 0, 0, 0, 0, kappa);
        rc = dft(w_x, w_in, Mf, -1);                     // Hilbert transform of x, :112-125
        if (rc != NFT_SUCCESS) return rc;
        op(INV_HILBERT, Mf, w_in, nullptr, w_in);
        rc = dft(w_in, w_out, Mf, +1);
        if (rc != NFT_SUCCESS) return rc;
        op(INV_SPEC_RESP, Mf, w_x, w_out, w_in);         // exp(x - i y)/M, :129-130
        rc = dft(w_in, w_out, Mf, +1);
        if (rc != NFT_SUCCESS) return rc;
        op(INV_REV_CONJ, deg + 1, w_out, nullptr, d_result, nullptr, 0, 0, 0, (long long)deg);
        int hst[4] = {0, 0, 0, 0};
        be.d2h(hst, dstatus, sizeof(hst));
        rc = be.sync();
        if (warn) *warn = (hst[0] & 8) ? 1 : 0;
        return rc;
    }
    // host-pointer form (the exported fnft__poly_specfact)
    int specfact_host(size_t deg, const cd *poly, cd *result, size_t oversampling, int kappa, int *warn)
    {
        const size_t Mf = specfact_len(deg, oversampling);
        int rc = init(Mf);
        if (rc != NFT_SUCCESS) return rc;
        cplx *dp = nullptr, *dr = nullptr, *w0 = nullptr, *w1 = nullptr, *w2 = nullptr;
        const bool ok = pl.alloc(dp, deg + 1) && pl.alloc(dr, deg + 1) && pl.alloc(w0, Mf) && pl.alloc(w1, Mf) && pl.alloc(w2, Mf);
        rc = ok ? NFT_SUCCESS : NFT_EC_NOMEM;
        if (ok) {
            be.h2d(dp, poly, (deg + 1) * sizeof(cplx));
            rc = specfact(deg, dp, dr, oversampling, kappa, w0, w1, w2, warn);
            if (rc == NFT_SUCCESS) {
                be.d2h(result, dr, (deg + 1) * sizeof(cplx));
                rc = be.sync();
            }
        }
        be.free(dp); be.free(dr); be.free(w0); be.free(w1); be.free(w2);
        return rc;
    }

    // ---- continuous part: transfer matrix from the given representation, :302-676 ------------------------------
    // contspec (M values, host) is modified as the reference modifies its argument; tm: 4*(deg+1) host values.
    // cstype 0 rho, 1 b(xi), 2 B(tau); method 0/1 A = 1 (:302-369), 2 iteration (:375-508).  pf: the phase factor
    // the host formed (fnft__nse_discretization.c:240-258 / :319-379)
    int transfer_matrix(size_t M, cd *contspec, const double *XI, size_t K, const cd *bound_states, size_t D,
                        const double *T, size_t deg, cd *tm, int kappa, int cstype, int method, size_t max_iter,
                        size_t oversampling, double pf, int *warn_specfact, int *warn_maxiter)
    {
        *warn_specfact = 0;
        *warn_maxiter = 0;
        const size_t os = (cstype == 0) ? 32 : oversampling;
        const bool need_sf = (cstype != 0) || method == 2;
        const size_t sf_deg = (cstype == 1) ? deg : D - 1;
        const size_t Mf = need_sf ? specfact_len(sf_deg, os) : 0;
        int rc = init(std::max(M, Mf));
        if (rc != NFT_SUCCESS) return rc;
        const double eps_t = (T[1] - T[0]) / (double)(D - 1);
        cplx *dc = nullptr, *dr = nullptr, *db = nullptr, *da = nullptr, *dbs = nullptr;
        cplx *w0 = nullptr, *w1 = nullptr, *w2 = nullptr;
        bool ok = pl.alloc(dc, M) && pl.alloc(dr, M) && pl.alloc(db, std::max(M, deg + 1)) && pl.alloc(da, deg + 1)
                  && pl.alloc(dbs, K ? K : 1);
        if (need_sf) ok = ok && pl.alloc(w0, Mf) && pl.alloc(w1, Mf) && pl.alloc(w2, Mf);
        rc = ok ? NFT_SUCCESS : NFT_EC_NOMEM;
        std::vector<cd> hb, ha;
        for (size_t i = 0; i < 4 * (deg + 1); i++) tm[i] = 0.0;
        if (rc == NFT_SUCCESS && cstype != 2) {
            // :251-296 (and :1013-1033 for the reflection coefficient): phases off, FFT order
            const double eps_xi = (XI[1] - XI[0]) / (double)(M - 1);
            be.h2d(dc, contspec, M * sizeof(cplx));
            const size_t Kp = (cstype == 0) ? K : 0;
            if (Kp) be.h2d(dbs, bound_states, Kp * sizeof(cplx));
            op(INV_PREP, M, dc, nullptr, dc, dr, XI[0], eps_xi, pf, 0, 0, (int)Kp, dbs);
            be.d2h(contspec, dc, M * sizeof(cplx));
            rc = be.sync();
        }
        if (rc == NFT_SUCCESS && (cstype == 1 || (cstype == 0 && method != 2))) {
            rc = dft(dr, db, M, -1);                                       // B(z) from an M-point FFT, :340 / :594
            hb.resize(M);
            if (rc == NFT_SUCCESS) { be.d2h(hb.data(), db, M * sizeof(cplx)); rc = be.sync(); }
            if (rc == NFT_SUCCESS) {
                const size_t i0 = (deg <= M - 1) ? 0 : deg - (M - 1);
                const double invM = 1.0 / (double)M;
                for (size_t i = i0; i <= deg; i++) {
                    tm[1 * (deg + 1) + i] = -(double)kappa * std::conj(hb[M - 1 - deg + i] * invM);
                    tm[2 * (deg + 1) + i] = hb[deg - i] * invM;
                }
                if (cstype == 0) {                                         // A(z) = 1, :363-364
                    tm[deg] = 1.0;
                    tm[3 * (deg + 1)] = 1.0;
                } else {                                                   // A by spectral factorization, :615-620
                    be.h2d(db, tm + 2 * (deg + 1), (deg + 1) * sizeof(cplx));
                    rc = specfact(deg, db, da, os, kappa, w0, w1, w2, warn_specfact);
                    if (rc == NFT_SUCCESS) { be.d2h(tm, da, (deg + 1) * sizeof(cplx)); rc = be.sync(); }
                    for (size_t i = 0; i <= deg; i++) tm[3 * (deg + 1) + i] = tm[deg - i];
                }
            }
        } else if (rc == NFT_SUCCESS && cstype == 0) {
            // Algorithm 1 of arXiv:1607.01305v2, :375-508 (M = D = deg, defocusing; checked by the host)
            double prev_change = INFINITY, prev_diff = INFINITY;
            size_t iter = 0;
            std::vector<double> part((D + 255) / 256);
            for (; iter < max_iter && rc == NFT_SUCCESS; iter++) {
                op(INV_ITER_FIN, D, dr, nullptr, db, nullptr, 0, 0, 0, 0, kappa);
                rc = dft(db, w0, D, -1);
                if (rc != NFT_SUCCESS) break;
                op(INV_REVERSE, D, w0, nullptr, db);                                  // b_coeffs
                int w = 0;
                rc = specfact(D - 1, db, da, 32, kappa, w0, w1, w2, &w);               // a_coeffs
                if (rc != NFT_SUCCESS) break;
                *warn_specfact |= w;
                op(INV_REVERSE, D, da, nullptr, w0);
                rc = dft(w0, w1, D, +1);
                if (rc != NFT_SUCCESS) break;
                op(INV_ITER_PHASE, D, w1, dc, dr);
                be.d2h(part.data(), dacc, part.size() * sizeof(double));
                rc = be.sync();
                if (rc != NFT_SUCCESS) break;
                double cur = 0.0;
                for (double v : part) cur += v;
                cur /= (double)D;
                const double diff = std::fabs(cur - prev_change);
                if (diff < 10.0 * 2.220446049250313e-16) break;
                prev_change = cur;
                if (diff > 0.9 * prev_diff) break;
                prev_diff = diff;
            }
            if (iter == max_iter) *warn_maxiter = 1;
            if (rc == NFT_SUCCESS) {
                ha.resize(D); hb.resize(D);
                be.d2h(ha.data(), da, D * sizeof(cplx));
                be.d2h(hb.data(), db, D * sizeof(cplx));
                rc = be.sync();
            }
            if (rc == NFT_SUCCESS)
                for (size_t i = 0; i < D; i++) {
                    tm[1 + i] = ha[i];
                    tm[1 * (deg + 1) + i] = -(double)kappa * std::conj(hb[D - 1 - i]);
                    tm[2 * (deg + 1) + 1 + i] = hb[i];
                    tm[3 * (deg + 1) + i] = ha[D - 1 - i];
                }
        } else if (rc == NFT_SUCCESS) {
            // B(tau), :632-676 (degree1step = 1 for both admissible discretizations)
            be.h2d(dc, contspec, D * sizeof(cplx));
            op(INV_BTAU, D, dc, nullptr, db, nullptr, eps_t);
            rc = specfact(D - 1, db, da, os, kappa, w0, w1, w2, warn_specfact);
            if (rc == NFT_SUCCESS) {
                ha.resize(D); hb.resize(D);
                be.d2h(ha.data(), da, D * sizeof(cplx));
                be.d2h(hb.data(), db, D * sizeof(cplx));
                rc = be.sync();
            }
            if (rc == NFT_SUCCESS)
                for (size_t i = 0; i < D; i++) {
                    tm[1 + i] = ha[i];
                    tm[2 * (deg + 1) + 1 + i] = hb[i];
                    tm[1 * (deg + 1) + i] = -(double)kappa * std::conj(hb[D - 1 - i]);
                    tm[3 * (deg + 1) + i] = ha[D - 1 - i];
                }
        }
        be.free(dc); be.free(dr); be.free(db); be.free(da); be.free(dbs); be.free(w0); be.free(w1); be.free(w2);
        return rc;
    }

    // ---- discrete part, :680-903 ------------------------------------------------------------------------------
    // bs / nc: sorted bound states and NORMING CONSTANTS (the host converted residues); mode 0: pure solitons
    // (:797-842), 1: Darboux steps on the seed q (:843-889)
    int add_discrete(size_t K, const cd *bs, const cd *nc, size_t D, cd *q, const double *T, int mode)
    {
        const double eps_t = (T[1] - T[0]) / (double)(D - 1);
        InvDsParams P;
        std::memset(&P, 0, sizeof(P));
        cplx *dbs = nullptr, *dnc = nullptr, *dq = nullptr, *work = nullptr;
        bool ok = pl.alloc(dbs, K) && pl.alloc(dnc, K) && pl.alloc(dq, D) && pl.alloc(work, (mode ? 2 : 1) * K * D);
        int rc = ok ? NFT_SUCCESS : NFT_EC_NOMEM;
        cplx *dq2 = nullptr, *cm = nullptr, *bnd = nullptr, *bndp = nullptr, *PHI = nullptr, *PSI = nullptr, *dout = nullptr;
        if (rc == NFT_SUCCESS) {
            be.h2d(dbs, bs, K * sizeof(cplx));
            be.h2d(dnc, nc, K * sizeof(cplx));
            P.D = (long long)D; P.K = (int)K; P.bs = dbs; P.nc = dnc; P.T0 = T[0]; P.eps_t = eps_t;
            P.q = dq; P.work = work;
            size_t zc = 0;                                    // :726-733: first sample with t >= 0 (0 if none)
            for (size_t i = 0; i < D; i++)
                if (T[0] + eps_t * (double)i >= 0.0) { zc = i; break; }
            P.zc = (long long)zc;
        }
        if (rc == NFT_SUCCESS && mode == 0) {
            be.template run<KInvSolitons>((int)((D + 255) / 256), 1, P);
        } else if (rc == NFT_SUCCESS) {
            // eigenfunctions of the seed, :908-1007: two half steps per sample interval = the chunk-parallel
            // scatterer of the discrete spectrum on the signal q' = (q0, q1, q1, q2, q2, ...) with step eps_t/2
            const size_t D2 = 2 * (D - 1);
            BsParams B;
            std::memset(&B, 0, sizeof(B));
            size_t L = (D2 + 16383) / 16384;
            if (L < 16) L = 16;
            if (L % 2) L++;
            const size_t nchunk = (D2 + L - 1) / L;
            ok = pl.alloc(dq2, D2) && pl.alloc(cm, K * nchunk * 8) && pl.alloc(bnd, K * (nchunk + 1) * 2)
                 && pl.alloc(bndp, K * (nchunk + 1) * 2) && pl.alloc(PHI, K * D * 2) && pl.alloc(PSI, K * D * 2)
                 && pl.alloc(dout, 3 * K);
            rc = ok ? NFT_SUCCESS : NFT_EC_NOMEM;
            if (rc == NFT_SUCCESS) {
                be.h2d(dq, q, D * sizeof(cplx));
                InvOpParams O;
                std::memset(&O, 0, sizeof(O));
                O.op = INV_DOUBLE_Q; O.n = (long long)D2; O.a = dq; O.out = dq2;
                be.template run<KInvOp>((int)((D2 + 255) / 256), 1, O);
                B.q = dq2; B.D = (long long)D2; B.ups = 2; B.lscale = 1.0;
                B.eps = 0.5 * eps_t;
                B.T0 = T[0] + 0.5 * B.eps;                 // the combine kernels start at T0 - eps/2 and end at T1 + eps/2
                B.T1 = T[1] - 0.5 * B.eps;
                B.K = (int)K; B.lam = dbs; B.L = (int)L; B.nchunk = (int)nchunk;
                B.cm = cm; B.bnd = bnd; B.bndp = bndp; B.PHI = PHI; B.PSI = PSI;
                B.a = dout; B.aprime = dout + K; B.b = dout + 2 * K;
                const int gx = (int)((nchunk + 63) / 64);
                be.template run<KBsChunk<false>>(gx, (int)K, B);
                be.template run<KBsCombine<false>>((int)K, 1, B);
                be.template run<KBsPhi>(gx, (int)K, B);
                be.template run<KBsChunk<true>>(gx, (int)K, B);
                be.template run<KBsCombine<true>>((int)K, 1, B);
                be.template run<KBsPsi>(gx, (int)K, B);
                P.PHI = PHI; P.PSI = PSI;
                be.template run<KInvCdt>((int)((D + 255) / 256), 1, P);
            }
        }
        if (rc == NFT_SUCCESS) {
            be.d2h(q, dq, D * sizeof(cplx));
            rc = be.sync();
        }
        be.free(dbs); be.free(dnc); be.free(dq); be.free(work); be.free(dq2); be.free(cm); be.free(bnd); be.free(bndp);
        be.free(PHI); be.free(PSI); be.free(dout);
        return rc;
    }
};
