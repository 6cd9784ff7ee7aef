// nft_discspec.h -- back-end independent host logic of the discrete spectrum of fnft_nsev
// (bound states, norming constants, residues; kappa = +1).
//
// Mirrors the control flow of the reference (file:line relative to the FNFT source tree):
//   fnft_nsev (outer)                    src/fnft_nsev.c:276-309 (SUBSAMPLE_AND_REFINE), :316-442 (Richardson)
//   nsev_compute_boundstates             src/fnft_nsev.c:596-741
//   nsev_refine_bound_states_newton      src/fnft_nsev.c:971-1038
//   nsev_compute_normconsts_or_residues  src/fnft_nsev.c:895-968
//   misc_filter / misc_merge / l2norm2   src/private/fnft__misc.c:114-157, :228-259, :90-112
// The heavy parts run on the device: the product tree (NftPlan), all roots of the a-polynomial
// (Ehrlich-Aberth kernels instead of the reference's Fortran QR) and the chunk-parallel slow
// scatterer (body_bs_*).
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <memory>
#include <vector>

#include <chrono>
#include <cstdlib>

#include "nft_plan.h"

// stage times of the last discrete-spectrum call of this thread (bench.py --workload cfg4 reports them through
// fnft_amd_discspec_stage_ms); the product library reads no environment variable
struct NftDsStage { const char *what; double ms; };
inline std::vector<NftDsStage> &nft_ds_stages()
{
    static thread_local std::vector<NftDsStage> v;
    return v;
}
struct NftDsClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    NftDsClock() { nft_ds_stages().clear(); }
    void lap(const char *what)
    {
        const auto t1 = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        nft_ds_stages().push_back(NftDsStage{what, ms});
        t0 = t1;
    }
};

struct NftDsOpts {
    int bsfilt;        // fnft_nsev_bsfilt_t: 0 NONE, 1 BASIC, 2 FULL
    int bsloc;         // fnft_nsev_bsloc_t: 0 FAST_EIGENVALUE, 1 NEWTON, 2 SUBSAMPLE_AND_REFINE
    size_t niter;
    size_t Dsub;
    int dstype;        // fnft_nsev_dstype_t: 0 NORMING_CONSTANTS, 1 RESIDUES, 2 BOTH
    int nse_disc;
    int richardson;
};

template <class BE> class NftDiscSpec {
public:
    typedef std::complex<double> cd;
    BE &be;
    int warn_more_than_K = 0;   // set when more bound states were found than the caller has room for
    int warn_roots_unconverged = 0;   // the root finder stopped at its sweep limit with a small but non-zero correction
    double last_root_corr = 0.0;      // largest relative correction of the root finder's last sweep
    static constexpr int kAberthMaxSweeps = 80;
    static constexpr double kAberthTol = 4.0e-14;   // stop: largest |correction| / |root| of a sweep
    static constexpr double kAberthFastAbove = 1.0e-3;   // last sweep's correction above this: single-precision Aberth sum
    static constexpr double kAberthFail = 1.0e-6;   // above this after the last sweep: not converged, error
    explicit NftDiscSpec(BE &be_) : be(be_) {}

    // one preprocessed signal on the device
    struct Prepared {
        std::unique_ptr<NftPlan<BE>> pl;
        cplx *d_in = nullptr;         // uploaded samples (owned)
        const cplx *d_qpre = nullptr; // preprocessed samples (d_in or the plan's buffer)
        std::vector<cd> h_qpre;       // host copy (bounding box)
        size_t Deff = 0, Dsub = 0;
        int ups = 1, deg0 = 0;
        double T[2] = {0, 0}, eps_t = 0;
        bool coeffs_done = false;
    };
    void release(Prepared &P)
    {
        if (P.pl) P.pl->destroy();
        P.pl.reset();
        be.free(P.d_in);
        P.d_in = nullptr;
    }

    // fnft__nse_discretization_preprocess_signal(D, q, eps_t, kappa = +1, &Dsub, ...), :386-656
    // need_tree = false: the signal is only handed to the slow scatterer (Newton refinement, norming constants);
    // without resampling (2SPLIT*) that needs the samples on the device and no plan at all.
    int prepare(size_t D, const cd *q, const double T[2], size_t Dsub_wish, int nse_disc, Prepared &P,
                bool need_tree = true)
    {
        const int akns = nft_nse_to_akns(nse_disc);
        if (akns < 0) return NFT_EC_INVALID_ARGUMENT;
        P.ups = nft_nse_upsampling(nse_disc);
        P.deg0 = nft_akns_degree(akns);
        size_t Dsub = Dsub_wish;
        if (Dsub < 2) Dsub = 2;
        if (Dsub > D) Dsub = D;
        const size_t nskip = (size_t)std::llround((double)D / (double)Dsub);
        Dsub = (size_t)std::llround((double)D / (double)nskip);
        const double eps_in = (T[1] - T[0]) / (double)(D - 1);
        P.Dsub = Dsub;
        P.Deff = Dsub * (size_t)P.ups;
        P.T[0] = T[0];
        P.T[1] = T[0] + (double)((Dsub - 1) * nskip) * eps_in;
        P.eps_t = (P.T[1] - P.T[0]) / (double)(Dsub - 1);
        std::vector<cd> hq;
        const cd *src = q;
        size_t Din = D;
        if (P.ups == 1 && nskip > 1) {
            hq.resize(Dsub);
            for (size_t i = 0; i < Dsub; i++) hq[i] = q[i * nskip];
            src = hq.data();
            Din = Dsub;
        }
        int rc = NFT_SUCCESS;
        if (!need_tree && P.ups == 1) {
            P.d_in = (cplx *)be.alloc(Din * sizeof(cplx));
            if (!P.d_in) return NFT_EC_NOMEM;
            be.h2d(P.d_in, src, Din * sizeof(cplx));
            P.d_qpre = P.d_in;
            P.h_qpre.assign(src, src + Din);
            return be.sync();
        }
        P.pl.reset(new NftPlan<BE>(be, P.Deff, 0, 1, akns, P.deg0));
        P.pl->set_front(Din, P.ups == 1 ? 1 : nskip, P.ups);
        rc = P.pl->init();
        if (rc != NFT_SUCCESS) return rc;
        P.d_in = (cplx *)be.alloc(Din * sizeof(cplx));
        if (!P.d_in) return NFT_EC_NOMEM;
        be.h2d(P.d_in, src, Din * sizeof(cplx));
        double Tfront[2] = {T[0], P.ups == 1 ? P.T[1] : T[1]}, Tsub[2];
        rc = P.pl->run_front(P.d_in, Tfront, +1, Tsub);   // resampling (4SPLIT) + level 0 of the tree
        if (rc == NFT_SUCCESS) rc = P.pl->read_status();   // MODAL step-size check, fnft__akns_fscatter.c:122-126
        if (rc != NFT_SUCCESS) return rc;
        P.coeffs_done = true;
        P.d_qpre = (P.ups == 1) ? P.d_in : P.pl->qpre;
        P.h_qpre.resize(P.Deff);
        if (P.ups == 1) std::copy(src, src + Din, P.h_qpre.begin());
        else {
            be.d2h(P.h_qpre.data(), P.pl->qpre, P.Deff * sizeof(cplx));
            rc = be.sync();
        }
        return rc;
    }

    // ---- slow scatterer, fnft__nse_scatter_bound_states.c (BO / CF4_2) -----------------------
    int scatter(const Prepared &P, size_t K, const cd *lam, cd *a, cd *ap, cd *b, bool skip_b)
    {
        if (K == 0) return NFT_SUCCESS;
        if (P.ups == 2 && P.Deff % 2 != 0) return NFT_EC_OTHER;
        BsParams B;
        std::memset(&B, 0, sizeof(B));
        B.q = P.d_qpre;
        B.D = (long long)P.Deff;
        B.ups = P.ups;
        B.lscale = (P.ups == 2) ? 0.5 : 1.0;
        B.T0 = P.T[0]; B.T1 = P.T[1]; B.eps = P.eps_t;
        B.K = (int)K;
        // chunks of >= 16 samples, about 16384 of them on a long signal (one lane each in the chunk kernels;
        // the combine kernels take them in runs of nchunk/256 per lane)
        size_t L = (P.Deff + 16383) / 16384;
        if (L < 16) L = 16;
        if (L % 2) L++;
        B.L = (int)L;
        B.nchunk = (int)((P.Deff + L - 1) / L);
        const size_t Dg = P.Deff / (size_t)P.ups;
        cplx *d_lam = (cplx *)be.alloc(K * sizeof(cplx));
        cplx *d_out = (cplx *)be.alloc(3 * K * sizeof(cplx));
        cplx *cm = (cplx *)be.alloc(K * (size_t)B.nchunk * 8 * sizeof(cplx));
        cplx *bnd = (cplx *)be.alloc(K * (size_t)(B.nchunk + 1) * 2 * sizeof(cplx));
        cplx *bndp = nullptr, *PHI = nullptr, *best = nullptr;
        bool ok = d_lam && d_out && cm && bnd;
        if (!skip_b) {
            bndp = (cplx *)be.alloc(K * (size_t)(B.nchunk + 1) * 2 * sizeof(cplx));
            PHI = (cplx *)be.alloc(K * (Dg + 1) * 2 * sizeof(cplx));
            best = (cplx *)be.alloc(K * (size_t)B.nchunk * 2 * sizeof(cplx));
            ok = ok && bndp && PHI && best;
        }
        int rc = ok ? NFT_SUCCESS : NFT_EC_NOMEM;
        if (ok) {
            be.h2d(d_lam, lam, K * sizeof(cplx));
            B.lam = d_lam; B.cm = cm; B.bnd = bnd; B.bndp = bndp; B.PHI = PHI; B.best = best;
            B.a = d_out; B.aprime = d_out + K; B.b = d_out + 2 * K;
            const int gx = (B.nchunk + 63) / 64;
            be.template run<KBsChunk<false>>(gx, (int)K, B);
            be.template run<KBsCombine<false>>((int)K, 1, B);
            if (!skip_b) {
                be.template run<KBsPhi>(gx, (int)K, B);
                be.template run<KBsChunk<true>>(gx, (int)K, B);
                be.template run<KBsCombine<true>>((int)K, 1, B);
                be.template run<KBsMetric>(gx, (int)K, B);
                be.template run<KBsPick>((int)K, 1, B);
            }
            std::vector<cd> h(3 * K);
            be.d2h(h.data(), d_out, (skip_b ? 2 : 3) * K * sizeof(cplx));
            rc = be.sync();
            for (size_t i = 0; i < K; i++) {
                a[i] = h[i];
                ap[i] = h[K + i];
                if (!skip_b) b[i] = h[2 * K + i];
            }
        }
        be.free(d_lam); be.free(d_out); be.free(cm); be.free(bnd);
        be.free(bndp); be.free(PHI); be.free(best);
        return rc;
    }

    // ---- all roots of a polynomial on the device (coefficients highest power first) ----------
    int roots(const cplx *d_coef, size_t n, std::vector<cd> &z)
    {
        z.assign(n, cd(0, 0));
        if (n == 0) return NFT_SUCCESS;
        std::vector<cd> c(n + 1);
        be.d2h(c.data(), d_coef, (n + 1) * sizeof(cplx));
        int rc = be.sync();
        if (rc != NFT_SUCCESS) return rc;
        {   // coefficients that are exactly zero at either end (the 12 entry of a transfer matrix with symmetric samples
            // can have them): the degree drops -- roots at infinity -- and z = 0 is a root; the iteration runs on the rest
            size_t nl = 0, nt = 0;
            while (nl < n && c[nl] == cd(0, 0)) nl++;
            while (nt < n - nl && c[n - nt] == cd(0, 0)) nt++;
            if (nl + nt > 0) {
                const size_t m = n - nl - nt;
                std::vector<cd> zr;
                rc = (m > 0) ? roots(d_coef + nl, m, zr) : NFT_SUCCESS;
                if (rc != NFT_SUCCESS) return rc;
                for (size_t i = 0; i < m; i++) z[i] = zr[i];
                for (size_t i = 0; i < nl; i++) z[m + i] = cd(INFINITY, 0.0);
                for (size_t i = 0; i < nt; i++) z[m + nl + i] = cd(0.0, 0.0);
                return NFT_SUCCESS;
            }
        }
        // start values: moduli from the upper convex hull of log|c_k| (Bini), angles equispaced
        std::vector<double> la(n + 1);
        for (size_t k = 0; k <= n; k++) {
            const double m = std::abs(c[n - k]);   // ascending powers
            la[k] = m > 0.0 ? std::log(m) : -1.0e300;
        }
        std::vector<size_t> hull;
        for (size_t k = 0; k <= n; k++) {
            while (hull.size() >= 2) {
                const size_t k1 = hull[hull.size() - 2], k2 = hull.back();
                if ((la[k2] - la[k1]) * (double)(k - k1) <= (la[k] - la[k1]) * (double)(k2 - k1)) hull.pop_back();
                else break;
            }
            hull.push_back(k);
        }
        const double tau = 6.283185307179586476925286766559;
        size_t pos = 0;
        for (size_t h = 0; h + 1 < hull.size(); h++) {
            const size_t k1 = hull[h], k2 = hull[h + 1], m = k2 - k1;
            double lr = (la[k1] - la[k2]) / (double)m;
            if (lr > 300.0) lr = 300.0;
            if (lr < -300.0) lr = -300.0;
            const double r = std::exp(lr);
            for (size_t j = 0; j < m; j++) {
                const double ang = tau * (double)j / (double)m + tau * (double)h / (double)n + 0.7;
                z[pos++] = cd(r * std::cos(ang), r * std::sin(ang));
            }
        }
        AberthParams A;
        A.coef = d_coef;
        A.n = (long long)n;
        // segments of the two O(n^2) kernels: about BE::kTargetWorkgroups (2048) workgroups of 256 lanes per launch.  The segment arrays
        // hold kSegCap values per estimate; with fewer estimates left, a sweep uses more segments.
        constexpr size_t kSegCap = 16;
        cplx *zbuf[2] = {(cplx *)be.alloc(n * sizeof(cplx)), (cplx *)be.alloc(n * sizeof(cplx))};
        int *ibuf[2] = {(int *)be.alloc(n * sizeof(int)), (int *)be.alloc(n * sizeof(int))};
        A.pp = (cplx *)be.alloc(kSegCap * n * sizeof(cplx));
        A.pd = (cplx *)be.alloc(kSegCap * n * sizeof(cplx));
        A.ps = (cplx *)be.alloc(kSegCap * n * sizeof(cplx));
        A.pe = (double *)be.alloc(kSegCap * n * sizeof(double));
        A.hit = (int *)be.alloc(n * sizeof(int));
        // {bits of the largest correction, number of estimates that moved}
        unsigned long long *d_state = (unsigned long long *)be.alloc(2 * sizeof(unsigned long long));
        A.maxcorr = d_state;
        A.cnt = (int *)(d_state + 1);
        if (!zbuf[0] || !zbuf[1] || !ibuf[0] || !ibuf[1] || !A.pp || !A.pd || !A.ps || !A.pe || !A.hit || !d_state)
            rc = NFT_EC_NOMEM;
        int cur = 0, icur = 0;
        double mc = 1.0;
        size_t na = n;
        if (rc == NFT_SUCCESS) {
            std::vector<int> ident(n);
            for (size_t i = 0; i < n; i++) ident[i] = (int)i;
            be.h2d(zbuf[0], z.data(), n * sizeof(cplx));
            be.h2d(zbuf[1], z.data(), n * sizeof(cplx));
            be.h2d(ibuf[0], ident.data(), n * sizeof(int));
            be.memset0(A.hit, n * sizeof(int));
            for (int it = 0; it < kAberthMaxSweeps && na > 0; it++) {
                // estimates are double-buffered: a sweep reads zbuf[cur] everywhere and writes zbuf[cur ^ 1]
                A.z = zbuf[cur];
                A.z_out = zbuf[cur ^ 1];
                A.idx = ibuf[icur];
                A.idx_out = ibuf[icur ^ 1];
                A.na = (long long)na;
                const int gx = (int)((na + 255) / 256);
                size_t S = (BE::kTargetWorkgroups + (size_t)gx - 1) / (size_t)gx;
                if (S > 64) S = 64;
                if (S * na > kSegCap * n) S = kSegCap * n / na;
                if (n < 128 || S < 1) S = 1;
                A.S = (int)S;
                A.L = (long long)(((n + 1 + S - 1) / S + 7) / 8 * 8);
                A.J = (long long)((n + S - 1) / S);
                A.fast = (mc > kAberthFastAbove) ? 1 : 0;   // single-precision repulsion sum while far from convergence
                be.memset0(d_state, 2 * sizeof(unsigned long long));
                be.template run<KAberthNewton>(gx, (int)S, A);
                be.template run<KAberthSum>(gx, (int)S, A);
                be.template run<KAberthApply>(gx, 1, A);
                cur ^= 1;
                icur ^= 1;
                unsigned long long st[2] = {0, 0};
                be.d2h(st, d_state, sizeof(st));
                rc = be.sync();
                if (rc != NFT_SUCCESS) break;
                std::memcpy(&mc, &st[0], sizeof(mc));
                na = (size_t)(st[1] & 0xffffffffull);
#ifdef FNFT_AMD_TUNING
                std::fprintf(stderr, "[aberth] n=%zu sweep %d max rel corr %.3e, %zu estimates moved, S=%zu\n", n, it, mc, na, S);
#endif
                if (mc < kAberthTol) break;
            }
            // polish (AberthParams::polish): two plain Newton steps on every estimate, half the work of a sweep each.
            // eiscor's QR (fnft__poly_roots_fasteigen.c:29-48) is backward stable; this brings clustered roots from the
            // sweeps' worst-case stopping level to what the conditioning of the polynomial allows
            // The first polish sweep takes every estimate; the second only those the first one still moved by more than
            // rounding (the clustered ones: a few hundred of 41 120 at cfg 4 -- a full second sweep was 1 ms of 20).
            if (rc == NFT_SUCCESS && mc < kAberthFail) {
                be.h2d(ibuf[0], ident.data(), n * sizeof(int));
                size_t np = n;
                for (int it = 0; it < 2 && np > 0 && rc == NFT_SUCCESS; it++) {
                    A.z = zbuf[cur];
                    A.z_out = (it == 0) ? zbuf[cur ^ 1] : zbuf[cur];   // second sweep: its subset in place (no sum over others)
                    A.idx = ibuf[it];
                    A.idx_out = ibuf[it ^ 1];
                    A.na = (long long)np;
                    A.polish = 1;
                    // both sweeps cut the polynomial into the same segments (those of a sweep over all n estimates): an
                    // estimate's second step is then the same arithmetic whether or not its neighbours take part
                    const int gx = (int)((np + 255) / 256);
                    const int gxn = (int)((n + 255) / 256);
                    size_t S = (BE::kTargetWorkgroups + (size_t)gxn - 1) / (size_t)gxn;
                    if (S > kSegCap) S = kSegCap;
                    if (n < 128 || S < 1) S = 1;
                    A.S = (int)S;
                    A.L = (long long)(((n + 1 + S - 1) / S + 7) / 8 * 8);
                    A.J = (long long)((n + S - 1) / S);
                    be.memset0(d_state, 2 * sizeof(unsigned long long));
                    be.template run<KAberthNewton>(gx, (int)S, A);
                    be.template run<KAberthApply>(gx, 1, A);
                    if (it == 0) {
                        cur ^= 1;
                        unsigned long long st[2] = {0, 0};
                        be.d2h(st, d_state, sizeof(st));
                        rc = be.sync();
                        np = (size_t)(st[1] & 0xffffffffull);
                    }
                }
            }
            if (rc == NFT_SUCCESS) {
                be.d2h(z.data(), zbuf[cur], n * sizeof(cplx));
                rc = be.sync();
            }
        }
        be.free(zbuf[0]); be.free(zbuf[1]); be.free(ibuf[0]); be.free(ibuf[1]);
        be.free(A.pp); be.free(A.pd); be.free(A.ps); be.free(A.pe); be.free(A.hit); be.free(d_state);
        last_root_corr = mc;
        if (rc == NFT_SUCCESS && !(mc < kAberthTol)) {
            // the sweep limit was reached: the reference's QR (eiscor) reports non-convergence as an error of
            // poly_roots_fasteigen (src/private/fnft__poly_roots_fasteigen.c:41-45); a correction that is merely
            // above the stopping tolerance but small is returned with a warning
            if (!(mc < kAberthFail)) return NFT_EC_OTHER;
            warn_roots_unconverged = 1;
        }
        return rc;
    }

    // ---- fnft__misc.c helpers ------------------------------------------------------------------
    static double l2norm2(size_t N, const cd *Z, double a, double b)
    {
        if (N < 2 || a >= b) return NAN;
        const double h = (b - a) / (double)N;
        double val = 0.5 * h * std::norm(Z[0]);
        for (size_t i = 1; i + 1 < N; i++) val += h * std::norm(Z[i]);
        val += 0.5 * h * std::norm(Z[N - 1]);
        return val;
    }
    static void filter_merge(std::vector<cd> &v, const double box[4])
    {
        size_t nf = 0;
        for (size_t i = 0; i < v.size(); i++) {
            if (!(v[i].real() >= box[0]) || !(v[i].real() <= box[1])) continue;
            if (!(v[i].imag() >= box[2]) || !(v[i].imag() <= box[3])) continue;
            v[nf++] = v[i];
        }
        v.resize(nf);
        if (nf == 0) return;
        const double tol = std::sqrt(2.220446049250313e-16);
        size_t kept = 1;
        for (size_t i = 1; i < nf; i++) {
            double dist = -1.0;
            for (size_t j = 0; j < i; j++) {
                dist = std::abs(v[j] - v[i]);
                if (dist < tol) break;
            }
            if (dist < tol) continue;
            v[kept++] = v[i];
        }
        v.resize(kept);
    }

    // ---- fnft_nsev_base, discrete part (:545-560) -------------------------------------------
    // in: bs holds the initial guesses for NEWTON.  out: bs, and nc = [normconsts | residues] per dstype
    int base(Prepared &P, const NftDsOpts &o, int bsloc, std::vector<cd> &bs, std::vector<cd> *nc,
             std::vector<cd> *aprimes)
    {
        const double inf = INFINITY;
        double box[4] = {-inf, inf, -inf, inf};
        if (o.bsfilt == 1) { box[2] = 0.0; }
        else if (o.bsfilt == 2) {
            box[1] = 0.9 * 3.14159265358979323846 / std::fabs(2.0 / (double)P.deg0 * P.eps_t);
            box[0] = -box[1];
            box[2] = 0.0;
            if (P.ups == 1) box[3] = 1.5 * 0.25 * l2norm2(P.Dsub, P.h_qpre.data(), P.T[0], P.T[1]);
            else {
                std::vector<cd> qt(P.Dsub);
                for (size_t i = 0, j = 1; i < P.Dsub; i++, j += (size_t)P.ups) qt[i] = (double)P.ups * P.h_qpre[j];
                box[3] = 1.5 * 0.25 * l2norm2(P.Dsub, qt.data(), P.T[0], P.T[1]);
            }
        }
        int rc = NFT_SUCCESS;
        if (bsloc == 1) {   // NEWTON, :971-1038 -- all eigenvalues advance together, each with its own stop
            const size_t K = bs.size();
            if (K > 0 && o.niter > 0) {
                if (!(box[0] <= box[1]) || !(box[2] <= box[3])) return NFT_EC_INVALID_ARGUMENT;
                std::vector<char> active(K, 1);
                std::vector<size_t> iters(K, 0);
                std::vector<cd> lam, a, ap, bdummy;
                std::vector<size_t> idx;
                const double eprec = 100.0 * 2.220446049250313e-16;
                for (;;) {
                    lam.clear(); idx.clear();
                    for (size_t i = 0; i < K; i++) if (active[i]) { lam.push_back(bs[i]); idx.push_back(i); }
                    if (lam.empty()) break;
                    a.resize(lam.size()); ap.resize(lam.size()); bdummy.resize(lam.size());
                    rc = scatter(P, lam.size(), lam.data(), a.data(), ap.data(), bdummy.data(), true);
                    if (rc != NFT_SUCCESS) return -std::abs(rc);
                    for (size_t t = 0; t < idx.size(); t++) {
                        const size_t i = idx[t];
                        if (a[t] == cd(0, 0)) { active[i] = 0; continue; }
                        if (ap[t] == cd(0, 0)) return NFT_EC_DIV_BY_ZERO;
                        const cd err = a[t] / ap[t];
                        bs[i] -= err;
                        iters[i]++;
                        if (bs[i].imag() > box[3] || bs[i].real() > box[1] || bs[i].real() < box[0]
                            || bs[i].imag() < box[2]) { active[i] = 0; continue; }
                        if (!(std::abs(err) > eprec && iters[i] < o.niter)) active[i] = 0;
                    }
                }
            }
        } else if (bsloc == 0) {   // FAST_EIGENVALUE, :683-706
            rc = P.pl->run_tree();
            if (rc != NFT_SUCCESS) return -std::abs(rc);
            P.pl->export_tm();
            const size_t deg = P.pl->res_deg;
            std::vector<cd> z;
            rc = roots(P.pl->tm_out, deg, z);
            if (rc != NFT_SUCCESS) return -std::abs(rc);
            const cd den = cd(0.0, 2.0 * P.eps_t / (double)(P.deg0 * P.ups));   // z -> lambda, :225-240
            bs.resize(deg);
            for (size_t i = 0; i < deg; i++) bs[i] = std::log(z[i]) / den;
        } else {
            return NFT_EC_INVALID_ARGUMENT;
        }
        if (o.bsfilt != 0) filter_merge(bs, box);
        if (nc != nullptr && !bs.empty()) {   // :895-968
            const size_t K = bs.size();
            std::vector<cd> a(K), ap(K), b(K);
            rc = scatter(P, K, bs.data(), a.data(), ap.data(), b.data(), false);
            if (rc != NFT_SUCCESS) return -std::abs(rc);
            nc->assign(b.begin(), b.end());
            if (o.dstype != 0) {
                if (o.dstype == 2) nc->insert(nc->end(), b.begin(), b.end());
                const size_t off = (o.dstype == 2) ? K : 0;
                for (size_t i = 0; i < K; i++) {
                    if (ap[i] == cd(0, 0)) return NFT_EC_DIV_BY_ZERO;
                    (*nc)[off + i] /= ap[i];
                }
            }
            if (aprimes) *aprimes = ap;
        } else if (nc != nullptr) {
            nc->clear();
        }
        return NFT_SUCCESS;
    }

    // ---- fnft_nsev, discrete part.  *K_ptr: capacity in, number of bound states out -----------
    int run(size_t D, const cd *q, const double T[2], const NftDsOpts &o, size_t *K_ptr, cd *bound_states,
            cd *normconsts_or_residues)
    {
        Prepared full, sub;
        std::vector<cd> bs, nc, ap;
        NftDsOpts ob = o;
        NftDsClock clk;
        if (o.richardson && o.dstype == 1) ob.dstype = 2;   // residues need norming constants too, :248-258
        int rc = prepare(D, q, T, D, o.nse_disc, full, o.bsloc == 0);
        clk.lap("prepare(full)");
        if (rc == NFT_SUCCESS) {
            if (o.bsloc == 2) {   // SUBSAMPLE_AND_REFINE, :276-304
                size_t Dsub = o.Dsub;
                if (Dsub == 0) Dsub = (size_t)std::sqrt((double)D * std::log2((double)D) * std::log2((double)D));
                rc = prepare(D, q, T, Dsub, o.nse_disc, sub);
                clk.lap("prepare(sub)");
                if (rc == NFT_SUCCESS) rc = base(sub, ob, 0, bs, nullptr, nullptr);
                clk.lap("roots + filter (sub)");
                release(sub);
                if (rc == NFT_SUCCESS && bs.size() > *K_ptr) { warn_more_than_K = 1; bs.resize(*K_ptr); }
                if (rc == NFT_SUCCESS)
                    rc = base(full, ob, 1, bs, normconsts_or_residues ? &nc : nullptr, &ap);
                clk.lap("newton + normconsts (full)");
            } else {
                if (o.bsloc == 1) bs.assign(bound_states, bound_states + *K_ptr);
                rc = base(full, ob, o.bsloc, bs, normconsts_or_residues ? &nc : nullptr, &ap);
            }
        }
        if (rc == NFT_SUCCESS && bs.size() > *K_ptr) {
            warn_more_than_K = 1;
            const size_t K0 = bs.size(), K1 = *K_ptr;
            std::vector<cd> nc2;
            for (size_t part = 0; part * K0 < nc.size(); part++)
                nc2.insert(nc2.end(), nc.begin() + part * K0, nc.begin() + part * K0 + K1);
            nc.swap(nc2);
            bs.resize(K1);
            ap.resize(std::min(ap.size(), K1));
        }
        // Richardson extrapolation of the discrete spectrum, :340-364, :376-392, :406-441
        if (rc == NFT_SUCCESS && o.richardson && !bs.empty()) {
            Prepared half;
            std::vector<cd> bs_s(bs), nc_s, ap_s;
            rc = prepare(D, q, T, D / 2, o.nse_disc, half, false);
            if (rc == NFT_SUCCESS) rc = base(half, ob, 1, bs_s, normconsts_or_residues ? &nc_s : nullptr, &ap_s);
            if (rc == NFT_SUCCESS && !bs_s.empty()) {
                const double eps_t = (T[1] - T[0]) / (double)(D - 1);
                const double sn = std::pow(half.eps_t / eps_t, (double)nft_nse_method_order(o.nse_disc));
                const double sd = sn - 1.0;
                const size_t K = bs.size(), Ks = bs_s.size();
                for (size_t i = 0; i < K; i++) {
                    size_t loc = Ks;
                    double thr = eps_t;
                    for (size_t j = 0; j < Ks; j++) {
                        const double e = std::abs(bs[i] - bs_s[j]) / std::abs(bs[i]);
                        if (e < thr) { thr = e; loc = j; }
                    }
                    if (loc < Ks) {
                        bs[i] = (sn * bs[i] - bs_s[loc]) / sd;
                        if (normconsts_or_residues && (o.dstype == 1 || o.dstype == 2)) {
                            // a' from norming constant / residue, Richardson step on a', residue again
                            cd api = nc[i] / nc[K + i], aps = nc_s[loc] / nc_s[Ks + loc];
                            api = (sn * api - aps) / sd;
                            nc[K + i] = nc[i] / api;
                        }
                    }
                }
            }
            release(half);
        }
        release(full);
        if (rc != NFT_SUCCESS) return rc;
        const size_t K = bs.size();
        std::copy(bs.begin(), bs.end(), bound_states);
        if (normconsts_or_residues) {
            if (o.richardson && o.dstype == 1) std::copy(nc.begin() + K, nc.begin() + 2 * K, normconsts_or_residues);
            else std::copy(nc.begin(), nc.end(), normconsts_or_residues);
        }
        *K_ptr = K;
        return NFT_SUCCESS;
    }
};
