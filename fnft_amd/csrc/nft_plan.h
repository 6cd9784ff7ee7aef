// nft_plan.h -- back-end independent host logic of the fnft_nsev continuous-spectrum path:
// workspace layout in HBM, level schedule of the product tree, chirp z-transform set-up.
//
// Mirrors the control flow of the reference (file:line relative to the FNFT source tree):
//   fnft_nsev_base           src/fnft_nsev.c:458-565
//   nse_fscatter             src/private/fnft__nse_fscatter.c:44-91
//   akns_fscatter            src/private/fnft__akns_fscatter.c:64-925
//   poly_fmult2x2            src/private/fnft__poly_fmult.c:381-546
//   nsev_compute_contspec    src/fnft_nsev.c:744-891
//   poly_chirpz              src/private/fnft__poly_chirpz.c:33-105
//
// BE (back end) supplies device memory, copies, kernel launches and stage timers; see
// hip_backend.hip (product) and tests/emu/emu_backend.h (CPU lane emulator, tests only).
#pragma once
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <vector>

#include "nft_dispatch.h"
#include "nft_schemes.h"

// return codes (include/fnft_errwarn.h:44-94)
enum {
    NFT_SUCCESS = 0, NFT_EC_NOMEM = 1, NFT_EC_INVALID_ARGUMENT = 2, NFT_EC_DIV_BY_ZERO = 3,
    NFT_EC_OTHER = 5, NFT_EC_NOT_YET_IMPLEMENTED = 6
};

inline size_t nft_nextpow2(size_t n)
{
    size_t r = 1;
    while (r < n) r *= 2;
    return r;
}
inline int nft_log2(size_t n)
{
    int l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}

// fnft__akns_discretization.c:29-67 restricted to what the coefficient kernel implements
inline int nft_akns_degree(int akns_disc)
{
    switch (akns_disc) {
    case 0: case 1: case 2: case 3: case 4: case 5: return 1;   // MODAL,1A,1B,2A,2B,2S
    case 8: case 10: return 2;                                   // 3S, 4B
    case 6: case 7: return 3;                                    // 3A, 3B
    case 9: return 4;                                            // 4A
    case 14: return 6;                                           // 6B
    case 13: case 18: return 12;                                 // 6A, 8B
    case 11: case 12: return 15;                                 // 5A, 5B
    case 17: return 24;                                          // 8A
    case 15: case 16: return 105;                                // 7A, 7B
    default: return 0;
    }
}
// fnft__nse_discretization.c:108-200 (upsampling-factor-1 splitting schemes); -1 otherwise
inline int nft_nse_to_akns(int nse_disc)
{
    switch (nse_disc) {
    case 0: return 0;    // 2SPLIT2_MODAL
    case 2: return 1;    // 2SPLIT1A
    case 3: return 2;    // 2SPLIT1B
    case 4: return 3;    // 2SPLIT2A
    case 5: return 4;    // 2SPLIT2B
    case 6: return 5;    // 2SPLIT2S
    case 7: return 6;    // 2SPLIT3A
    case 8: return 7;    // 2SPLIT3B
    case 9: return 8;    // 2SPLIT3S
    case 10: return 9;   // 2SPLIT4A
    case 11: return 10;  // 2SPLIT4B
    case 12: case 13: case 14: case 15: case 16: case 17: case 18: case 19:
        return nse_disc - 1;   // 2SPLIT5A .. 2SPLIT8B -> akns 11 .. 18
    case 20: return 9;   // 4SPLIT4A: per-sample formulas of 2SPLIT4A (fnft__akns_fscatter.c:362-363)
    case 21: return 10;  // 4SPLIT4B: ... of 2SPLIT4B (:402-403)
    default: return -1;
    }
}

// fnft__akns_discretization.c:114-152: preprocessed samples per step
inline int nft_nse_upsampling(int nse_disc) { return (nse_disc == 20 || nse_disc == 21) ? 2 : 1; }
// fnft__akns_discretization.c:157-192: order used by Richardson extrapolation
inline int nft_nse_method_order(int nse_disc) { return (nse_disc == 20 || nse_disc == 21) ? 4 : 2; }

// log of a complex number through libm's clog -- the routine glibc's cpow() (which the reference
// calls for every chirp factor, src/private/fnft__poly_chirpz.c:69,77,81,95) is built on.  It keeps
// the tiny real part log|z| of a z that is on the unit circle only up to rounding; the generic
// std::log(std::complex) of some C++ runtimes returns log(hypot) = 0 there, which changes
// |z|^(n^2/2) by up to 1e-9 at n = 2^16.
inline std::complex<double> nft_clog(std::complex<double> z)
{
    __complex__ double zz;
    __real__ zz = z.real();
    __imag__ zz = z.imag();
    const __complex__ double r = __builtin_clog(zz);
    return std::complex<double>(__real__ r, __imag__ r);
}

// transform length used for a product of two degree-d polynomials
inline size_t nft_product_len(size_t d)
{
    const size_t p = nft_nextpow2(2 * d);
    return (p == 2 * d) ? p : nft_nextpow2(2 * d + 1);
}

constexpr int kFineLog2 = 12;           // master twiddle: NMAX = 2^24
constexpr int kMaxTwTable = 16384;      // per-length tables up to this length (longest column transform; the real path's
                                        // quarter-step tables of 4*N1 entries)
constexpr size_t kMaxSplitTree = (size_t)kRowTree * 8192;    // largest column transform: 8192 (N up to 2^24)
constexpr size_t kMaxSplitChirp = (size_t)kRowChirp * 8192;  // chirp length up to 2^25 (any-length DFTs up to 2^24 points)
// fine tables of the two master twiddle pairs: 2^12 entries exp(-2 pi i j/2^24), then 2^13 entries exp(-2 pi i j/2^26)
constexpr size_t kTwLoEntries = ((size_t)1 << kFineLog2) + ((size_t)1 << (kFineLog2 + 1));

template <class BE> class NftPlan {
public:
    BE &be;
    size_t D, M, batch;
    int akns_disc, deg0;
    size_t Dpad, plane, n0;

    // device buffers
    cplx *body[2] = {nullptr, nullptr};
    cplx *tail[2] = {nullptr, nullptr};
    double *scale[2] = {nullptr, nullptr};
    unsigned *max2[2] = {nullptr, nullptr};  // ping-pong across split levels, kMax2Slots per matrix
    int *wexp[2] = {nullptr, nullptr};  // per matrix, ping-pong with body/tail/scale
    int *status = nullptr;
    int *wuser = nullptr;   // exponents of caller-supplied transfer matrices (run_contspec_tm)
    // monomial program of the order 5..8 schemes (nft_schemes.h), device copies
    double *prog_bfrac = nullptr, *prog_mw = nullptr;
    int *prog_ptr = nullptr;
    unsigned char *prog_fac = nullptr;
    int prog_nB = 0, prog_maxf = 0;
    // KdV (fnft_kdvv): r = -1 for every sample (fnft__kdv_fscatter.c:74-75), general 2x2 tree
    bool kdv = false;
    cplx *rneg = nullptr;
    // real-coefficient path (nft_real.h): r = -1 and a real potential make every coefficient real.  want_real is the
    // caller's request for the NEXT run_front / run_coeffs (hip_backend.hip decides per call); real_run says which layout
    // the current level arrays hold
    bool want_real = false;
    bool real_run = false;
    bool use_r3 = true;      // column lengths 3*2^j on the split levels of the real path
    bool use_rleaf = true;   // real even-order schemes: leaf kernel with direct products (body_rleaf_strang)
    // 4SPLIT4A/B front end (set_front): Din input samples per signal, every nskip-th step kept,
    // ups preprocessed samples per kept step; D = ups * Dsub matrices enter the tree
    size_t Din = 0, nskip = 1;
    int ups = 1;
    cplx *qpre = nullptr, *rsX = nullptr, *rsX12 = nullptr, *rsQ12 = nullptr, *rsY = nullptr, *rsV = nullptr;
    size_t Lr = 0;
    cplx *Y = nullptr, *Z = nullptr, *Z2 = nullptr;   // Z ping-pongs when spectral doubling is on
    cplx *chY = nullptr, *chV = nullptr, *chH = nullptr;
    cplx *chVS = nullptr;                 // cached spectrum of the chirp filter
    double vs_key[4] = {0, 0, 0, 0};      // log W (re, im), M, deg+1 it was computed for
    bool vs_valid = false;
    cplx *tm_out = nullptr;
    cplx *twtab = nullptr;   // concatenated tables for N = 2,4,...,kMaxTwTable
    cplx *twlo = nullptr;    // exp(-2 pi i j / 2^24), j < 4096
    // lengths 3*2^b (column transforms of the real path, nft_real.h): tables for L = 3, 6, ..., 3*2^kTw3MaxLog back to
    // back (offset L - 3), and the fine table exp(-2 pi i j/(3*2^22)), j < 4096, of the master pair for 3*2^22
    cplx *tw3tab = nullptr, *twlo3 = nullptr;
    size_t Lc = 0;           // chirp transform length
    size_t bytes = 0;
    int cur = 0;             // index of the body/tail/scale set holding the current level
    bool tree_valid = false;
    // The last split level leaves its maxima unreduced (64 slots per matrix): turning them into the pending scale and
    // the final exponent is a one-wave job (KFinalizeScales, 6 us as a launch of its own between the tree and the
    // evaluation).  The chirp transform's first kernel does it on the way (every workgroup reduces the slots of its
    // signal for itself, one of them stores scale and exponent: ChirpParams::fin_*); every other consumer of the root
    // (export, discrete spectrum, block matrices) calls ensure_final() first.
    bool final_pending = false;
    TreeLevel final_L;
    size_t final_n = 0;
    void defer_final(const TreeLevel &L, size_t n_out) { final_L = L; final_n = n_out; final_pending = true; }
    void ensure_final()
    {
        if (!final_pending) return;
        be.template run<KFinalizeScales>((int)final_n, 1, final_L);
        final_pending = false;
    }
    size_t res_deg = 0;      // degree of the transfer matrix of the last tree run
    size_t start_n = 0, start_d = 0;  // matrices (all signals) / degree the tree run starts from
    bool use_leaf = true;    // fuse coefficients + first levels (nft_kernels.h body_leaf)
    bool use_bridge = true;  // fuse inverse/forward column steps of consecutive split levels
    bool use_doubling = true; // ... with spectral doubling when N = 2d (body_col_bridge2)
    bool use_sym = true;     // NSE symmetry: store/transform only the first column (ne = 2)
    bool use_multi = true;   // several consecutive fused levels per launch (body_multi_fft)
    bool use_direct4 = true; // first split level: row kernel forms the length-4 column transform itself
    bool use_leaf_multi = true;   // ... with the leaf kernel in front of the first of them (body_leaf_multi)
    bool leaf_pending = false;    // run_coeffs left the leaf to the first launch of run_tree
    LeafParams leaf_lp;
    int ne = 4;              // stored entries per matrix in the current tree run
    int kappa_run = 1;
    unsigned long long *dbg_stamps = nullptr;   // -DFNFT_AMD_STAMPS builds: 16 u64 per wave of the row kernel
    int stamp_level = -1;                        // split level (0 = first) whose row kernel is stamped
    int last_warn = 0;       // bit 0: the resampler found the signal not band-limited (set by read_status)
    int tune_stagger = 0;    // row kernel start delay of the second half of a one-round grid (BigLevel::stagger)
    int dbg_flags = 0;       // timing ablation: only builds with -DFNFT_AMD_ABLATION ever set it (hip_backend.hip)

    NftPlan(BE &be_, size_t D_, size_t M_, size_t batch_, int akns_disc_, int deg0_)
        : be(be_), D(D_), M(M_), batch(batch_), akns_disc(akns_disc_), deg0(deg0_)
    {
        Dpad = nft_nextpow2(D);
        n0 = batch * Dpad;
        plane = n0 * (size_t)deg0;
    }

    // fnft__nse_discretization.c:419-427: samples kept when every nskip-th of Din is used
    static size_t sub_count(size_t Din_, size_t nskip_) { return (size_t)std::llround((double)Din_ / (double)nskip_); }
    // call before init(): this plan was constructed with D = ups_ * sub_count(Din_, nskip_)
    void set_front(size_t Din_, size_t nskip_, int ups_) { Din = Din_; nskip = nskip_; ups = ups_; }

    template <class T> bool alloc(T *&p, size_t count)
    {
        const size_t b = count * sizeof(T);
        p = (T *)be.alloc(b ? b : 16);
        if (!p) return false;
        bytes += b;
        return true;
    }

    const cplx *tw_table(size_t N) const
    {   // tables are stored back to back: N=2 at offset 0, N=4 at 2, N=8 at 6, ... offset = N-2
        return twtab + (N - 2);
    }
    static constexpr int kTw3MaxLog = 12;
    const cplx *tw3_table(size_t L) const { return tw3tab + (L - 3); }
    BigTwiddle big_tw3(size_t N) const   // N = 3*2^a <= 3*2^22
    {
        BigTwiddle t;
        t.hi = tw3_table((size_t)3 << 10);   // exp(-2 pi i jh/3072)
        t.lo = twlo3;
        t.fine_log2 = kFineLog2;
        t.shift = 22 - nft_log2(N / 3);
        return t;
    }
    BigTwiddle big_tw(size_t N) const
    {
        BigTwiddle t;
        if (N > ((size_t)1 << (2 * kFineLog2))) {   // 2^24 < N <= 2^26: the pair with 2^13-entry tables
            t.hi = tw_table((size_t)1 << (kFineLog2 + 1));
            t.lo = twlo + ((size_t)1 << kFineLog2);
            t.fine_log2 = kFineLog2 + 1;
            t.shift = 2 * (kFineLog2 + 1) - nft_log2(N);
            return t;
        }
        t.hi = tw_table((size_t)1 << kFineLog2);   // exp(-2 pi i jh / 2^FINE)
        t.lo = twlo;
        t.fine_log2 = kFineLog2;
        t.shift = 2 * kFineLog2 - nft_log2(N);
        return t;
    }

    // largest pair product done by one workgroup.  General (4-entry) form: 2048 -- a 4096-point pair needs 8 + 4
    // transforms of 16 points per lane in one workgroup (616 B of scratch per lane, 75 us for ONE pair); as a split
    // transform of 4 columns x 1024-point rows it is three small launches (~30 us for one pair, rows over 4 CUs)
    size_t fused_max_len() const { return (ne == 4 && FA_MID_GEN) ? (size_t)2048 : (size_t)kFusedMaxN; }

    int init()
    {
        if (D < 1 || batch < 1 || deg0 < 1) return NFT_EC_INVALID_ARGUMENT;
        if (Din == 0) Din = D;
        if (ups == 1 && (nskip != 1 || Din != D)) return NFT_EC_NOT_YET_IMPLEMENTED;
        if (ups == 2 && (Din <= 2 || D != 2 * sub_count(Din, nskip))) return NFT_EC_INVALID_ARGUMENT;
        // largest product transform of the tree and chirp length must be within the split limits
        const size_t topN = (Dpad > 1) ? nft_product_len(Dpad / 2 * (size_t)deg0) : 2;
        if (topN > kMaxSplitTree) return NFT_EC_NOT_YET_IMPLEMENTED;
        bool ok = true;
        for (int i = 0; i < 2; i++) {
            ok = ok && alloc(body[i], 4 * plane) && alloc(tail[i], 4 * n0) && alloc(scale[i], n0)
                 && alloc(wexp[i], n0);
        }
        {   // split levels hold at most n0*deg0/2048 matrices, kMax2Slots slots each
            const size_t nm = n0 * (size_t)deg0 / 32 + 4 * (size_t)kMax2Slots;
            ok = ok && alloc(max2[0], nm) && alloc(max2[1], nm) && alloc(status, 4);
            if (ok) be.memset0(status, 4 * sizeof(int));
        }
        {   // scratch of the split transforms: 4*n_in polynomials of N forward, 4*n_out inverse,
            // maximised over the levels that use them (N can exceed 2d when d is not 2^k)
            size_t needY = 0, needZ = 0, n = n0, d = (size_t)deg0;
            while (n / batch > 1) {
                const size_t N = nft_product_len(d);
                if (d > (size_t)kSchoolMaxDeg && N > fused_max_len()) {
                    if (4 * n * N > needY) needY = 4 * n * N;
                    if (2 * n * N > needZ) needZ = 2 * n * N;
                }
                n /= 2;
                d *= 2;
            }
            if (needY) ok = ok && alloc(Y, needY) && alloc(Z, needZ) && alloc(Z2, needZ);
        }
        if (M > 0) {
            const size_t Np = D * (size_t)deg0 + 1;
            Lc = nft_nextpow2(Np + M - 1);
            if (Lc < 2 * (size_t)kRowChirp) Lc = 2 * (size_t)kRowChirp;
            if (Lc > kMaxSplitChirp) return NFT_EC_NOT_YET_IMPLEMENTED;
            ok = ok && alloc(chY, batch * 2 * Lc) && alloc(chV, Lc) && alloc(chVS, Lc);
        }
        ok = ok && alloc(tm_out, batch * 4 * (D * (size_t)deg0 + 1));
        ok = ok && alloc(twtab, (size_t)2 * kMaxTwTable) && alloc(twlo, kTwLoEntries);
        if (kdv) ok = ok && alloc(tw3tab, (size_t)3 << (kTw3MaxLog + 1)) && alloc(twlo3, (size_t)1 << kFineLog2);
        if (kdv) {
            ok = ok && alloc(rneg, batch * D);
            if (ok) {
                std::vector<cplx> h(batch * D, cmake(-1.0, 0.0));
                be.h2d(rneg, h.data(), h.size() * sizeof(cplx));
            }
        }
        if (ups == 2) {
            Lr = nft_nextpow2(2 * Din - 1);
            if (Lr < 2 * (size_t)kRowChirp) Lr = 2 * (size_t)kRowChirp;
            if (Lr > kMaxSplitChirp) return NFT_EC_NOT_YET_IMPLEMENTED;
            ok = ok && alloc(qpre, batch * D) && alloc(rsX, batch * Din) && alloc(rsX12, batch * 2 * Din)
                 && alloc(rsQ12, batch * 2 * Din) && alloc(rsY, batch * 2 * Lr) && alloc(rsV, Lr);
        }
        CoeffProgramHost prog;
        const bool has_prog = akns_disc >= 11 && akns_disc <= 18;   // 2SPLIT5A .. 2SPLIT8B
        if (has_prog) {
            if (!nft_build_coeff_program(akns_disc, deg0, prog)) return NFT_EC_NOT_YET_IMPLEMENTED;
            ok = ok && alloc(prog_bfrac, prog.bfrac.size()) && alloc(prog_mw, prog.mw.size())
                 && alloc(prog_ptr, prog.tgt_ptr.size()) && alloc(prog_fac, prog.mfac.size());
        }
        if (!ok) return NFT_EC_NOMEM;
        if (has_prog) {
            be.h2d(prog_bfrac, prog.bfrac.data(), prog.bfrac.size() * sizeof(double));
            be.h2d(prog_mw, prog.mw.data(), prog.mw.size() * sizeof(double));
            be.h2d(prog_ptr, prog.tgt_ptr.data(), prog.tgt_ptr.size() * sizeof(int));
            be.h2d(prog_fac, prog.mfac.data(), prog.mfac.size());
            prog_nB = (int)prog.bfrac.size();
            prog_maxf = prog.maxf;
        }
        upload_twiddles();
        return NFT_SUCCESS;
    }

    void destroy()
    {
        for (int i = 0; i < 2; i++) { be.free(body[i]); be.free(tail[i]); be.free(scale[i]); be.free(wexp[i]); }
        be.free(max2[0]); be.free(max2[1]); be.free(status); be.free(Y); be.free(Z); be.free(Z2);
        be.free(chY); be.free(chV); be.free(chH); be.free(chVS); be.free(tm_out); be.free(twtab); be.free(twlo);
        be.free(prog_bfrac); be.free(prog_mw); be.free(prog_ptr); be.free(prog_fac);
        be.free(rneg); be.free(dbg_stamps); be.free(wuser); be.free(tw3tab); be.free(twlo3);
        be.free(qpre); be.free(rsX); be.free(rsX12); be.free(rsQ12); be.free(rsY); be.free(rsV);
    }

    // host copies of the tables, computed once per process (long-double sines: ~2 ms per plan otherwise)
    struct HostTables {
        std::vector<cplx> tw, lo, tw3, lo3;
        HostTables()
        {
            const long double tau = 6.283185307179586476925286766559005768L;
            tw.resize((size_t)2 * kMaxTwTable);
            for (size_t N = 2; N <= (size_t)kMaxTwTable; N *= 2)
                for (size_t j = 0; j < N; j++) {
                    const long double a = -tau * (long double)j / (long double)N;
                    tw[N - 2 + j] = cmake((double)cosl(a), (double)sinl(a));
                }
            lo.resize(kTwLoEntries);
            const long double nmax = (long double)((size_t)1 << (2 * kFineLog2));
            for (size_t j = 0; j < ((size_t)1 << kFineLog2); j++) {
                const long double a = -tau * (long double)j / nmax;
                lo[j] = cmake((double)cosl(a), (double)sinl(a));
            }
            for (size_t j = 0; j < ((size_t)1 << (kFineLog2 + 1)); j++) {
                const long double a = -tau * (long double)j / (4.0L * nmax);
                lo[((size_t)1 << kFineLog2) + j] = cmake((double)cosl(a), (double)sinl(a));
            }
            tw3.resize((size_t)3 << (kTw3MaxLog + 1));
            for (int b = 0; b <= kTw3MaxLog; b++) {
                const size_t Lb = (size_t)3 << b;
                for (size_t j = 0; j < Lb; j++) {
                    const long double a = -tau * (long double)j / (long double)Lb;
                    tw3[Lb - 3 + j] = cmake((double)cosl(a), (double)sinl(a));
                }
            }
            lo3.resize((size_t)1 << kFineLog2);
            const long double nmax3 = 3.0L * (long double)((size_t)1 << 22);
            for (size_t j = 0; j < lo3.size(); j++) {
                const long double a = -tau * (long double)j / nmax3;
                lo3[j] = cmake((double)cosl(a), (double)sinl(a));
            }
        }
    };
    static const HostTables &host_tables()
    {
        static const HostTables t;
        return t;
    }
    void upload_twiddles()
    {
        const HostTables &H = host_tables();
        be.h2d(twtab, H.tw.data(), H.tw.size() * sizeof(cplx));
        be.h2d(twlo, H.lo.data(), H.lo.size() * sizeof(cplx));
        if (tw3tab) {
            be.h2d(tw3tab, H.tw3.data(), H.tw3.size() * sizeof(cplx));
            be.h2d(twlo3, H.lo3.data(), H.lo3.size() * sizeof(cplx));
        }
    }

    // ---- front end: fnft__nse_discretization_preprocess_signal (:386-656) + level 0 ------------
    // T: interval of the Din input samples.  Tsub: interval of the kept steps (fnft_nsev.c:381-384),
    // to be handed to run_contspec.
    int run_front(const void *d_q, const double T[2], int kappa, double Tsub[2])
    {
        const double eps_in = (T[1] - T[0]) / (double)(Din - 1);
        if (ups == 1) {
            Tsub[0] = T[0];
            Tsub[1] = T[1];
            return run_coeffs(d_q, kdv ? rneg : nullptr, eps_in, kappa);
        }
        const size_t Dsub = D / 2;
        Tsub[0] = T[0];
        Tsub[1] = T[0] + (double)((Dsub - 1) * nskip) * eps_in;
        const double eps_t = (Tsub[1] - Tsub[0]) / (double)(Dsub - 1);
        // forward DFT of every signal (fnft__misc.c:366-370)
        ChirpParams C;
        std::memset(&C, 0, sizeof(C));
        C.poly = (const cplx *)d_q;
        C.deg = (long long)Din - 1;
        C.batch = (int)batch;
        C.npoly = 1;
        C.M = (long long)Din;
        C.Ybuf = rsY; C.Vbuf = rsV; C.Hbuf = rsX;
        fill_chirp_geometry(C, Lr);
        C.status = status;
        C.cstype = -1;
        C.dft_len = (long long)Din;
        C.dft_sign = -1;
        int rc = run_chirp(C);
        if (rc != NFT_SUCCESS) return rc;
        // phase ramps for the shifts -/+ sqrt(3)/6 of the kept step (:482-485, fnft__misc.c:383-393)
        ResampleParams R;
        R.X = rsX; R.X12 = rsX12; R.Q12 = rsQ12; R.qpre = qpre;
        R.Din = (long long)Din; R.Dsub = (long long)Dsub; R.nskip = (long long)nskip;
        R.batch = (int)batch;
        const double scl = std::sqrt(3.0) / 6.0;
        R.delta_over_span = scl * (double)nskip / (double)Din;
        R.w0 = 0.25 + scl;
        R.w1 = 0.25 - scl;
        R.status = status;
        be.template run<KBandCheck>((int)batch, 1, R);     // fnft__misc.c:371-381 (warning only)
        be.template run<KResamplePhase>((int)((batch * Din + 255) / 256), 1, R);
        // inverse DFTs (:395-399), then the weighted pairs (:493-499)
        C.poly = rsX12;
        C.npoly = 2;
        C.Hbuf = rsQ12;
        C.dft_sign = +1;
        rc = run_chirp(C);
        if (rc != NFT_SUCCESS) return rc;
        be.template run<KResampleCombine>((int)((batch * Dsub + 255) / 256), 1, R);
        return run_coeffs(qpre, nullptr, eps_t, kappa, false);
    }

    // ---- level 0 from samples (fnft__akns_fscatter.c:116-917) --------------------------------
    int run_coeffs(const void *d_q, const void *d_r, double eps_t, int kappa, bool clear_status = true)
    {
        (void)clear_status;   // the status word is cleared by whoever reads it (read_status), not per transform
        CoeffParams p;
        p.q = (const cplx *)d_q;
        p.r = (const cplx *)d_r;
        p.body = body[0];
        p.tail = tail[0];
        p.scale = scale[0];
        p.wexp = wexp[0];
        p.status = status;
        p.plane = plane;
        p.eps_t = eps_t;
        p.D = (int)D;
        p.Dpad = (int)Dpad;
        p.batch = (int)batch;
        p.kappa = kappa;
        p.disc = akns_disc;
        p.deg = deg0;
        real_run = want_real && kdv && d_r == rneg;
        p.real_out = real_run ? 1 : 0;
        cur = 0;
        kappa_run = kappa;
        const int spt = use_leaf ? leaf_spt(deg0) : 0;
        const bool leaf = spt > 1 && Dpad >= (size_t)spt;
        // the symmetric form needs r = -kappa*conj(q) (no explicit r) and starts above the
        // direct-product levels, which the leaf kernel guarantees
        ne = (use_sym && leaf && d_r == nullptr && (size_t)deg0 * spt > (size_t)kSchoolMaxDeg) ? 2 : 4;
        p.ne = ne;
        leaf_pending = false;
        if (leaf) {
            LeafParams lp;
            lp.c = p;
            lp.spt = spt;
            start_n = n0 / (size_t)spt;
            start_d = (size_t)deg0 * (size_t)spt;
            // symmetric form with d = 8: the leaf is computed inside the first multi-level launch
            if (use_leaf_multi && use_multi && ne == 2 && start_d == 8 && dbg_flags == 0 && start_n / batch >= 4) {
                leaf_lp = lp;
                leaf_pending = true;
                return NFT_SUCCESS;
            }
            if (!dispatch_leaf(be, lp)) return NFT_EC_NOT_YET_IMPLEMENTED;
            return NFT_SUCCESS;
        }
        const int rl_spt = (real_run && use_rleaf) ? rleaf_strang_samples(akns_disc) : 0;
        if (rl_spt > 1 && Dpad >= (size_t)rl_spt) {
            // even-order splitting schemes on the real path: coefficients from their elementary factors AND the ordered
            // product of rl_spt consecutive samples by direct multiplication (body_rleaf_strang, nft_real.h)
            ne = 4;
            LeafParams lp;
            lp.c = p;
            lp.c.ne = 4;
            lp.spt = rl_spt;
            if (!dispatch_rleaf_strang(be, lp)) return NFT_EC_NOT_YET_IMPLEMENTED;
            start_n = n0 / (size_t)rl_spt;
            start_d = (size_t)deg0 * (size_t)rl_spt;
            return NFT_SUCCESS;
        }
        if (real_run && dispatch_rcoeffs_strang(be, p)) {
            // ... (short signals) coefficients only, composed from their elementary factors
            ne = 4;
        } else if (prog_ptr != nullptr) {
            // schemes of order 5..8: generated coefficient program; the symmetric form can start at
            // level 0 because every level is an FFT level (deg0 > kSchoolMaxDeg)
            ne = (use_sym && d_r == nullptr) ? 2 : 4;
            CoeffProgParams pp;
            pp.c = p;
            pp.c.ne = ne;
            pp.bfrac = prog_bfrac; pp.tgt_ptr = prog_ptr; pp.mw = prog_mw; pp.mfac = prog_fac;
            pp.nB = prog_nB; pp.maxf = prog_maxf;
            be.template run<KCoeffsProg>((int)((n0 + 63) / 64), 1, pp);
        } else if (!dispatch_coeffs(be, p)) return NFT_EC_NOT_YET_IMPLEMENTED;
        start_n = n0;
        start_d = (size_t)deg0;
        return NFT_SUCCESS;
    }

    // ---- level 0 from explicit coefficient matrices in the reference layout (host) ------------
    // p_host: [4][n*(deg0+1)], n = D matrices (fnft__poly_fmult.c:398-401)
    int load_level0_from_host(const std::complex<double> *p_host)
    {
        std::vector<cplx> hb(4 * plane), ht(4 * n0);
        std::vector<double> hs(n0, 1.0);
        const size_t w = (size_t)deg0 + 1;
        for (int e = 0; e < 4; e++)
            for (size_t j = 0; j < Dpad; j++) {
                for (size_t k = 0; k < w; k++) {
                    cplx v = cmake(0.0, 0.0);
                    if (j < D) {
                        const std::complex<double> z = p_host[(size_t)e * D * w + j * w + k];
                        v = cmake(z.real(), z.imag());
                    } else if (k == 0 && (e == 0 || e == 3)) {
                        v = cmake(1.0, 0.0);  // identity pad z^deg * I, fnft__poly_fmult.c:422-438
                    }
                    if (k < (size_t)deg0) hb[(size_t)e * plane + j * deg0 + k] = v;
                    else ht[(size_t)e * n0 + j] = v;
                }
            }
        be.h2d(body[0], hb.data(), hb.size() * sizeof(cplx));
        be.h2d(tail[0], ht.data(), ht.size() * sizeof(cplx));
        be.h2d(scale[0], hs.data(), hs.size() * sizeof(double));
        be.memset0(wexp[0], n0 * sizeof(int));
        cur = 0;
        ne = 4;
        real_run = false;
        start_n = n0;
        start_d = (size_t)deg0;
        return NFT_SUCCESS;
    }

    // ---- product tree (fnft__poly_fmult.c:460-519) ---------------------------------------------
    // ---- level 0 from coefficient matrices in DEVICE memory (reference layout, D matrices of degree deg0) ------
    int load_level0_from_device(const void *d_p)
    {
        ImportParams I;
        I.p = (const cplx *)d_p;
        I.body = body[0]; I.tail = tail[0]; I.scale = scale[0]; I.wexp = wexp[0];
        I.plane = plane;
        I.n = (long long)D; I.npad = (long long)Dpad; I.deg = (long long)deg0;
        const long long tot = 4 * I.npad * (I.deg + 1);
        be.template run<KImportLevel0>((int)((tot + 255) / 256), 1, I);
        cur = 0;
        ne = 4;
        real_run = false;
        start_n = n0;
        start_d = (size_t)deg0;
        return NFT_SUCCESS;
    }

    // the same tree on real coefficients (nft_real.h): folded transforms of length M = nft_real_len(d)
    int run_tree_real()
    {
        size_t n = start_n, d = start_d;
        bool y_from_bridge = false, in_pending = false;
        int zcur = 0, mcur = 0;
        // degrees 3*2^a: the split levels take columns of 3*K rows of kRowGen points -- transform length M = d exactly
        // (nft_real.h) -- when the last level's column length is instantiated
        bool r3 = false;
        if (use_r3 && tw3tab) {
            size_t odd = start_d;
            while (odd % 2 == 0) odd /= 2;
            const size_t dtop = start_d * (start_n / batch) / 2;   // degree of the last level's factors
            r3 = (odd == 3) && dtop % (size_t)kRowGen == 0 && dtop / (size_t)kRowGen <= (size_t)3 * kR3MaxK;
        }
        while (n / batch > 1) {
            TreeLevel L;
            L.body_in = body[cur]; L.tail_in = tail[cur]; L.scale_in = scale[cur];
            L.body_out = body[cur ^ 1]; L.tail_out = tail[cur ^ 1]; L.scale_out = scale[cur ^ 1];
            L.max2_out = max2[mcur ^ 1];
            L.max2_in = max2[mcur];
            L.in_pending = in_pending ? 1 : 0;
            L.wexp_in = wexp[cur];
            L.wexp_out = wexp[cur ^ 1];
            L.plane = plane;
            L.n_in = (int)n;
            L.d = (int)d;
            L.pairs_per_signal = (int)(n / 2 / batch);
            L.ne = 4;
            L.kappa = kappa_run;
            L.dbg = 0;
            const size_t M = nft_real_len(d);
            bool ok;
            if (d <= (size_t)kSchoolMaxDeg) {
                ok = dispatch_rpair_school(be, L);
            } else if (r3 && d % (size_t)kRowGen == 0 && d / (size_t)kRowGen >= 3) {
                BigLevel G;
                std::memset(&G, 0, sizeof(G));
                G.L = L;
                G.Y = Y;
                G.Z = zcur ? Z2 : Z;
                G.N2 = kRowGen;
                G.N1 = (int)(d / (size_t)kRowGen);   // 3*K
                const size_t K = (size_t)G.N1 / 3;
                G.btw = big_tw3(4 * d);              // row twiddle w_{4M}^{(4 k1 - 1) n2}, M = d
                G.rtwist = 1;
                G.row_mod = 4ull * (unsigned long long)d;
                G.tw1 = tw_table(K >= 2 ? K : 2);
                G.tw2 = tw_table((size_t)G.N2);
                G.tw1x2 = tw_table(2 * K);
                G.tw3 = tw3_table((size_t)G.N1);
                G.tw3x2 = tw3_table((size_t)2 * G.N1);
                G.twq = tw3_table((size_t)4 * G.N1);
                G.twq2 = ((size_t)8 * G.N1 <= ((size_t)3 << kTw3MaxLog)) ? tw3_table((size_t)8 * G.N1) : nullptr;
                G.y_unscaled = y_from_bridge ? 1 : 0;
                ok = true;
                if (!y_from_bridge) ok = dispatch_r3col_fwd(be, G);
                if (ok) run_mid(be, G);
                const bool next_split = use_bridge && (n / 2 / batch > 1) && K <= (size_t)kR3BridgeMaxK;
                if (ok) ok = next_split ? dispatch_r3bridge(be, G) : dispatch_r3col_inv(be, G);
                zcur ^= 1;
                y_from_bridge = ok && next_split;
                const bool last_level = (n / 2 / batch <= 1);
                if (ok && last_level) defer_final(L, n / 2);
                in_pending = !last_level;
                mcur ^= 1;
            } else if (M <= (size_t)kRealFusedMaxM) {
                L.tw = tw_table(M);
                L.twx = tw_table(4 * M);
                ok = dispatch_rpair(be, L, (int)M);
            } else {
                BigLevel G;
                std::memset(&G, 0, sizeof(G));
                G.L = L;
                G.Y = Y;
                G.Z = zcur ? Z2 : Z;
                G.N2 = row_len_gen(M);
                G.N1 = (int)(M / (size_t)G.N2);
                if (4 * (size_t)G.N1 > (size_t)kMaxTwTable) return NFT_EC_NOT_YET_IMPLEMENTED;
                G.btw = big_tw(4 * M);   // row twiddle w_{4M}^{(4 k1 - 1) n2}
                G.rtwist = 1;
                G.tw1 = tw_table((size_t)G.N1);
                G.tw2 = tw_table((size_t)G.N2);
                G.tw1x2 = tw_table((size_t)2 * G.N1);
                G.twq = tw_table((size_t)4 * G.N1);
                G.twq2 = (8 * (size_t)G.N1 <= (size_t)kMaxTwTable) ? tw_table((size_t)8 * G.N1) : nullptr;
                G.y_unscaled = y_from_bridge ? 1 : 0;
                ok = true;
                if (!y_from_bridge) ok = dispatch_rcol_fwd(be, G);
                if (ok) run_mid(be, G);
                const bool next_split = use_bridge && (n / 2 / batch > 1) && G.N1 <= kRBridgeMaxN1
                                        && row_len_gen(2 * M) == G.N2;
                if (ok) ok = next_split ? dispatch_rbridge(be, G) : dispatch_rcol_inv(be, G);
                zcur ^= 1;
                y_from_bridge = ok && next_split;
                const bool last_level = (n / 2 / batch <= 1);
                if (ok && last_level) defer_final(L, n / 2);
                in_pending = !last_level;
                mcur ^= 1;
            }
            if (!ok) return NFT_EC_NOT_YET_IMPLEMENTED;
            cur ^= 1;
            n /= 2;
            d *= 2;
        }
        res_deg = D * (size_t)deg0;
        tree_valid = true;
        return NFT_SUCCESS;
    }

    int run_tree()
    {
        final_pending = false;   // an unconsumed root of an earlier run is dropped
        if (real_run) return run_tree_real();
        size_t n = start_n;     // matrices at the current level, all signals
        size_t d = start_d;
        bool y_from_bridge = false;
        bool y_split = false;      // Y holds odd rows only, even rows are the previous Z
        int zcur = 0;
        bool in_pending = false;   // previous level was split: its rescale is still pending
        int mcur = 0;
        int split_idx = 0;
        while (n / batch > 1) {
            TreeLevel L;
            L.body_in = body[cur]; L.tail_in = tail[cur]; L.scale_in = scale[cur];
            L.body_out = body[cur ^ 1]; L.tail_out = tail[cur ^ 1]; L.scale_out = scale[cur ^ 1];
            L.max2_out = max2[mcur ^ 1];
            L.max2_in = max2[mcur];
            L.in_pending = in_pending ? 1 : 0;
            L.wexp_in = wexp[cur];
            L.wexp_out = wexp[cur ^ 1];
            L.plane = plane;
            L.n_in = (int)n;
            L.d = (int)d;
            L.pairs_per_signal = (int)(n / 2 / batch);
            L.ne = ne;
            L.kappa = kappa_run;
            L.dbg = dbg_flags;
            const size_t N = nft_product_len(d);
            L.tw = (N <= (size_t)kMaxTwTable) ? tw_table(N) : nullptr;
            bool ok;
            // consecutive fused levels of the symmetric form in one launch: as many as fit (<= 3)
            int stages = 1;
            if (use_multi && ne == 2 && d > (size_t)kSchoolMaxDeg && N == 2 * d && N >= 16 && dbg_flags == 0) {
                while (stages < 3 && (N << stages) <= (size_t)kFusedMaxN && ((n / batch) >> (stages + 1)) >= 1) stages++;
            }
            if (leaf_pending && stages < 2) {   // cannot happen (start_n/batch >= 4), but never skip the leaf
                if (!dispatch_leaf(be, leaf_lp)) return NFT_EC_NOT_YET_IMPLEMENTED;
                leaf_pending = false;
            }
            if (stages > 1) {
                for (int s = 0; s < stages; s++) L.twm[s] = tw_table(N << s);
                if (leaf_pending) {
                    LeafMultiParams Q;
                    Q.lp = leaf_lp;
                    Q.L = L;
                    ok = dispatch_leaf_multi(be, Q, stages);
                    leaf_pending = false;
                    if (!ok) {   // configuration without a fused instantiation: leaf, then the levels
                        if (!dispatch_leaf(be, leaf_lp)) return NFT_EC_NOT_YET_IMPLEMENTED;
                        ok = dispatch_multi(be, L, (int)N, stages);
                    }
                } else {
                    ok = dispatch_multi(be, L, (int)N, stages);
                }
                if (!ok) return NFT_EC_NOT_YET_IMPLEMENTED;
                cur ^= 1;
                n >>= stages;
                d <<= stages;
                continue;
            }
            if (d <= (size_t)kSchoolMaxDeg) {
                ok = dispatch_pair_school(be, L);
            } else if (N <= fused_max_len()) {
                ok = dispatch_pair_fft(be, L, (int)N);
            } else {
                BigLevel G;
                G.L = L;
                G.Y = Y;
                G.Z = zcur ? Z2 : Z;
                G.Zprev = zcur ? Z : Z2;
                G.y_split = y_split ? 1 : 0;
                G.N2 = (ne == 4 && FA_MID_GEN) ? row_len_gen(N) : kRowTree;
                G.N1 = (int)(N / (size_t)G.N2);
                G.btw = big_tw(N);
                G.tw1 = (G.N1 >= 2) ? tw_table((size_t)G.N1) : nullptr;
                G.tw2 = tw_table((size_t)G.N2);
                G.tw1x2 = tw_table((size_t)2 * G.N1);
                G.btw2 = big_tw(2 * N);
                G.y_unscaled = y_from_bridge ? 1 : 0;
                // first split level: a length-4 column transform of two non-zero rows is done by the row
                // kernel on the fly (saves the column launch and its 32 + 64 MB)
                G.y_direct = (use_direct4 && !y_from_bridge && G.N1 == 4 && N == 2 * d && dbg_flags == 0 && (ne == 2 || !FA_MID_GEN)) ? 1 : 0;
                G.stagger = (n / 2 * (size_t)G.N1 == 512) ? tune_stagger : 0;   // exactly one round of workgroups
                G.stamps = (dbg_stamps && split_idx == stamp_level) ? dbg_stamps : nullptr;
                split_idx++;
                ok = true;
                if (!y_from_bridge && !G.y_direct) ok = dispatch_col_fwd(be, G);
                if (ok) run_mid(be, G);
                // bridge straight into the next level's column step when that level is split too
                const bool can_double = use_doubling;   // N = 2d and N > 2d alike (body_col_bridge2)
                const bool next_split = use_bridge && (n / 2 / batch > 1) && G.N1 <= (can_double ? 4096 : 512)
                                        && nft_product_len(2 * d) == 2 * N
                                        && ((ne == 4 && FA_MID_GEN) ? row_len_gen(2 * N) : kRowTree) == G.N2;
                const bool doubling = next_split && can_double;
                if (ok) {
                    if (doubling) ok = dispatch_col_bridge2(be, G);
                    else if (next_split) ok = dispatch_col_bridge(be, G);
                    else ok = dispatch_col_inv(be, G);
                }
                y_split = ok && doubling;
                zcur ^= 1;
                y_from_bridge = ok && next_split;
                // the consumer of the next level finalizes this one; the last level needs a kernel
                const bool last_level = (n / 2 / batch <= 1);
                if (ok && last_level) defer_final(L, n / 2);
                in_pending = !last_level;
                mcur ^= 1;
            }
            if (!ok) return NFT_EC_NOT_YET_IMPLEMENTED;
            cur ^= 1;
            n /= 2;
            d *= 2;
        }
        res_deg = D * (size_t)deg0;
        tree_valid = true;
        return NFT_SUCCESS;
    }

    // ---- result in the reference layout (fnft__poly_fmult.c:522-538) ---------------------------
    // dst / stride / unscale: the result straight into a caller's strided device array, times 2^W (layer peeling)
    void export_tm(cplx *dst = nullptr, size_t stride = 0, bool unscale = false)
    {
        ensure_final();
        ExportParams E;
        E.body = body[cur]; E.tail = tail[cur]; E.scale = scale[cur];
        E.out = dst ? dst : tm_out;
        E.out_stride = dst ? (long long)stride : 0;
        E.W = unscale ? wexp[cur] : nullptr;
        E.plane = plane;
        E.deg_tot = (long long)(Dpad * (size_t)deg0);
        E.deg = (long long)res_deg;
        E.batch = (int)batch;
        E.ne = ne;
        E.kappa = kappa_run;
        E.real_layout = real_run ? 1 : 0;
        const long long tot = 4 * (E.deg + 1) * E.batch;
        be.template run<KExportTm>((int)((tot + 255) / 256), 1, E);
    }

    // ---- chirp z + epilogue (fnft_nsev.c:744-891) ----------------------------------------------
    struct Contspec {
        double T[2], XI[2];
        int nse_disc;
        int cstype;
        int normalization_flag;
    };

    void fill_chirp_geometry(ChirpParams &C, size_t L)
    {
        C.N2 = kRowChirp;
        C.N1 = (int)(L / kRowChirp);
        C.btw = big_tw(L);
        C.tw1 = tw_table((size_t)C.N1);
        C.tw2 = tw_table(kRowChirp);
        C.jobs_per_group = 2;
    }

    int run_contspec(void *d_contspec, const Contspec &cs) { return run_contspec_impl(d_contspec, cs, nullptr, 0); }

    // nsev_compute_contspec (src/fnft_nsev.c:744-891) on a transfer matrix the CALLER supplies: d_tm holds, per
    // signal, [r11|r12|r21|r22] of deg+1 = D*deg0+1 coefficients each (highest power first, the layout of
    // fnft__poly_fmult2x2 / fnft_amd_plan_get_transfer_matrix_device), true matrix = stored * 2^W.
    int run_contspec_tm(void *d_contspec, const Contspec &cs, const void *d_tm, int W)
    {
        if (!d_tm) return NFT_EC_INVALID_ARGUMENT;
        if (!wuser && !alloc(wuser, batch)) return NFT_EC_NOMEM;
        std::vector<int> hw(batch, W);
        be.h2d(wuser, hw.data(), batch * sizeof(int));
        return run_contspec_impl(d_contspec, cs, (const cplx *)d_tm, 1);
    }

    int run_contspec_impl(void *d_contspec, const Contspec &cs, const cplx *d_tm, int from_tm)
    {
        // step size and phase factors refer to D_given = D/upsampling (fnft_nsev.c:766-774);
        // lambda -> z uses degree*upsampling (fnft__akns_discretization.c:204-219)
        const double deg1 = (double)(deg0 * ups);
        const size_t Dg = D / (size_t)ups;
        const double eps_t = (cs.T[1] - cs.T[0]) / (double)(Dg - 1);
        const double eps_xi = (cs.XI[1] - cs.XI[0]) / (double)(M - 1);
        // lambda -> z, fnft__akns_discretization.c:204-219 called from fnft_nsev.c:822-827
        const double phiV = 2.0 * eps_xi * eps_t / deg1;
        const double phiA = 2.0 * (-cs.XI[0]) * eps_t / deg1;
        const std::complex<double> V(std::cos(phiV), std::sin(phiV));
        const std::complex<double> A(std::cos(phiA), std::sin(phiA));
        const std::complex<double> lV = nft_clog(V), lA = nft_clog(A);

        ChirpParams C;
        std::memset(&C, 0, sizeof(C));
        C.body = body[cur]; C.tail = tail[cur]; C.scale = scale[cur];
        C.poly = from_tm ? d_tm : nullptr;
        C.poly_tm = from_tm;
        C.plane = plane;
        C.deg_tot = (long long)(Dpad * (size_t)deg0);
        C.deg = from_tm ? (long long)(D * (size_t)deg0) : (long long)res_deg;
        C.batch = (int)batch;
        C.npoly = 2;
        C.ne = from_tm ? 4 : ne;
        C.entry[0] = 0;                    // H11, fnft_nsev.c:829
        C.entry[1] = (C.ne == 4) ? 2 : 1;  // H21, fnft_nsev.c:832 (plane 1 in the symmetric form)
        C.logA[0] = lA.real(); C.logA[1] = lA.imag();
        C.logW[0] = lV.real(); C.logW[1] = lV.imag();
        C.M = (long long)M;
        C.Ybuf = chY; C.Vbuf = chV; C.Hbuf = nullptr;
        fill_chirp_geometry(C, Lc);
        C.contspec = (cplx *)d_contspec;
        C.W = from_tm ? wuser : wexp[cur];
        C.status = status;
        C.xi0 = cs.XI[0];
        C.eps_xi = eps_xi;
        // phase factors, fnft__nse_discretization.c:240-379, boundary coefficient 0.5
        const double bc = 0.5;
        const bool shifted = (cs.nse_disc == 0 /*MODAL*/ || cs.nse_disc == 4 /*2SPLIT2A*/);
        C.pf_rho = -2.0 * (cs.T[1] + eps_t * bc) + (shifted ? eps_t / deg1 : 0.0);
        C.pf_a = -eps_t * (double)Dg + (cs.T[1] + eps_t * bc) - (cs.T[0] - eps_t * bc);
        C.pf_b = -eps_t * (double)Dg - (cs.T[1] + eps_t * bc) - (cs.T[0] - eps_t * bc)
                 + (shifted ? eps_t / deg1 : 0.0);
        C.cstype = cs.cstype;
        C.use_W = 1;  // W is the exponent actually taken out, whatever normalization_flag says
        if (!from_tm) hand_final_to(C);
        return run_chirp_cached(C);
    }

    // fnft_kdvv.c:126-209 (tf2contspec_negxi): entries 12 and 22 on the grid -(XI0 + m*eps_xi)
    int run_contspec_kdv(void *d_contspec, const double T[2], const double XI[2], bool scheme_2A)
    {
        const double deg1 = (double)deg0;
        const double eps_t = (T[1] - T[0]) / (double)(D - 1);
        const double eps_xi = (XI[1] - XI[0]) / (double)(M - 1);
        const double phiV = -2.0 * eps_xi * eps_t / deg1;   // :159
        const double phiA = 2.0 * XI[0] * eps_t / deg1;     // :160
        const std::complex<double> V(std::cos(phiV), std::sin(phiV));
        const std::complex<double> A(std::cos(phiA), std::sin(phiA));
        const std::complex<double> lV = nft_clog(V), lA = nft_clog(A);
        ChirpParams C;
        std::memset(&C, 0, sizeof(C));
        C.body = body[cur]; C.tail = tail[cur]; C.scale = scale[cur];
        C.plane = plane;
        C.deg_tot = (long long)(Dpad * (size_t)deg0);
        C.deg = (long long)res_deg;
        C.batch = (int)batch;
        C.npoly = 2;
        C.ne = ne;
        if (ne != 4) return NFT_EC_OTHER;   // explicit r: the general form
        C.entry[0] = 1;   // H12, :165
        C.entry[1] = 3;   // H22, :172
        C.logA[0] = lA.real(); C.logA[1] = lA.imag();
        C.logW[0] = lV.real(); C.logW[1] = lV.imag();
        C.M = (long long)M;
        C.Ybuf = chY; C.Vbuf = chV;
        fill_chirp_geometry(C, Lc);
        C.contspec = (cplx *)d_contspec;
        C.W = wexp[cur];
        C.status = status;
        C.xi0 = -XI[0];
        C.eps_xi = -eps_xi;
        C.pf_rho = 2.0 * (T[1] + 0.5 * eps_t);          // :199, boundary coefficient 0.5
        C.pf_a = scheme_2A ? -eps_t / deg1 : 0.0;       // :186-195
        C.cstype = 10;
        C.real_layout = real_run ? 1 : 0;
        hand_final_to(C);
        return run_chirp_cached(C);
    }

    // the root's pending finalize rides on the chirp transform's column kernel (one matrix per signal at the root)
    void hand_final_to(ChirpParams &C)
    {
        if (!final_pending) return;
        if (final_n != batch) { ensure_final(); return; }
        C.fin_max2 = final_L.max2_out;
        C.fin_scale = final_L.scale_out;
        C.fin_wexp = final_L.wexp_out;
        final_pending = false;
    }

    // the spectrum of the chirp filter only depends on W, M, the degree and the transform length:
    // repeated transforms on the same grids (T, XI fixed) reuse it
    int run_chirp_cached(ChirpParams &C)
    {
        const double key[4] = {C.logW[0], C.logW[1], (double)C.M, (double)(C.deg + 1)};
        const bool hit = vs_valid && std::memcmp(key, vs_key, sizeof(key)) == 0;
        C.VS = chVS;
        C.v_mode = hit ? 2 : 1;
        const int rc = run_chirp(C);
        if (rc == NFT_SUCCESS) { std::memcpy(vs_key, key, sizeof(key)); vs_valid = true; }
        return rc;
    }

    int run_chirp(const ChirpParams &C)
    {
        if (!dispatch_chirp_col_fwd(be, C)) return NFT_EC_NOT_YET_IMPLEMENTED;
        const int njobs = C.batch * C.npoly;
        be.template run<KChirpRows>(C.N1, (njobs + C.jobs_per_group - 1) / C.jobs_per_group, C);
        if (!dispatch_chirp_col_inv(be, C)) return NFT_EC_NOT_YET_IMPLEMENTED;
        return NFT_SUCCESS;
    }

    // stand-alone chirp z-transform of one host polynomial (fnft__poly_chirpz.c:33-105)
    // d_out != NULL: the M values stay on the device (d_out), `result` is not touched
    static int chirpz_host(BE &be, size_t deg, const std::complex<double> *p,
                           std::complex<double> A, std::complex<double> Wc, size_t Mo,
                           std::complex<double> *result, cplx *d_out = nullptr)
    {
        NftPlan pl(be, 2, 0, 1, 0, 1);  // only the twiddle tables of the plan are used
        size_t L = nft_nextpow2(deg + 1 + Mo - 1);
        if (L < 2 * (size_t)kRowChirp) L = 2 * (size_t)kRowChirp;
        if (L > kMaxSplitChirp) return NFT_EC_NOT_YET_IMPLEMENTED;
        bool ok = pl.alloc(pl.twtab, (size_t)2 * kMaxTwTable) && pl.alloc(pl.twlo, kTwLoEntries);
        cplx *dp = nullptr, *dY = nullptr, *dV = nullptr, *dH = nullptr;
        int *dstatus = nullptr;
        ok = ok && pl.alloc(dp, deg + 1) && pl.alloc(dY, L) && pl.alloc(dV, L) && pl.alloc(dH, Mo)
             && pl.alloc(dstatus, 4);
        int rc = NFT_EC_NOMEM;
        if (ok) {
            pl.upload_twiddles();
            be.h2d(dp, p, (deg + 1) * sizeof(cplx));
            be.memset0(dstatus, 4 * sizeof(int));
            const std::complex<double> lA = nft_clog(A), lW = nft_clog(Wc);
            ChirpParams C;
            std::memset(&C, 0, sizeof(C));
            C.poly = dp;
            C.deg = (long long)deg;
            C.batch = 1;
            C.npoly = 1;
            C.logA[0] = lA.real(); C.logA[1] = lA.imag();
            C.logW[0] = lW.real(); C.logW[1] = lW.imag();
            C.M = (long long)Mo;
            C.Ybuf = dY; C.Vbuf = dV; C.Hbuf = d_out ? d_out : dH;
            pl.fill_chirp_geometry(C, L);
            C.status = dstatus;
            C.cstype = -1;
            rc = pl.run_chirp(C);
            if (rc == NFT_SUCCESS) {
                if (!d_out) be.d2h(result, dH, Mo * sizeof(cplx));
                rc = be.sync();
            }
        }
        be.free(dp); be.free(dY); be.free(dV); be.free(dH); be.free(dstatus);
        be.free(pl.twtab); be.free(pl.twlo);
        return rc;
    }

    // stand-alone band-limited shift of one host signal by delta (fnft__misc.c:326-407): DFT of any length
    // (chirp kernels in DFT mode), phase ramp exp(2 pi i delta f), inverse DFT, 1/D
    static int resample_host(BE &be, size_t Dn, double eps_t, const std::complex<double> *q, double delta,
                             std::complex<double> *q_new, int *warn_not_bandlimited = nullptr)
    {
        NftPlan pl(be, 2, 0, 1, 0, 1);  // only the twiddle tables of the plan are used
        size_t L = nft_nextpow2(2 * Dn - 1);
        if (L < 2 * (size_t)kRowChirp) L = 2 * (size_t)kRowChirp;
        if (L > kMaxSplitChirp) return NFT_EC_NOT_YET_IMPLEMENTED;
        bool ok = pl.alloc(pl.twtab, (size_t)2 * kMaxTwTable) && pl.alloc(pl.twlo, kTwLoEntries);
        cplx *dq = nullptr, *dX = nullptr, *dX12 = nullptr, *dQ12 = nullptr, *dY = nullptr, *dV = nullptr;
        int *dstatus = nullptr;
        ok = ok && pl.alloc(dq, Dn) && pl.alloc(dX, Dn) && pl.alloc(dX12, 2 * Dn) && pl.alloc(dQ12, 2 * Dn)
             && pl.alloc(dY, 2 * L) && pl.alloc(dV, L) && pl.alloc(dstatus, 4);
        int rc = NFT_EC_NOMEM;
        if (ok) {
            pl.upload_twiddles();
            be.h2d(dq, q, Dn * sizeof(cplx));
            be.memset0(dstatus, 4 * sizeof(int));
            ChirpParams C;
            std::memset(&C, 0, sizeof(C));
            C.poly = dq;
            C.deg = (long long)Dn - 1;
            C.batch = 1;
            C.npoly = 1;
            C.M = (long long)Dn;
            C.Ybuf = dY; C.Vbuf = dV; C.Hbuf = dX;
            pl.fill_chirp_geometry(C, L);
            C.status = dstatus;
            C.cstype = -1;
            C.dft_len = (long long)Dn;
            C.dft_sign = -1;
            rc = pl.run_chirp(C);                                   // X = DFT(q), :366-370
            if (rc == NFT_SUCCESS) {
                ResampleParams R;
                std::memset(&R, 0, sizeof(R));
                R.X = dX; R.X12 = dX12; R.Q12 = dQ12; R.qpre = nullptr;
                R.Din = (long long)Dn; R.Dsub = (long long)Dn; R.nskip = 1;
                R.batch = 1;
                R.delta_over_span = delta / ((double)Dn * eps_t);    // freq[i]*delta, :383-393
                R.status = dstatus;
                be.template run<KBandCheck>(1, 1, R);                // :371-381
                be.template run<KResamplePhase>((int)((Dn + 255) / 256), 1, R);
                C.poly = dX12;
                C.npoly = 2;
                C.Hbuf = dQ12;
                C.dft_sign = +1;
                rc = pl.run_chirp(C);                               // inverse DFTs of the -delta and +delta copies
            }
            if (rc == NFT_SUCCESS) {
                int hst[4] = {0, 0, 0, 0};
                be.d2h(hst, dstatus, sizeof(hst));
                be.d2h(q_new, dQ12 + Dn, Dn * sizeof(cplx));        // the +delta copy
                rc = be.sync();
                if (warn_not_bandlimited) *warn_not_bandlimited = (hst[0] & 4) ? 1 : 0;
                const double inv = 1.0 / (double)Dn;                // :395-399
                if (rc == NFT_SUCCESS)
                    for (size_t i = 0; i < Dn; i++) q_new[i] *= inv;
            }
        }
        be.free(dq); be.free(dX); be.free(dX12); be.free(dQ12); be.free(dY); be.free(dV); be.free(dstatus);
        be.free(pl.twtab); be.free(pl.twlo);
        return rc;
    }

    // device-side status word -> return code (after the stream has been waited for)
    int read_status()
    {
        int h[4] = {0, 0, 0, 0};
        be.d2h(h, status, sizeof(h));
        const int rc = be.sync();
        if (rc != NFT_SUCCESS) return rc;
        // sticky bits of every transform since the last read; cleared here (off the transforms' critical path)
        be.memset0(status, 4 * sizeof(int));
        last_warn = (h[0] & 4) ? 1 : 0;            // fnft__misc.c:371-381: not an error
        if (h[0] & 8) return NFT_EC_INVALID_ARGUMENT;   // real-coefficient path on a potential that is not real
        if (h[0] & 1) return -NFT_EC_OTHER;        // fnft__akns_fscatter.c:122-126 via CHECK_RETCODE
        if (h[0] & 2) return -NFT_EC_DIV_BY_ZERO;  // fnft_nsev.c:850-853 via CHECK_RETCODE
        return NFT_SUCCESS;
    }
};
