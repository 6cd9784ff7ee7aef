// nft_dispatch.h -- kernel functors (one per kernel body + tiling) and the run-time -> compile-time
// dispatch shared by the HIP back end (hip_backend.hip) and the CPU lane emulator (tests/emu).
//
// A kernel functor K provides: Params, THREADS, lds_bytes(), body(const Params&).
// A back end BE provides: template<class K> void run(int gx, int gy, const typename K::Params&).
#pragma once
#include "nft_kernels.h"
#include "nft_real.h"

#ifndef FA_ROW_TREE
#define FA_ROW_TREE 2048
#endif
constexpr int kRowTree = FA_ROW_TREE;      // N2 of the split transforms of the product tree
#ifndef FA_ROW_CHIRP
#define FA_ROW_CHIRP 4096
#endif
constexpr int kRowChirp = FA_ROW_CHIRP;     // N2 of the split transforms of the chirp z-transform
constexpr int kFusedMaxN = 4096;    // largest pair product done by one workgroup
constexpr int kSchoolMaxDeg = 3;    // direct products up to this input degree

// ---- simple one-lane-per-item kernels --------------------------------------------------------
template <int DEG> struct KCoeffs {
    using Params = CoeffParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_coeffs<DEG>(p); }
};
struct KCoeffsProg {
    using Params = CoeffProgParams;
    static constexpr int THREADS = 64;
    static constexpr size_t lds_bytes() { return (size_t)(24 + 16) * 64 * sizeof(cplx); }
    static FA_DEV void body(const Params &p) { body_coeffs_prog(p); }
};
struct KResamplePhase {
    using Params = ResampleParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_resample_phase(p); }
};
struct KBandCheck {
    using Params = ResampleParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 3 * 256 * sizeof(double); }
    static FA_DEV void body(const Params &p) { body_band_check(p); }
};
struct KResampleCombine {
    using Params = ResampleParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_resample_combine(p); }
};
template <bool BACKWARD> struct KBsChunk {
    using Params = BsParams;
    static constexpr int THREADS = 64;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_bs_chunk<BACKWARD>(p); }
};
template <bool BACKWARD> struct KBsCombine {
    using Params = BsParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 256 * 12 * sizeof(cplx); }
    static FA_DEV void body(const Params &p) { body_bs_combine<BACKWARD>(p); }
};
struct KBsMatrix {
    using Params = BsParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 256 * 8 * sizeof(cplx); }
    static FA_DEV void body(const Params &p) { body_bs_matrix(p); }
};
struct KBsPhi {
    using Params = BsParams;
    static constexpr int THREADS = 64;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_bs_phi(p); }
};
struct KBsMetric {
    using Params = BsParams;
    static constexpr int THREADS = 64;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_bs_metric(p); }
};
struct KBsPick {
    using Params = BsParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 256 * (sizeof(double) + sizeof(int)); }
    static FA_DEV void body(const Params &p) { body_bs_pick(p); }
};
struct KBsPsi {
    using Params = BsParams;
    static constexpr int THREADS = 64;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_bs_psi(p); }
};
struct KGridMark {
    using Params = GridSearchParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_gridsearch_mark(p); }
};
struct KGridMarkPh {
    using Params = GridSearchParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_gridsearch_mark_ph(p); }
};
struct KCompactCount {
    using Params = GridSearchParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 256 * sizeof(int); }
    static FA_DEV void body(const Params &p) { body_compact_count(p); }
};
struct KCompactScatter {
    using Params = GridSearchParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 256 * sizeof(int); }
    static FA_DEV void body(const Params &p) { body_compact_scatter(p); }
};
struct KInvOp {
    using Params = InvOpParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 256 * sizeof(double); }
    static FA_DEV void body(const Params &p) { body_inv_op(p); }
};
struct KPeelImport {
    using Params = PeelIoParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_peel_import(p); }
};
template <int N> struct KPeelProduct {
    using Params = PeelProdParams;
    static constexpr int R = 8;
    static constexpr int THREADS = 4 * N / R;
    static constexpr int MIN_WAVES = (THREADS >= 1024) ? 4 : ((THREADS >= 512) ? 2 : 1);
    static constexpr size_t lds_bytes() { return (size_t)4 * N * sizeof(cplx); }
    static FA_DEV void body(const Params &p) { body_peel_product<N, R>(p); }
};
struct KPeelLeaf {
    using Params = PeelLeafParams;
    static constexpr int THREADS = 192;
    static constexpr size_t lds_bytes() { return 256 * sizeof(cplx) + 64; }
    static FA_DEV void body(const Params &p) { body_peel_leaf(p); }
};
struct KInvSolitons {
    using Params = InvDsParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_inv_solitons(p); }
};
struct KInvCdt {
    using Params = InvDsParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_inv_cdt(p); }
};
struct KAberthNewton {
    using Params = AberthParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 2 * 512 * (sizeof(cplx) + sizeof(double)); }
    static FA_DEV void body(const Params &p) { body_aberth_newton<512>(p); }
};
struct KAberthSum {
    using Params = AberthParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 256 * sizeof(cplx); }
    static FA_DEV void body(const Params &p) { body_aberth_sum<256>(p); }
};
struct KAberthApply {
    using Params = AberthParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_aberth_apply(p); }
};
template <int DEG> struct LeafCfg {
    static constexpr int SPT = (DEG == 1) ? 8 : (DEG == 2 ? 4 : 2);
};
template <int DEG> struct KLeaf {
    using Params = LeafParams;
    static constexpr int THREADS = 128;
    static constexpr size_t lds_bytes() { return (size_t)THREADS * DEG * LeafCfg<DEG>::SPT * sizeof(cplx); }
    static FA_DEV void body(const Params &p) { body_leaf<DEG, LeafCfg<DEG>::SPT>(p); }
};
template <int DEG> struct KPairSchool {
    using Params = TreeLevel;
    static constexpr int THREADS = 128;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_pair_school<DEG>(p); }
};
struct KFinalizeScales {
    using Params = TreeLevel;
    static constexpr int THREADS = 64;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_finalize_scales(p); }
};
struct KImportLevel0 {
    using Params = ImportParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_import_level0(p); }
};
struct KExportTm {
    using Params = ExportParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_export_tm(p); }
};

// ---- fused pair product ----------------------------------------------------------------------
#ifndef FA_PAIR_R_SYM
#define FA_PAIR_R_SYM 8
#endif
#ifndef FA_PAIR_R_GEN2K
#define FA_PAIR_R_GEN2K 4   // N = 2048, general form: 512 lanes, 210 VGPRs, 2 waves/SIMD (8: 382 + 126 registers at 1)
#endif
#ifndef FA_PAIR_R_GEN
#define FA_PAIR_R_GEN 4   // general form at 4 points per lane: 170-200 VGPRs, 2 waves/SIMD (8: 320-416 VGPRs at 1 wave/SIMD;
#endif                    // fnft_kdvv cfg 5: the eight single-launch levels 2.66 -> 2.19 ms)
template <int N, int NE = 4> struct PairCfg {
    // points per lane: 8; (experiment knobs: 4 for N <= 1024, symmetric and general form separately)
    static constexpr int R = (N >= 16 && N <= 1024) ? (NE == 2 ? FA_PAIR_R_SYM : FA_PAIR_R_GEN)
                             : ((N == 2048 && NE == 4) ? FA_PAIR_R_GEN2K : 8);
    static constexpr int THREADS = (N / R > 256) ? N / R : 256;
    static constexpr int B = THREADS / (N / R);
    static constexpr bool DB = (N <= 2048);  // N = 4096: one 64 KB buffer
};
template <int N, int NE> struct KPairFft {
    using Params = TreeLevel;
    using C = PairCfg<N, NE>;
    static constexpr int THREADS = C::THREADS;
    // general form: 1 wave/SIMD for 256-lane groups (no scratch spills at ~400 registers); the
    // 512-lane N = 4096 group needs 2 waves/SIMD to be resident at all.  The symmetric form
    // holds half the spectra and fits 2 waves/SIMD.
    static constexpr int MIN_WAVES = (NE == 4 && C::R == 4) ? 2 : ((C::THREADS > 256 || NE == 2) ? (C::R == 4 ? 4 : 2) : 1);
    static constexpr size_t lds_bytes()
    {
        return ((N > C::R && C::DB) ? (size_t)2 : (size_t)1) * N * C::B * sizeof(cplx)
               + ((N > C::R && (N <= 512 || N == 4096)) ? (size_t)N * sizeof(cplx) : 0)
               + (size_t)C::B * 8;
    }
    static FA_DEV void body(const Params &p) { body_pair_fft<N, C::R, C::B, C::DB, NE>(p); }
};

// ---- split transforms ------------------------------------------------------------------------
template <int N1> struct ColCfg {
    // N1 <= 16: one lane per column, no LDS.  Larger: 16 points per lane so that a tile is
    // BC >= 8 columns wide (global rows of >= 128 bytes), one LDS buffer (two barriers/exchange).
#ifndef FA_COL_R_BIG
#define FA_COL_R_BIG 16
#endif
    // points per lane: the whole column up to 16; 16 (two passes up to 256); knob for the longer ones
#ifndef FA_COL_R_MID
#define FA_COL_R_MID 16
#endif
    static constexpr int R = (N1 <= 16) ? N1 : ((N1 >= 512) ? FA_COL_R_BIG : FA_COL_R_MID);
#ifndef FA_COL_T
#define FA_COL_T 256
#endif
    static constexpr int THREADS = (N1 <= 16) ? 256 : ((N1 <= 256) ? FA_COL_T : 512);
    static constexpr int BC = THREADS / (N1 / R);
    static constexpr bool DB = false;
    static constexpr size_t lds_bytes()
    {
        return (N1 > R) ? (size_t)(DB ? 2 : 1) * N1 * BC * sizeof(cplx) : 0;
    }
};
// chirp z-transform column steps: 8 points per lane for the long columns (measured: 0.141 -> 0.120 ms at
// L = 2^21, while the tree's inverse column kernel is faster with 16)
template <int N1> struct ChirpColCfg {
#ifndef FA_CHIRP_R_BIG
#define FA_CHIRP_R_BIG 8
#endif
    // (8192: 16 points per lane, one column per 512-lane workgroup -- the any-length DFTs of the inverse transform at
    // D = 2^20 and oversampling 8; slow, but there)
    static constexpr int R = (N1 <= 16) ? N1 : ((N1 >= 8192) ? 16 : ((N1 >= 512) ? FA_CHIRP_R_BIG : 16));
    static constexpr int THREADS = (N1 <= 16) ? 256 : ((N1 <= 256) ? 256 : 512);
    static constexpr int BC = THREADS / (N1 / R);
    static constexpr bool DB = false;
    static constexpr size_t lds_bytes() { return (N1 > R) ? (size_t)N1 * BC * sizeof(cplx) : 0; }
};
template <int N1> struct KColFwd {
    using Params = BigLevel;
    using C = ColCfg<N1>;
    static constexpr int THREADS = C::THREADS;
    static constexpr size_t lds_bytes() { return C::lds_bytes(); }
    static FA_DEV void body(const Params &p) { body_col_fwd<N1, C::R, C::BC, C::DB>(p); }
};
template <int N1> struct KColInv {
    using Params = BigLevel;
    using C = ColCfg<N1>;
    static constexpr int THREADS = C::THREADS;
    static constexpr size_t lds_bytes() { return C::lds_bytes(); }
    static FA_DEV void body(const Params &p) { body_col_inv<N1, C::R, C::BC, C::DB>(p); }
};
#ifndef FA_BRIDGE_R
#define FA_BRIDGE_R 16
#endif
#ifndef FA_BRIDGE_T
#define FA_BRIDGE_T 256
#endif
template <int N1> struct BridgeCfg {
    static constexpr int R = (N1 <= 16) ? N1 : FA_BRIDGE_R;
    static constexpr int THREADS = (N1 <= 16) ? 256 : FA_BRIDGE_T;
    static constexpr int BC = THREADS / (N1 / R);
    static constexpr bool DB = false;
    static constexpr size_t lds_bytes() { return (N1 > R) ? (size_t)2 * N1 * BC * sizeof(cplx) : 0; }
};
template <int N1> struct KColBridge {
    using Params = BigLevel;
    using C = BridgeCfg<N1>;
    static constexpr int THREADS = C::THREADS;
    static constexpr size_t lds_bytes() { return C::lds_bytes(); }
    static FA_DEV void body(const Params &p) { body_col_bridge<N1, C::R, C::BC, C::DB>(p); }
};
template <int N1> struct KColBridge2 {   // spectral doubling: same tiling as the plain column steps
    using Params = BigLevel;
    using C = ColCfg<N1>;
    static constexpr int THREADS = C::THREADS;
    static constexpr size_t lds_bytes() { return C::lds_bytes(); }
    static FA_DEV void body(const Params &p) { body_col_bridge2<N1, C::R, C::BC, C::DB>(p); }
};
#ifndef FA_MID_R
#define FA_MID_R 8   // points per lane of the row kernel (4: 512 lanes per row, 4 waves per SIMD)
#endif
#ifndef FA_MID4_R
#define FA_MID4_R 4   // points per lane of the general 4-entry row kernel: 512 lanes per row, 232 VGPRs, 2 waves per SIMD
                     // (8: 256 lanes, 394-410 VGPRs, 1 wave per SIMD; cfg 5 row kernel 390 -> 345 us)
#endif
template <int NE> struct KMid {
    using Params = BigLevel;
    static constexpr int R = (NE == 4) ? FA_MID4_R : FA_MID_R;
    static constexpr int THREADS = kRowTree / R;
#ifndef FA_MID4_WAVES
#define FA_MID4_WAVES 1   // general 4-entry form: 1 = 394 VGPRs, no spills (2: 186 spilled); cfg 5 split levels 6.35 -> 6.17 ms
#endif
    static constexpr int MIN_WAVES = (NE == 4) ? (R == 4 ? 2 : FA_MID4_WAVES) : (R == 4 ? 4 : 2);
    static constexpr size_t lds_bytes() { return (size_t)2 * kRowTree * sizeof(cplx); }
    static FA_DEV void body(const Params &p) { body_mid<kRowTree, R, NE>(p); }
};
template <int N1, bool DFT = false> struct KChirpColFwd {
    using Params = ChirpParams;
    using C = ChirpColCfg<N1>;
    static constexpr int THREADS = C::THREADS;
    static constexpr size_t lds_bytes() { return C::lds_bytes(); }
    static FA_DEV void body(const Params &p) { body_chirp_col_fwd<N1, C::R, C::BC, C::DB, DFT>(p); }
};
template <int N1, bool DFT = false, bool KDV = false> struct KChirpColInv {
    using Params = ChirpParams;
    using C = ChirpColCfg<N1>;
    static constexpr int THREADS = C::THREADS;
    static constexpr size_t lds_bytes() { return C::lds_bytes(); }
    static FA_DEV void body(const Params &p) { body_chirp_col_inv<N1, C::R, C::BC, C::DB, DFT, KDV>(p); }
};
struct KChirpRows {
    using Params = ChirpParams;
    static constexpr int R = 8;
    static constexpr int THREADS = kRowChirp / R;
#ifndef FA_CHIRP_ROWS_DB
#define FA_CHIRP_ROWS_DB 1
#endif
    static constexpr bool DB = FA_CHIRP_ROWS_DB != 0;
    static constexpr int MIN_WAVES = DB ? 2 : 4;
    static constexpr size_t lds_bytes() { return (size_t)(DB ? 2 : 1) * kRowChirp * sizeof(cplx); }
    static FA_DEV void body(const Params &p) { body_chirp_rows<kRowChirp, R, DB>(p); }
};

// ---------------------------------------------------------------------------------------------
// dispatch: returns false if the size is not instantiated
// ---------------------------------------------------------------------------------------------
template <class BE> bool dispatch_coeffs(BE &be, const CoeffParams &p)
{
    const long long n = (long long)p.batch * p.Dpad;
    const int g = (int)((n + 255) / 256);
    switch (p.deg) {
    case 1: be.template run<KCoeffs<1>>(g, 1, p); return true;
    case 2: be.template run<KCoeffs<2>>(g, 1, p); return true;
    case 3: be.template run<KCoeffs<3>>(g, 1, p); return true;
    case 4: be.template run<KCoeffs<4>>(g, 1, p); return true;
    default: return false;
    }
}

// samples per lane of the leaf kernel for this degree (0: no leaf kernel)
inline int leaf_spt(int deg)
{
    switch (deg) {
    case 1: return LeafCfg<1>::SPT;
    case 2: return LeafCfg<2>::SPT;
    case 3: return LeafCfg<3>::SPT;
    case 4: return LeafCfg<4>::SPT;
    default: return 0;
    }
}
template <class BE> bool dispatch_leaf(BE &be, const LeafParams &p)
{
    const long long n = (long long)p.c.batch * (p.c.Dpad / p.spt);
    const int g = (int)((n + 127) / 128);
    switch (p.c.deg) {
    case 1: be.template run<KLeaf<1>>(g, 1, p); return true;
    case 2: be.template run<KLeaf<2>>(g, 1, p); return true;
    case 3: be.template run<KLeaf<3>>(g, 1, p); return true;
    case 4: be.template run<KLeaf<4>>(g, 1, p); return true;
    default: return false;
    }
}

template <class BE> bool dispatch_pair_school(BE &be, const TreeLevel &L)
{
    const int pairs = L.n_in / 2;
    const int g = (pairs + 127) / 128;
    switch (L.d) {
    case 1: be.template run<KPairSchool<1>>(g, 1, L); return true;
    case 2: be.template run<KPairSchool<2>>(g, 1, L); return true;
    case 3: be.template run<KPairSchool<3>>(g, 1, L); return true;
    default: return false;
    }
}

template <class BE, int N> void run_pair_fft(BE &be, const TreeLevel &L)
{
    const int pairs = L.n_in / 2;
    constexpr int B4 = PairCfg<N, 4>::B, B2 = PairCfg<N, 2>::B;
    if (L.ne == 4) be.template run<KPairFft<N, 4>>((pairs + B4 - 1) / B4, 1, L);
    else be.template run<KPairFft<N, 2>>((pairs + B2 - 1) / B2, 1, L);
}
// STAGES levels in one launch (symmetric form): T = 256 lanes (512 when the last stage is 4096 long)
template <int N0, int STAGES> struct MultiCfg {
    static constexpr int R = 8;
    static constexpr int NF = N0 << (STAGES - 1);
#ifndef FA_MULTI_T
#define FA_MULTI_T 256   // lanes of a fused-level workgroup whose last stage is shorter than 4096 (knob: 64 = one wave, 128)
#endif
    static constexpr int TMIN = NF / R;                                   // one transform of the last stage
    static constexpr int THREADS = (NF >= 4096) ? 512 : ((FA_MULTI_T > TMIN) ? FA_MULTI_T : TMIN);
    static constexpr int BF = THREADS * R / NF;
    static constexpr int P0 = BF << (STAGES - 1);
#ifndef FA_MULTI_DB
#define FA_MULTI_DB 2
#endif
#ifndef FA_LEAFMULTI_DB
#define FA_LEAFMULTI_DB 0
#endif
    // buffering mode (nft_kernels.h, MultiStage): 0 single buffer, 1 double buffer, 2 pair-interleaved transforms
    static constexpr int DB = FA_MULTI_DB;
    static constexpr int DB_LEAF = FA_LEAFMULTI_DB;
};
template <int N0, int STAGES> struct KMulti {
    using Params = TreeLevel;
    using C = MultiCfg<N0, STAGES>;
    static constexpr int THREADS = C::THREADS;
    static constexpr int MIN_WAVES = 2;
    static constexpr size_t lds_bytes()
    {
        constexpr size_t tw = (N0 * ((1 << STAGES) - 1) <= 1024) ? (size_t)N0 * ((1 << STAGES) - 1) : 0;
        return ((size_t)((C::DB != 0) ? 2 : 1) * C::THREADS * C::R + (size_t)4 * C::P0 + tw) * sizeof(cplx)
               + (size_t)((C::BF + 1) & ~1) * sizeof(unsigned long long);
    }
    static FA_DEV void body(const Params &p) { body_multi_fft<N0, STAGES, C::R, C::BF, C::DB>(p); }
};
// leaf + first multi-level launch (N0 = 2*DEG*SPT = 16 for the three leaf configurations with d = 8)
template <int DEG, int STAGES> struct KLeafMulti {
    using Params = LeafMultiParams;
    static constexpr int SPT = LeafCfg<DEG>::SPT;
    using C = MultiCfg<2 * DEG * SPT, STAGES>;
    static constexpr int THREADS = C::THREADS;
    static constexpr int MIN_WAVES = 2;
    static constexpr size_t lds_bytes()
    {
        constexpr size_t tw = (size_t)(2 * DEG * SPT) * ((1 << STAGES) - 1);
        return ((size_t)((C::DB_LEAF != 0) ? 2 : 1) * C::THREADS * C::R + (size_t)4 * C::P0 + tw + (size_t)2 * C::THREADS) * sizeof(cplx)
               + (size_t)((C::BF + 1) & ~1) * sizeof(unsigned long long) + (size_t)C::THREADS * sizeof(int);
    }
    static FA_DEV void body(const Params &p) { body_leaf_multi<DEG, SPT, STAGES, C::R, C::BF, C::DB_LEAF>(p); }
};
template <class BE> bool dispatch_leaf_multi(BE &be, const LeafMultiParams &Q, int stages)
{
    const int deg = Q.lp.c.deg;
#define X(dg, st) if (deg == dg && stages == st && 2 * dg * LeafCfg<dg>::SPT == 16) { be.template run<KLeafMulti<dg, st>>((Q.L.n_in + MultiCfg<16, st>::THREADS - 1) / MultiCfg<16, st>::THREADS, 1, Q); return true; }
    X(1, 3) X(2, 3) X(4, 3) X(1, 2) X(2, 2) X(4, 2)
#undef X
    return false;
}
// N: transform length of the first of the `stages` levels
template <class BE> bool dispatch_multi(BE &be, const TreeLevel &L, int N, int stages)
{
#define X(n0, st) if (N == n0 && stages == st) { be.template run<KMulti<n0, st>>((L.n_in + 2 * MultiCfg<n0, st>::P0 - 1) / (2 * MultiCfg<n0, st>::P0), 1, L); return true; }
    X(16, 3) X(32, 3) X(64, 3) X(128, 3) X(256, 3) X(512, 3) X(1024, 3)
    X(16, 2) X(32, 2) X(64, 2) X(128, 2) X(256, 2) X(512, 2) X(1024, 2) X(2048, 2)
#undef X
    return false;
}
// symmetric row kernel with back-to-back loads and pair-interleaved transforms (body_mid_sym)
template <bool DIRECT> struct KMidSym {
    using Params = BigLevel;
    static constexpr int R = FA_MID_R;
    static constexpr int THREADS = kRowTree / R;
    static constexpr int MIN_WAVES = (R == 4) ? 4 : (R == 16 ? 1 : 2);
    static constexpr size_t lds_bytes() { return (size_t)2 * kRowTree * sizeof(cplx); }
    static FA_DEV void body(const Params &p) { body_mid_sym<kRowTree, R, DIRECT>(p); }
};
#ifndef FA_MID_SYM
#define FA_MID_SYM 1   // 0: the generic-IO row kernel (KMid<2>) for the symmetric form as well
#endif
#ifndef FA_MID_GEN
#define FA_MID_GEN 1   // 0: the generic-IO row kernel KMid<4> for the general form
#endif
template <class BE> void run_mid_gen(BE &be, int g, const BigLevel &G);
template <class BE> void run_mid(BE &be, const BigLevel &G)
{
    const int g = (G.L.n_in / 2) * G.N1;
    if (G.L.ne == 4) {
        if constexpr (FA_MID_GEN) run_mid_gen(be, g, G);
        else be.template run<KMid<4>>(g, 1, G);
    }
    else if constexpr (!FA_MID_SYM) be.template run<KMid<2>>(g, 1, G);
    else if (G.y_direct) be.template run<KMidSym<true>>(g, 1, G);
    else be.template run<KMidSym<false>>(g, 1, G);
}
template <class BE> bool dispatch_pair_fft(BE &be, const TreeLevel &L, int N)
{
    switch (N) {
    case 8: run_pair_fft<BE, 8>(be, L); return true;
    case 16: run_pair_fft<BE, 16>(be, L); return true;
    case 32: run_pair_fft<BE, 32>(be, L); return true;
    case 64: run_pair_fft<BE, 64>(be, L); return true;
    case 128: run_pair_fft<BE, 128>(be, L); return true;
    case 256: run_pair_fft<BE, 256>(be, L); return true;
    case 512: run_pair_fft<BE, 512>(be, L); return true;
    case 1024: run_pair_fft<BE, 1024>(be, L); return true;
    case 2048: run_pair_fft<BE, 2048>(be, L); return true;
    case 4096: run_pair_fft<BE, 4096>(be, L); return true;
    default: return false;
    }
}

#define FA_FOR_EACH_N1(X) X(2) X(4) X(8) X(16) X(32) X(64) X(128) X(256) X(512) X(1024) X(2048) X(4096) X(8192)
#define FA_FOR_EACH_CHIRP_N1(X) X(2) X(4) X(8) X(16) X(32) X(64) X(128) X(256) X(512) X(1024) X(2048) X(4096) X(8192)

template <class BE> bool dispatch_col_fwd(BE &be, const BigLevel &G)
{
    const int polys = G.L.ne * G.L.n_in;
    switch (G.N1) {
#define X(n1) case n1: be.template run<KColFwd<n1>>(G.N2 / ColCfg<n1>::BC, polys, G); return true;
        FA_FOR_EACH_N1(X)
#undef X
    default: return false;
    }
}
template <class BE> bool dispatch_col_inv(BE &be, const BigLevel &G)
{
    const int polys = G.L.ne * (G.L.n_in / 2);
    switch (G.N1) {
#define X(n1) case n1: be.template run<KColInv<n1>>(G.N2 / ColCfg<n1>::BC, polys, G); return true;
        FA_FOR_EACH_N1(X)
#undef X
    default: return false;
    }
}
#define FA_FOR_EACH_BRIDGE_N1(X) X(2) X(4) X(8) X(16) X(32) X(64) X(128) X(256) X(512)
template <class BE> bool dispatch_col_bridge(BE &be, const BigLevel &G)
{
    const int polys = G.L.ne * (G.L.n_in / 2);
    switch (G.N1) {
#define X(n1) case n1: be.template run<KColBridge<n1>>(G.N2 / BridgeCfg<n1>::BC, polys, G); return true;
        FA_FOR_EACH_BRIDGE_N1(X)
#undef X
    default: return false;
    }
}
template <class BE> bool dispatch_col_bridge2(BE &be, const BigLevel &G)
{
    const int polys = G.L.ne * (G.L.n_in / 2);
    switch (G.N1) {
#define X(n1) case n1: be.template run<KColBridge2<n1>>(G.N2 / ColCfg<n1>::BC, polys, G); return true;
        FA_FOR_EACH_BRIDGE_N1(X) X(1024) X(2048) X(4096)   // doubling transforms N1 points, not 2*N1
#undef X
    default: return false;
    }
}
template <class BE> bool dispatch_chirp_col_fwd(BE &be, const ChirpParams &C)
{
    const int jobs = C.batch * C.npoly + (C.v_mode == 2 ? 0 : 1);   // the filter job is skipped when its spectrum is cached
    switch (C.N1) {
#define X(n1) case n1: if (C.dft_len > 0) be.template run<KChirpColFwd<n1, true>>(C.N2 / ChirpColCfg<n1>::BC, jobs, C); else be.template run<KChirpColFwd<n1>>(C.N2 / ChirpColCfg<n1>::BC, jobs, C); return true;
        FA_FOR_EACH_CHIRP_N1(X)
#undef X
    default: return false;
    }
}
template <class BE> bool dispatch_chirp_col_inv(BE &be, const ChirpParams &C)
{
    switch (C.N1) {
#define X(n1) case n1: if (C.dft_len > 0) be.template run<KChirpColInv<n1, true>>(C.N2 / ChirpColCfg<n1>::BC, C.batch, C); else if (C.cstype == 10) be.template run<KChirpColInv<n1, false, true>>(C.N2 / ChirpColCfg<n1>::BC, C.batch, C); else be.template run<KChirpColInv<n1>>(C.N2 / ChirpColCfg<n1>::BC, C.batch, C); return true;
        FA_FOR_EACH_CHIRP_N1(X)
#undef X
    default: return false;
    }
}

// ---- real-coefficient path (nft_real.h) -----------------------------------------------------------
constexpr int kRealFusedMaxM = 2048;   // largest folded transform done by one workgroup
template <int DEG> struct KRPairSchool {
    using Params = TreeLevel;
    static constexpr int THREADS = 128;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_rpair_school<DEG>(p); }
};
#ifndef FA_RPAIR_T
#define FA_RPAIR_T 256   // lanes of a workgroup that holds several pairs (knob: 64 = one wave, barriers within the wave)
#endif
template <int M> struct RPairCfg {
    static constexpr int R = 4;
    static constexpr int THREADS = (M / R > FA_RPAIR_T) ? M / R : FA_RPAIR_T;
    static constexpr int B = THREADS / (M / R);
    static constexpr bool DB = true;
};
template <int M> struct KRPair {
    using Params = TreeLevel;
    using C = RPairCfg<M>;
    static constexpr int THREADS = C::THREADS;
    static constexpr int MIN_WAVES = 2;
    static constexpr size_t lds_bytes()
    {
        return ((M > C::R && C::DB) ? (size_t)2 : (size_t)1) * M * C::B * sizeof(cplx)
               + ((M > C::R && M <= 512) ? (size_t)M * sizeof(cplx) : 0) + (size_t)C::B * 8;
    }
    static FA_DEV void body(const Params &p) { body_rpair<M, C::R, C::B, C::DB>(p); }
};
// pair product with the four entries of a factor transformed at once (body_rpair4): M = 32 ... 512
#ifndef FA_RPAIR4
#define FA_RPAIR4 1
#endif
constexpr int kRPair4MinM = 32, kRPair4MaxM = 512;   // (1024 at 8 points per lane: 168 us against 133 for KRPair<1024>)
template <int M> struct RPair4Cfg {
    static constexpr int R = (M >= 1024) ? 8 : 4;
    static constexpr int LANES = 4 * (M / R);                              // of one pair
    static constexpr int THREADS = (LANES > 256) ? LANES : 256;
    static constexpr int BP = THREADS / LANES;
    static constexpr bool TWLDS = (M <= 512);
};
template <int M> struct KRPair4 {
    using Params = TreeLevel;
    using C = RPair4Cfg<M>;
    static constexpr int THREADS = C::THREADS;
    static constexpr int MIN_WAVES = 2;
    static constexpr size_t lds_bytes()
    {
        return ((size_t)2 * M * 4 * C::BP + (C::TWLDS ? (size_t)M : 0)) * sizeof(cplx) + (size_t)((C::BP + 1) & ~1) * 8;
    }
    static FA_DEV void body(const Params &p) { body_rpair4<M, C::R, C::BP, C::TWLDS>(p); }
};
template <int N1> struct KRColFwd {
    using Params = BigLevel;
    using C = ColCfg<N1>;
    static constexpr int THREADS = C::THREADS;
    static constexpr size_t lds_bytes() { return C::lds_bytes(); }
    static FA_DEV void body(const Params &p) { body_rcol_fwd<N1, C::R, C::BC, C::DB>(p); }
};
template <int N1> struct KRColInv {
    using Params = BigLevel;
    using C = ColCfg<N1>;
    static constexpr int THREADS = C::THREADS;
    static constexpr size_t lds_bytes() { return C::lds_bytes(); }
    static FA_DEV void body(const Params &p) { body_rcol_inv<N1, C::R, C::BC, C::DB>(p); }
};
// bridge of the real path: R points of the inverse and 2R of the forward transform per lane; 8 keeps it under 128
// registers (16: 270-280, one wave per SIMD)
template <int N1> struct RBridgeCfg {
    static constexpr int R = (N1 <= 8) ? N1 : 8;
    static constexpr int THREADS = (N1 >= 512) ? 512 : 256;
    static constexpr int BC = THREADS / (N1 / R);
    static constexpr bool DB = false;
    static constexpr size_t lds_bytes() { return (N1 > R) ? (size_t)2 * N1 * BC * sizeof(cplx) : 0; }
};
constexpr int kRBridgeMaxN1 = 1024;
template <int N1> struct KRBridge {
    using Params = BigLevel;
    using C = RBridgeCfg<N1>;
    static constexpr int THREADS = C::THREADS;
    static constexpr int MIN_WAVES = 2;
    static constexpr size_t lds_bytes() { return C::lds_bytes(); }
    static FA_DEV void body(const Params &p) { body_rbridge<N1, C::R, C::BC, C::DB>(p); }
};
template <class BE> bool dispatch_rpair_school(BE &be, const TreeLevel &L)
{
    const int g = (L.n_in / 2 + 127) / 128;
    switch (L.d) {
    case 1: be.template run<KRPairSchool<1>>(g, 1, L); return true;
    case 2: be.template run<KRPairSchool<2>>(g, 1, L); return true;
    case 3: be.template run<KRPairSchool<3>>(g, 1, L); return true;
    default: return false;
    }
}
#define FA_FOR_EACH_RPAIR_M(X) X(4) X(8) X(16) X(32) X(64) X(128) X(256) X(512) X(1024) X(2048)
template <class BE> bool dispatch_rpair(BE &be, const TreeLevel &L, int M)
{
    const int pairs = L.n_in / 2;
    if (FA_RPAIR4 && M >= kRPair4MinM && M <= kRPair4MaxM) {
        switch (M) {
#define X(m) case m: be.template run<KRPair4<m>>((pairs + RPair4Cfg<m>::BP - 1) / RPair4Cfg<m>::BP, 1, L); return true;
            X(32) X(64) X(128) X(256) X(512)
#undef X
        default: break;
        }
    }
    switch (M) {
#define X(m) case m: be.template run<KRPair<m>>((pairs + RPairCfg<m>::B - 1) / RPairCfg<m>::B, 1, L); return true;
        FA_FOR_EACH_RPAIR_M(X)
#undef X
    default: return false;
    }
}
#define FA_FOR_EACH_RCOL_N1(X) X(2) X(4) X(8) X(16) X(32) X(64) X(128) X(256) X(512) X(1024) X(2048) X(4096)
template <class BE> bool dispatch_rcol_fwd(BE &be, const BigLevel &G)
{
    const int polys = 4 * G.L.n_in;
    switch (G.N1) {
#define X(n1) case n1: be.template run<KRColFwd<n1>>(G.N2 / ColCfg<n1>::BC, polys, G); return true;
        FA_FOR_EACH_RCOL_N1(X)
#undef X
    default: return false;
    }
}
template <class BE> bool dispatch_rcol_inv(BE &be, const BigLevel &G)
{
    const int polys = 4 * (G.L.n_in / 2);
    switch (G.N1) {
#define X(n1) case n1: be.template run<KRColInv<n1>>(G.N2 / ColCfg<n1>::BC, polys, G); return true;
        FA_FOR_EACH_RCOL_N1(X)
#undef X
    default: return false;
    }
}
template <class BE> bool dispatch_rbridge(BE &be, const BigLevel &G)
{
    const int polys = 4 * (G.L.n_in / 2);
    switch (G.N1) {
#define X(n1) case n1: be.template run<KRBridge<n1>>(G.N2 / RBridgeCfg<n1>::BC, polys, G); return true;
        FA_FOR_EACH_BRIDGE_N1(X) X(1024)
#undef X
    default: return false;
    }
}
struct KRealCheck {
    using Params = RealCheckParams;
    static constexpr int THREADS = 256;
    static constexpr size_t lds_bytes() { return 0; }
    static FA_DEV void body(const Params &p) { body_real_check(p); }
};
template <int ORDER, bool BFIRST> struct KRCoeffsStrang {
    using Params = CoeffParams;
    static constexpr int THREADS = 64;
    static constexpr size_t lds_bytes() { return (size_t)64 * RStrangCfg<ORDER, BFIRST>::DEG * sizeof(double); }
    static FA_DEV void body(const Params &p) { body_rcoeffs_strang<ORDER, BFIRST>(p); }
};
// akns_disc 13 (2SPLIT6A), 14 (6B), 17 (8A), 18 (8B)
template <class BE> bool dispatch_rcoeffs_strang(BE &be, const CoeffParams &p)
{
    const long long n = (long long)p.batch * p.Dpad;
    const int g = (int)((n + 63) / 64);
    switch (p.disc) {
    case 13: be.template run<KRCoeffsStrang<6, false>>(g, 1, p); return true;
    case 14: be.template run<KRCoeffsStrang<6, true>>(g, 1, p); return true;
    case 17: be.template run<KRCoeffsStrang<8, false>>(g, 1, p); return true;
    case 18: be.template run<KRCoeffsStrang<8, true>>(g, 1, p); return true;
    default: return false;
    }
}
// general 4-entry row kernel with back-to-back loads and pair-interleaved transforms (body_mid_gen, nft_real.h)
#ifndef FA_MIDGEN_R
#define FA_MIDGEN_R 4
#endif
// rows of 1024 points (256 lanes, two workgroups per CU) beat rows of 2048 (512 lanes, one per CU): cfg 5 row kernel
// 147 -> 111 us; 2048 only where the column transform would otherwise exceed the instantiated lengths
#ifndef FA_ROW_GEN
#define FA_ROW_GEN 1024
#endif
constexpr int kRowGen = FA_ROW_GEN;
constexpr int kRowGenMaxN1 = 4096;
inline int row_len_gen(size_t len) { return (len / (size_t)kRowGen <= (size_t)kRowGenMaxN1) ? kRowGen : kRowTree; }
template <int N2> struct KMidGen {
    using Params = BigLevel;
    static constexpr int R = FA_MIDGEN_R;
    static constexpr int THREADS = N2 / R;
    static constexpr int MIN_WAVES = 2;
    static constexpr size_t lds_bytes() { return (size_t)2 * N2 * sizeof(cplx); }
    static FA_DEV void body(const Params &p) { body_mid_gen<N2, R>(p); }
};
template <class BE> void run_mid_gen(BE &be, int g, const BigLevel &G)
{
    if (G.N2 == kRowGen) be.template run<KMidGen<kRowGen>>(g, 1, G);
    else be.template run<KMidGen<kRowTree>>(g, 1, G);
}

// ---- column kernels of length N1 = 3*K (nft_real.h) --------------------------------------------------
template <int K> struct R3Cfg {
    static constexpr int R = (K < 4) ? K : 4;
    static constexpr int THREADS = (K >= 256) ? 512 : 256;
    static constexpr int BC = THREADS / (K / R);
    static constexpr size_t lds_col() { return (K > R) ? (size_t)K * BC * sizeof(cplx) : 0; }
    static constexpr size_t lds_bridge() { return (size_t)2 * K * BC * sizeof(cplx); }   // 2K points, 2R per lane: always > 2R? (K >= R)
};
// the long stand-alone column passes (first forward / last inverse pass of a tree) take more points per lane: fewer
// lanes per column = more columns per workgroup = wider contiguous runs in memory (K = 512 at R = 4: 32-byte stores)
#ifndef FA_R3COL_R_BIG
#define FA_R3COL_R_BIG 8
#endif
template <int K> struct R3ColCfg {
    static constexpr int R = (K >= 128) ? FA_R3COL_R_BIG : R3Cfg<K>::R;
    static constexpr int THREADS = R3Cfg<K>::THREADS;
    static constexpr int BC = THREADS / (K / R);
    static constexpr size_t lds_col() { return (K > R) ? (size_t)K * BC * sizeof(cplx) : 0; }
};
template <int K> struct KR3ColFwd {
    using Params = BigLevel;
    using C = R3ColCfg<K>;
    static constexpr int THREADS = C::THREADS;
    static constexpr int MIN_WAVES = 2;
    static constexpr size_t lds_bytes() { return C::lds_col(); }
    static FA_DEV void body(const Params &p) { body_r3col_fwd<K, C::R, C::BC>(p); }
};
template <int K> struct KR3ColInv {
    using Params = BigLevel;
    using C = R3ColCfg<K>;
    static constexpr int THREADS = C::THREADS;
    static constexpr int MIN_WAVES = 2;
    static constexpr size_t lds_bytes() { return C::lds_col(); }
    static FA_DEV void body(const Params &p) { body_r3col_inv<K, C::R, C::BC>(p); }
};
template <int K> struct KR3Bridge {
    using Params = BigLevel;
    using C = R3Cfg<K>;
    static constexpr int THREADS = C::THREADS;
    static constexpr int MIN_WAVES = 2;
    static constexpr size_t lds_bytes() { return (K > C::R) ? C::lds_bridge() : 0; }
    static FA_DEV void body(const Params &p) { body_r3bridge<K, C::R, C::BC>(p); }
};
constexpr int kR3MaxK = 512;         // column length up to 1536
constexpr int kR3BridgeMaxK = 256;   // bridges up to 768 -> 1536
#define FA_FOR_EACH_R3_K(X) X(1) X(2) X(4) X(8) X(16) X(32) X(64) X(128) X(256)
template <class BE> bool dispatch_r3col_fwd(BE &be, const BigLevel &G)
{
    const int polys = 4 * G.L.n_in;
    switch (G.N1 / 3) {
#define X(k) case k: be.template run<KR3ColFwd<k>>(G.N2 / R3ColCfg<k>::BC, polys, G); return true;
        FA_FOR_EACH_R3_K(X) X(512)
#undef X
    default: return false;
    }
}
template <class BE> bool dispatch_r3col_inv(BE &be, const BigLevel &G)
{
    const int polys = 4 * (G.L.n_in / 2);
    switch (G.N1 / 3) {
#define X(k) case k: be.template run<KR3ColInv<k>>(G.N2 / R3ColCfg<k>::BC, polys, G); return true;
        FA_FOR_EACH_R3_K(X) X(512)
#undef X
    default: return false;
    }
}
template <class BE> bool dispatch_r3bridge(BE &be, const BigLevel &G)
{
    const int polys = 4 * (G.L.n_in / 2);
    switch (G.N1 / 3) {
#define X(k) case k: be.template run<KR3Bridge<k>>(G.N2 / R3Cfg<k>::BC, polys, G); return true;
        FA_FOR_EACH_R3_K(X)
#undef X
    default: return false;
    }
}
#ifndef FA_RLEAF_DL
#define FA_RLEAF_DL 96   // degree of the leaf matrices of the real even-order schemes (96: one more level without transforms)
#endif
template <int ORDER, bool BFIRST> constexpr int rleaf_dl()
{   // 2SPLIT8A (degree 24): 48, the two rows of 97 coefficients plus a 100-entry step matrix do not fit the registers
    return (RStrangCfg<ORDER, BFIRST>::DEG >= 24 && FA_RLEAF_DL > 48) ? 48 : FA_RLEAF_DL;
}
template <int ORDER, bool BFIRST> struct KRLeafStrang {
    using Params = LeafParams;
    static constexpr int DL = rleaf_dl<ORDER, BFIRST>();
    static constexpr int THREADS = 128;
    static constexpr int MIN_WAVES = 1;
    static constexpr size_t lds_bytes()
    {
        constexpr size_t st = (size_t)THREADS * 2 * 16, us = (size_t)(THREADS / 2) * 8 * (RStrangCfg<ORDER, BFIRST>::DEG + 1);
        return ((size_t)THREADS + (st > us ? st : us)) * sizeof(double);
    }
    static FA_DEV void body(const Params &p) { body_rleaf_strang<ORDER, BFIRST, DL, THREADS>(p); }
};
inline int rleaf_strang_samples(int akns_disc)
{
    switch (akns_disc) {
    case 13: return rleaf_dl<6, false>() / RStrangCfg<6, false>::DEG;
    case 14: return rleaf_dl<6, true>() / RStrangCfg<6, true>::DEG;
    case 17: return rleaf_dl<8, false>() / RStrangCfg<8, false>::DEG;
    case 18: return rleaf_dl<8, true>() / RStrangCfg<8, true>::DEG;
    default: return 0;
    }
}
template <class BE> bool dispatch_rleaf_strang(BE &be, const LeafParams &lp)
{
    const CoeffParams &p = lp.c;
    const long long n = (long long)p.batch * (p.Dpad / lp.spt);
    const int g = (int)((2 * n + 127) / 128);
    switch (p.disc) {
    case 13: be.template run<KRLeafStrang<6, false>>(g, 1, lp); return true;
    case 14: be.template run<KRLeafStrang<6, true>>(g, 1, lp); return true;
    case 17: be.template run<KRLeafStrang<8, false>>(g, 1, lp); return true;
    case 18: be.template run<KRLeafStrang<8, true>>(g, 1, lp); return true;
    default: return false;
    }
}
