// fft_dev.h -- fp64 complex arithmetic, in-register DFTs and the workgroup-cooperative Stockham
// FFT every kernel of the product tree and of the chirp z-transform is built from.
//
// Replaces the reference's KissFFT back end (src/3rd_party/kiss_fft/kiss_fft.c:237-302 via
// include/private/fnft__fft_wrapper.h:124-137).  Semantics are the same (unnormalised, sign -1
// forward / +1 inverse); lengths are powers of two (any length >= the linear convolution length
// gives the same polynomial product).
//
// Design (CDNA4): each lane keeps R points in VGPRs, a radix-R butterfly is pure register work,
// and lanes exchange points through LDS once per pass (Stockham autosort, so every LDS read is
// lane-contiguous 16-byte ds_read_b128).  B independent transforms are interleaved with the
// batch index fastest, which keeps LDS accesses conflict-free for the small transforms of the
// lower tree levels.
#pragma once
#include "dev_compat.h"

struct __attribute__((aligned(16))) cplx {
    double x, y;
};

FA_HD cplx cmake(double x, double y) { cplx r; r.x = x; r.y = y; return r; }
FA_HD cplx operator+(cplx a, cplx b) { return cmake(a.x + b.x, a.y + b.y); }
FA_HD cplx operator-(cplx a, cplx b) { return cmake(a.x - b.x, a.y - b.y); }
FA_HD cplx operator*(cplx a, cplx b)
{
    return cmake(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
FA_HD cplx operator*(cplx a, double s) { return cmake(a.x * s, a.y * s); }
FA_HD cplx cconj(cplx a) { return cmake(a.x, -a.y); }
// c + a*b
FA_HD cplx cfma(cplx a, cplx b, cplx c)
{
    return cmake(fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y)));
}
FA_HD double cnorm2(cplx a) { return fma(a.x, a.x, a.y * a.y); }

// a * (SIGN * i)
template <int SIGN> FA_HD cplx mul_si(cplx a)
{
    return SIGN > 0 ? cmake(-a.y, a.x) : cmake(a.y, -a.x);
}
// conj for the inverse direction: tables hold exp(-2 pi i j / n)
template <int SIGN> FA_HD cplx tw_dir(cplx w) { return SIGN > 0 ? cconj(w) : w; }

// ---------------------------------------------------------------------------------------------
// In-register DFTs, natural order in and out:  X[k] = sum_j x[j] exp(SIGN*2*pi*i*j*k/R)
// ---------------------------------------------------------------------------------------------
template <int R, int SIGN> struct RegDft;

template <int SIGN> struct RegDft<1, SIGN> {
    static FA_HD void run(cplx (&)[1]) {}
};

template <int SIGN> struct RegDft<2, SIGN> {
    static FA_HD void run(cplx (&x)[2])
    {
        cplx a = x[0], b = x[1];
        x[0] = a + b;
        x[1] = a - b;
    }
};

template <int SIGN> struct RegDft<4, SIGN> {
    static FA_HD void run(cplx (&x)[4])
    {
        cplx s0 = x[0] + x[2], d0 = x[0] - x[2];
        cplx s1 = x[1] + x[3], d1 = mul_si<SIGN>(x[1] - x[3]);
        x[0] = s0 + s1;
        x[1] = d0 + d1;
        x[2] = s0 - s1;
        x[3] = d0 - d1;
    }
};

template <int SIGN> struct RegDft<8, SIGN> {
    static FA_HD void run(cplx (&x)[8])
    {
        const double h = 0.70710678118654752440;
        cplx e[4], o[4];
        for (int k = 0; k < 4; k++) e[k] = x[k] + x[k + 4];
        cplx t0 = x[0] - x[4], t1 = x[1] - x[5], t2 = x[2] - x[6], t3 = x[3] - x[7];
        const double S = (double)SIGN;
        o[0] = t0;
        o[1] = cmake(h * (t1.x - S * t1.y), h * (S * t1.x + t1.y));   // * w8^1 = h(1 + S i)
        o[2] = mul_si<SIGN>(t2);                                      // * w8^2 = S i
        o[3] = cmake(h * (-t3.x - S * t3.y), h * (S * t3.x - t3.y));  // * w8^3 = h(-1 + S i)
        RegDft<4, SIGN>::run(e);
        RegDft<4, SIGN>::run(o);
        for (int k = 0; k < 4; k++) {
            x[2 * k] = e[k];
            x[2 * k + 1] = o[k];
        }
    }
};

template <int SIGN> struct RegDft<16, SIGN> {
    static FA_HD void run(cplx (&x)[16])
    {
        // w16^k = cos(k pi/8) + S i sin(k pi/8)
        const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173;
        const double h = 0.70710678118654752440;
        const double S = (double)SIGN;
        cplx e[8], o[8];
        for (int k = 0; k < 8; k++) {
            e[k] = x[k] + x[k + 8];
            o[k] = x[k] - x[k + 8];
        }
        o[1] = o[1] * cmake(c1, S * s1);
        o[2] = o[2] * cmake(h, S * h);
        o[3] = o[3] * cmake(s1, S * c1);
        o[4] = mul_si<SIGN>(o[4]);
        o[5] = o[5] * cmake(-s1, S * c1);
        o[6] = o[6] * cmake(-h, S * h);
        o[7] = o[7] * cmake(-c1, S * s1);
        RegDft<8, SIGN>::run(e);
        RegDft<8, SIGN>::run(o);
        for (int k = 0; k < 8; k++) {
            x[2 * k] = e[k];
            x[2 * k + 1] = o[k];
        }
    }
};

template <int SIGN> struct RegDft<32, SIGN> {
    static FA_HD void run(cplx (&x)[32])
    {
        // w32^k = cos(k pi/16) + S i sin(k pi/16), k = 0..15
        const double cs[16] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                               0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173,
                               0.19509032201612826785, 0.0, -0.19509032201612826785, -0.38268343236508977173,
                               -0.55557023301960222474, -0.70710678118654752440, -0.83146961230254523708,
                               -0.92387953251128675613, -0.98078528040323044913};
        const double sn[16] = {0.0, 0.19509032201612826785, 0.38268343236508977173, 0.55557023301960222474,
                               0.70710678118654752440, 0.83146961230254523708, 0.92387953251128675613,
                               0.98078528040323044913, 1.0, 0.98078528040323044913, 0.92387953251128675613,
                               0.83146961230254523708, 0.70710678118654752440, 0.55557023301960222474,
                               0.38268343236508977173, 0.19509032201612826785};
        const double S = (double)SIGN;
        cplx e[16], o[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            e[k] = x[k] + x[k + 16];
            o[k] = x[k] - x[k + 16];
        }
#pragma unroll
        for (int k = 1; k < 16; k++) o[k] = (k == 8) ? mul_si<SIGN>(o[k]) : o[k] * cmake(cs[k], S * sn[k]);
        RegDft<16, SIGN>::run(e);
        RegDft<16, SIGN>::run(o);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            x[2 * k] = e[k];
            x[2 * k + 1] = o[k];
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Workgroup-cooperative FFT of B interleaved length-N sequences, R points per lane.
//
// Lane (v, c), v < N/R, c < B, owns the elements  v + (N/R)*i,  i < R  of sequence c, both on
// entry and on exit (natural order).  Stockham decimation-in-frequency pass of radix r on
// n_cur-point sub-problems with s = N/n_cur finished sub-transforms:
//     y[q + s*(r*p + k)] = w_{n_cur}^{p k} * DFT_r( x[q + s*(p + m*j)] )_k,   m = n_cur/r,
// for butterfly u = q + s*p; its inputs are x[u + (N/r)*j] -- registers j' + (R/r)*j of the lane
// that owns u -- so only the outputs cross lanes (through LDS).  The last pass (n_cur == r)
// writes to the positions it read, so its results stay in registers.
//
// tw: table of exp(-2*pi*i*j/N), j < N (forward); the inverse direction conjugates.
// lds: DB = true: 2 buffers of N*B elements each; `parity` selects the next buffer to write and
// must be carried across consecutive calls (one barrier per exchange is then sufficient).
// DB = false: one buffer, two barriers per exchange.
// ---------------------------------------------------------------------------------------------
// LDS element index swizzle: the first pass of a transform writes element 8p+k from lane p, a
// 128-byte lane stride that would hit the same banks 8 ways; XOR-ing the low three bits with the
// next three spreads those writes over all banks and keeps every aligned group of 8 consecutive
// elements a permutation of itself, so the lane-contiguous reads stay conflict-free.
template <int N> FA_HD int lds_swz(int a) { return (N >= 64) ? (a ^ ((a >> 3) & 7)) : a; }

template <int N, int R, int B, int SIGN, bool DB, int NCUR, int S, bool TWC = false> struct FftPass {
    static constexpr int r = (NCUR < R) ? NCUR : R;
    static constexpr int J = R / r;     // butterflies per lane
    static constexpr int m = NCUR / r;
    static constexpr bool last = (NCUR == r);

    static FA_DEV void run(cplx (&x)[R], cplx *lds, int v, int c, const cplx *__restrict__ tw,
                           int &parity)
    {
        cplx *buf = DB ? lds + (size_t)parity * (size_t)(N * B) : lds;
        if constexpr (!last && !DB) FA_SYNC_LDS();  // single buffer: previous readers must be done
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int u = v + (N / R) * j;
            const int q = u % S;
            const int p = u / S;
            cplx t[r];
#pragma unroll
            for (int k = 0; k < r; k++) t[k] = x[j + J * k];
            RegDft<r, SIGN>::run(t);
            if constexpr (!last) {
                if constexpr (TWC && r == 8) {
                    // three table reads (w, w^2, w^4), the other powers by multiplication: the tables of
                    // the long transforms live in L2, and the level kernels have VALU cycles to spare
                    const cplx w1 = tw_dir<SIGN>(tw[(size_t)p * S]);
                    const cplx w2 = tw_dir<SIGN>(tw[(size_t)(2 * p) * S]);
                    const cplx w4 = tw_dir<SIGN>(tw[(size_t)(4 * p) * S]);
                    // w^5 = w^4 w etc. applied as two factors: same 11 products, fewer live registers
                    t[4] = t[4] * w4; t[5] = t[5] * w4; t[6] = t[6] * w4; t[7] = t[7] * w4;
                    t[1] = t[1] * w1; t[5] = t[5] * w1;
                    t[2] = t[2] * w2; t[6] = t[6] * w2;
                    const cplx w3 = w1 * w2;
                    t[3] = t[3] * w3; t[7] = t[7] * w3;
                } else if constexpr (TWC && r == 16) {
                    const cplx w1 = tw_dir<SIGN>(tw[(size_t)p * S]);
                    const cplx w2 = tw_dir<SIGN>(tw[(size_t)(2 * p) * S]);
                    const cplx w4 = tw_dir<SIGN>(tw[(size_t)(4 * p) * S]);
                    const cplx w8 = tw_dir<SIGN>(tw[(size_t)(8 * p) * S]);
                    const cplx w3 = w1 * w2, w5 = w1 * w4, w6 = w2 * w4, w7 = w3 * w4;
                    t[1] = t[1] * w1;  t[2] = t[2] * w2;  t[3] = t[3] * w3;  t[4] = t[4] * w4;
                    t[5] = t[5] * w5;  t[6] = t[6] * w6;  t[7] = t[7] * w7;  t[8] = t[8] * w8;
                    t[9] = t[9] * (w1 * w8);   t[10] = t[10] * (w2 * w8);  t[11] = t[11] * (w3 * w8);
                    t[12] = t[12] * (w4 * w8); t[13] = t[13] * (w5 * w8);  t[14] = t[14] * (w6 * w8);
                    t[15] = t[15] * (w7 * w8);
                } else if constexpr (TWC && r == 4) {
                    const cplx w1 = tw_dir<SIGN>(tw[(size_t)p * S]);
                    const cplx w2 = tw_dir<SIGN>(tw[(size_t)(2 * p) * S]);
                    t[1] = t[1] * w1;  t[2] = t[2] * w2;  t[3] = t[3] * (w1 * w2);
                } else {
#pragma unroll
                    for (int k = 1; k < r; k++) t[k] = t[k] * tw_dir<SIGN>(tw[(size_t)(p * k) * S]);
                }
#pragma unroll
                for (int k = 0; k < r; k++) buf[(size_t)lds_swz<N>(q + S * (r * p + k)) * B + c] = t[k];
            } else {
#pragma unroll
                for (int k = 0; k < r; k++) x[j + J * k] = t[k];
            }
        }
        if constexpr (!last) {
            FA_SYNC_LDS();
#pragma unroll
            for (int i = 0; i < R; i++) x[i] = buf[(size_t)lds_swz<N>(v + (N / R) * i) * B + c];
            parity ^= 1;
            FftPass<N, R, B, SIGN, DB, NCUR / r, S * r, TWC>::run(x, lds, v, c, tw, parity);
        }
    }
};

template <int N, int R, int B, int SIGN, bool DB, int S, bool TWC> struct FftPass<N, R, B, SIGN, DB, 1, S, TWC> {
    static FA_DEV void run(cplx (&)[R], cplx *, int, int, const cplx *__restrict__, int &) {}
};

template <int N, int R, int B, int SIGN, bool DB = true, bool TWC = false>
FA_DEV void fft_wg(cplx (&x)[R], cplx *lds, int v, int c, const cplx *__restrict__ tw, int &parity)
{
    static_assert(N >= R, "fft_wg: N must be at least R");
    FftPass<N, R, B, SIGN, DB, N, 1, TWC>::run(x, lds, v, c, tw, parity);
}

// ---------------------------------------------------------------------------------------------
// Two transforms at once, software-pipelined half a pass apart ("pair-interleaved").
//
// The butterflies of one sequence are independent of the LDS round trip of the other, so each
// barrier interval holds  [LDS reads of Y issued] [butterflies of X] [LDS writes of X]  (and the
// mirror image in the next interval): the read latency and the barrier wait of one sequence hide
// behind the register work of the other.  With 1-2 waves per SIMD -- all the row kernels can hold
// -- that is the latency hiding the hardware scheduler cannot provide.
//
// lds: two single buffers of N*B elements, bufX = lds, bufY = lds + N*B (the footprint of the
// double-buffered single transform).  Hand-over rule: on entry nobody is still reading either
// buffer from an earlier call's exchange EXCEPT reads that were issued before that call's last
// barrier -- which is exactly what this function leaves behind, so consecutive calls chain without
// extra barriers.  Intervals (p = exchange pass):
//     [B_X(p) W_X(p)] | [R_X(p) B_Y(p) W_Y(p)] | [R_Y(p) B_X(p+1) W_X(p+1)] | ...
// W_X(p+1) follows the barrier that follows R_X(p) of every lane; W_Y(p) follows the barrier that
// follows R_Y(p-1) of every lane.
// ---------------------------------------------------------------------------------------------
// twiddles of one radix-8 pass of a lane (w, w^2, w^4 of its butterfly; the other powers by
// multiplication): both sequences of a pair-interleaved transform use the same set, and the set of the
// NEXT pass is requested one barrier interval ahead, before the caller's hook issues its own loads --
// so that waiting for twiddles never waits for the hook's (younger) loads
struct TwSet8 {
    cplx w1, w2, w4;
};

template <int N, int R, int B, int SIGN, int NCUR, int S, bool TWC> struct Fft2Pass {
    static constexpr int r = (NCUR < R) ? NCUR : R;
    static constexpr int J = R / r;
    static constexpr bool last = (NCUR == r);
    // one shared, prefetched twiddle set per pass: radix-8 passes of one butterfly per lane with the table in L2
    // (radix 4: w and w^2, the third factor by multiplication)
    static constexpr bool kShared = TWC && (r == 8 || r == 4) && J == 1 && !last;

    static FA_DEV TwSet8 load_tw(int v, const cplx *__restrict__ tw)
    {
        TwSet8 t;
        if constexpr (kShared) {
            const int p = v / S;
            t.w1 = tw_dir<SIGN>(tw[(size_t)p * S]);
            t.w2 = tw_dir<SIGN>(tw[(size_t)(2 * p) * S]);
            if constexpr (r == 8) t.w4 = tw_dir<SIGN>(tw[(size_t)(4 * p) * S]);
            else t.w4 = cmake(1.0, 0.0);
        } else {
            t.w1 = t.w2 = t.w4 = cmake(1.0, 0.0);
        }
        return t;
    }

    // radix-r butterflies (and twiddles, unless this is the last pass) of one sequence, in place:
    // butterfly j's output k is left in x[j + J*k]
    static FA_DEV void bfly(cplx (&x)[R], int v, const cplx *__restrict__ tw, const TwSet8 &ts)
    {
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int u = v + (N / R) * j;
            const int p = u / S;
            cplx t[r];
#pragma unroll
            for (int k = 0; k < r; k++) t[k] = x[j + J * k];
            RegDft<r, SIGN>::run(t);
            if constexpr (!last) {
                if constexpr (kShared && r == 4) {
                    t[1] = t[1] * ts.w1;
                    t[2] = t[2] * ts.w2;
                    t[3] = t[3] * (ts.w1 * ts.w2);
                } else if constexpr (kShared) {
                    t[4] = t[4] * ts.w4; t[5] = t[5] * ts.w4; t[6] = t[6] * ts.w4; t[7] = t[7] * ts.w4;
                    t[1] = t[1] * ts.w1; t[5] = t[5] * ts.w1;
                    t[2] = t[2] * ts.w2; t[6] = t[6] * ts.w2;
                    const cplx w3 = ts.w1 * ts.w2;
                    t[3] = t[3] * w3; t[7] = t[7] * w3;
                } else if constexpr (TWC && r == 8) {
                    const cplx w1 = tw_dir<SIGN>(tw[(size_t)p * S]);
                    const cplx w2 = tw_dir<SIGN>(tw[(size_t)(2 * p) * S]);
                    const cplx w4 = tw_dir<SIGN>(tw[(size_t)(4 * p) * S]);
                    t[4] = t[4] * w4; t[5] = t[5] * w4; t[6] = t[6] * w4; t[7] = t[7] * w4;
                    t[1] = t[1] * w1; t[5] = t[5] * w1;
                    t[2] = t[2] * w2; t[6] = t[6] * w2;
                    const cplx w3 = w1 * w2;
                    t[3] = t[3] * w3; t[7] = t[7] * w3;
                } else if constexpr (TWC && r == 16) {
                    const cplx w1 = tw_dir<SIGN>(tw[(size_t)p * S]);
                    const cplx w2 = tw_dir<SIGN>(tw[(size_t)(2 * p) * S]);
                    const cplx w4 = tw_dir<SIGN>(tw[(size_t)(4 * p) * S]);
                    const cplx w8 = tw_dir<SIGN>(tw[(size_t)(8 * p) * S]);
                    const cplx w3 = w1 * w2, w5 = w1 * w4, w6 = w2 * w4, w7 = w3 * w4;
                    t[1] = t[1] * w1;  t[2] = t[2] * w2;  t[3] = t[3] * w3;  t[4] = t[4] * w4;
                    t[5] = t[5] * w5;  t[6] = t[6] * w6;  t[7] = t[7] * w7;  t[8] = t[8] * w8;
                    t[9] = t[9] * (w1 * w8);   t[10] = t[10] * (w2 * w8);  t[11] = t[11] * (w3 * w8);
                    t[12] = t[12] * (w4 * w8); t[13] = t[13] * (w5 * w8);  t[14] = t[14] * (w6 * w8);
                    t[15] = t[15] * (w7 * w8);
                } else {
#pragma unroll
                    for (int k = 1; k < r; k++) t[k] = t[k] * tw_dir<SIGN>(tw[(size_t)(p * k) * S]);
                }
            }
#pragma unroll
            for (int k = 0; k < r; k++) x[j + J * k] = t[k];
        }
    }
    static FA_DEV void write(const cplx (&x)[R], cplx *buf, int v, int c)
    {
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int u = v + (N / R) * j;
            const int q = u % S, p = u / S;
#pragma unroll
            for (int k = 0; k < r; k++) buf[(size_t)lds_swz<N>(q + S * (r * p + k)) * B + c] = x[j + J * k];
        }
    }
    static FA_DEV void read(cplx (&x)[R], const cplx *buf, int v, int c)
    {
#pragma unroll
        for (int i = 0; i < R; i++) x[i] = buf[(size_t)lds_swz<N>(v + (N / R) * i) * B + c];
    }

    // entered with X's pass-p inputs in registers, Y's pass-p inputs READ ISSUED (or, for the first
    // pass, in registers) and this pass's shared twiddle set requested (ts).  hook(k) is called once per
    // barrier interval, k = 0, 1, 2, ... (two per pass): the caller's chance to put independent work --
    // global loads of the NEXT operands -- into the instruction stream of this transform.
    template <class Hook>
    static FA_DEV void run(cplx (&x)[R], cplx (&y)[R], cplx *bufX, cplx *bufY, int v, int c,
                           const cplx *__restrict__ tw, Hook &hook, int k0, const TwSet8 &ts)
    {
        using Next = Fft2Pass<N, R, B, SIGN, NCUR / r, S * r, TWC>;
        bfly(x, v, tw, ts);
        if constexpr (last) {
            hook(k0);
            bfly(y, v, tw, ts);
            hook(k0 + 1);
        } else {
            write(x, bufX, v, c);
            const TwSet8 tn = Next::load_tw(v, tw);   // next pass's set: older than the hook's loads
            hook(k0);
            FA_SYNC_LDS();
            read(x, bufX, v, c);
            bfly(y, v, tw, ts);
            write(y, bufY, v, c);
            hook(k0 + 1);
            FA_SYNC_LDS();
            read(y, bufY, v, c);
            Next::run(x, y, bufX, bufY, v, c, tw, hook, k0 + 2, tn);
        }
    }
};
template <int N, int R, int B, int SIGN, int S, bool TWC> struct Fft2Pass<N, R, B, SIGN, 1, S, TWC> {
    static FA_DEV TwSet8 load_tw(int, const cplx *__restrict__) { return TwSet8{cmake(1.0, 0.0), cmake(1.0, 0.0), cmake(1.0, 0.0)}; }
    template <class Hook>
    static FA_DEV void run(cplx (&)[R], cplx (&)[R], cplx *, cplx *, int, int, const cplx *__restrict__, Hook &, int,
                           const TwSet8 &) {}
};

struct FftNoHook {
    FA_DEV void operator()(int) const {}
};

template <int N, int R, int B, int SIGN, bool TWC = false, class Hook>
FA_DEV void fft_wg2(cplx (&x)[R], cplx (&y)[R], cplx *lds, int v, int c, const cplx *__restrict__ tw, Hook &hook)
{
    static_assert(N >= R, "fft_wg2: N must be at least R");
    using P0 = Fft2Pass<N, R, B, SIGN, N, 1, TWC>;
    const TwSet8 t0 = P0::load_tw(v, tw);
    P0::run(x, y, lds, lds + (size_t)N * B, v, c, tw, hook, 0, t0);
}
template <int N, int R, int B, int SIGN, bool TWC = false>
FA_DEV void fft_wg2(cplx (&x)[R], cplx (&y)[R], cplx *lds, int v, int c, const cplx *__restrict__ tw)
{
    FftNoHook h;
    fft_wg2<N, R, B, SIGN, TWC>(x, y, lds, v, c, tw, h);
}

// ---------------------------------------------------------------------------------------------
// exp(-2*pi*i*j/N) for any power of two N <= NMAX = 2^(2*FINE) from one pair of master tables:
//   J = j * (NMAX/N) = jh*2^FINE + jl,   w = hi[jh] * lo[jl],
//   hi[jh] = exp(-2 pi i jh / 2^FINE),   lo[jl] = exp(-2 pi i jl / NMAX).
// ---------------------------------------------------------------------------------------------
struct BigTwiddle {
    const cplx *hi;
    const cplx *lo;
    int fine_log2;  // FINE
    int shift;      // log2(NMAX/N)
};
FA_DEV cplx big_twiddle(const BigTwiddle &t, unsigned j)
{
    const unsigned J = j << t.shift;
    const cplx a = t.hi[J >> t.fine_log2];
    const cplx b = t.lo[J & ((1u << t.fine_log2) - 1u)];
    return a * b;
}
