// The 1024-point transform of fir_ols32.hip as a reusable core: one transform per HALF-WAVE,
// 1024 = 32 x 32 (see fir_ols32.hip for the decomposition).  Included by fir_ols32.hip (the FIR
// alone) and chain_fused.hip (FIR -> biquad -> gain in one pass).
#pragma once

#include "ols_math.hpp"

namespace pipehip {
namespace ols {

// Ablation builds for the energy table (scripts/build_ablate_lib.sh fir_ols32 PH_OLS_ABLATE ...; never
// the shipped library; the results are WRONG, the instruction streams are what is measured):
// bit 0: the two exchanges through the LDS plane become register copies; bit 1: the twiddle table and
// the tap spectrum are one value read once instead of 94 LDS reads per unit.
#ifndef PH_OLS_ABLATE
#define PH_OLS_ABLATE 0
#endif

constexpr int kM32 = 1024;
constexpr int kHalf32 = 513;          // H[0..512] (+1 pad)
constexpr int kPlane32 = 32 * 33;     // doubles per item exchange plane (stride 33)

// ---- butterflies that absorb a pending factor ------------------------------------------------
// A value may carry a factor it has not been multiplied by yet: quarter turns (free: a swap and
// signs) times a general complex w.  The butterfly that consumes it multiplies and adds in one
// fma chain and takes the difference as 2a - s: 6 instructions where multiply-then-butterfly
// takes 8, 10 instead of 12 when both inputs carry a factor.  The kernel runs at the socket's
// power cap (profiles/r02_clock_power_*.json): its speed is its float64 instruction count.
struct Pend {
    int q;     // multiply by (SIGN < 0 ? -i : +i)^q ...
    bool gen;  // ... and by w
    cd w;
};
__device__ __forceinline__ Pend pend_none() { return Pend{0, false, cd{1.0, 0.0}}; }
__device__ __forceinline__ Pend pend_w(cd w) { return Pend{0, true, w}; }
// W32^e (SIGN < 0: forward, = cos - i sin) or its conjugate, e compile-time after unrolling
template <int SIGN>
__device__ __forceinline__ Pend pend_w32(int e)
{
    constexpr double c[8] = {1.0,
                             0.98078528040323044913,
                             0.92387953251128675613,
                             0.83146961230254523708,
                             0.70710678118654752440,
                             0.55557023301960222474,
                             0.38268343236508977173,
                             0.19509032201612826785};
    constexpr double s[8] = {0.0,
                             0.19509032201612826785,
                             0.38268343236508977173,
                             0.55557023301960222474,
                             0.70710678118654752440,
                             0.83146961230254523708,
                             0.92387953251128675613,
                             0.98078528040323044913};
    e &= 31;
    const int r = e & 7;  // W32^e = (W32^8)^(e / 8) W32^r, W32^8 = -i (forward)
    return Pend{e >> 3, r != 0, cd{c[r], SIGN < 0 ? -s[r] : s[r]}};
}

template <int SIGN>
__device__ __forceinline__ cd rotq(cd v, int q)
{
    q &= 3;
    if (q == 0)
        return v;
    if (q == 2)
        return cd{-v.re, -v.im};
    const bool minus_i = (SIGN < 0) == (q == 1);
    return minus_i ? cd{v.im, -v.re} : cd{-v.im, v.re};
}

__device__ __forceinline__ cd cfma(cd w, cd v, cd acc)  // acc + w v
{
    cd r;
    r.re = __builtin_fma(-w.im, v.im, __builtin_fma(w.re, v.re, acc.re));
    r.im = __builtin_fma(w.re, v.im, __builtin_fma(w.im, v.re, acc.im));
    return r;
}

// s = pa a + pb b, d = pa a - pb b
template <int SIGN>
__device__ __forceinline__ void bf(cd a, Pend pa, cd b, Pend pb, cd &s, cd &d)
{
    a = rotq<SIGN>(a, pa.q);
    b = rotq<SIGN>(b, pb.q);
    if (!pa.gen && !pb.gen) {
        s = cd{a.re + b.re, a.im + b.im};
        d = cd{a.re - b.re, a.im - b.im};
    } else if (!pa.gen) {
        s = cfma(pb.w, b, a);
        d = cd{__builtin_fma(2.0, a.re, -s.re), __builtin_fma(2.0, a.im, -s.im)};
    } else if (!pb.gen) {
        s = cfma(pa.w, a, b);
        d = cd{__builtin_fma(-2.0, b.re, s.re), __builtin_fma(-2.0, b.im, s.im)};
    } else {
        const cd x = cmul(a, pa.w);
        s = cfma(pb.w, b, x);
        d = cd{__builtin_fma(2.0, x.re, -s.re), __builtin_fma(2.0, x.im, -s.im)};
    }
}

// 4-point DFT of (p0 x0, p1 x1, p2 x2, p3 x3), in place
template <int SIGN>
__device__ __forceinline__ void dft4p(cd &x0, Pend p0, cd &x1, Pend p1, cd &x2, Pend p2, cd &x3, Pend p3)
{
    cd s02, d02, s13, d13;
    bf<SIGN>(x0, p0, x2, p2, s02, d02);
    bf<SIGN>(x1, p1, x3, p3, s13, d13);
    const cd j13 = SIGN < 0 ? cd{d13.im, -d13.re} : cd{-d13.im, d13.re};
    x0 = cd{s02.re + s13.re, s02.im + s13.im};
    x2 = cd{s02.re - s13.re, s02.im - s13.im};
    x1 = cd{d02.re + j13.re, d02.im + j13.im};
    x3 = cd{d02.re - j13.re, d02.im - j13.im};
}

// second half of the 16-point DFT (n = j + 4i, k = m + 4p): t[j][m] at v[j + 4m] carries
// W16^(j m) = W32^(2 j m) into the 4-point DFTs over j; then (p, m) -> k = m + 4p
template <int SIGN>
__device__ __forceinline__ void dft16_stage2(cd (&v)[16])
{
#pragma unroll
    for (int m = 0; m < 4; ++m)
        dft4p<SIGN>(v[0 + 4 * m], pend_none(), v[1 + 4 * m], pend_w32<SIGN>(2 * m), v[2 + 4 * m],
                    pend_w32<SIGN>(4 * m), v[3 + 4 * m], pend_w32<SIGN>(6 * m));
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int m = p + 1; m < 4; ++m) {
            const cd t = v[p + 4 * m];
            v[p + 4 * m] = v[m + 4 * p];
            v[m + 4 * p] = t;
        }
}

// 16-point DFT in place of (W32^(e0 + n de) v[n]): the factors are compile-time constants
template <int SIGN>
__device__ __forceinline__ void dft16p(cd (&v)[16], int e0, int de)
{
#pragma unroll
    for (int j = 0; j < 4; ++j)
        dft4p<SIGN>(v[j], pend_w32<SIGN>(e0 + j * de), v[j + 4], pend_w32<SIGN>(e0 + (j + 4) * de), v[j + 8],
                    pend_w32<SIGN>(e0 + (j + 8) * de), v[j + 12], pend_w32<SIGN>(e0 + (j + 12) * de));
    dft16_stage2<SIGN>(v);
}

// 16-point DFT in place of (load(n) v[n]), the factors read from LDS one 4-point DFT ahead of
// their use (all sixteen in flight would cost 64 VGPRs); PLAIN0: v[0] carries no factor
template <int SIGN, bool PLAIN0, class LD>
__device__ __forceinline__ void dft16_rt(cd (&v)[16], LD load)
{
    cd w[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (!(PLAIN0 && i == 0))
            w[0][i] = load(4 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j + 1 < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w[(j + 1) & 1][i] = load(j + 1 + 4 * i);
        }
        const Pend p0 = (PLAIN0 && j == 0) ? pend_none() : pend_w(w[j & 1][0]);
        dft4p<SIGN>(v[j], p0, v[j + 4], pend_w(w[j & 1][1]), v[j + 8], pend_w(w[j & 1][2]), v[j + 12],
                    pend_w(w[j & 1][3]));
        __builtin_amdgcn_sched_barrier(0);
    }
    dft16_stage2<SIGN>(v);
}

// 32-point DFT, decimation in frequency: in x[n] = lo[n], x[16 + n] = hi[n];
// out X[2m] = lo[m], X[2m + 1] = hi[m]
template <int SIGN>
__device__ __forceinline__ void dft32_dif(cd (&lo)[16], cd (&hi)[16])
{
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const cd a{lo[n].re + hi[n].re, lo[n].im + hi[n].im};
        const cd b{lo[n].re - hi[n].re, lo[n].im - hi[n].im};
        lo[n] = a;
        hi[n] = b;  // W32^n pending
    }
    dft16p<SIGN>(lo, 0, 0);
    dft16p<SIGN>(hi, 0, 1);
}

// the same of (load(k) x[k]), k = 1..31 (x[0] plain): the factors from LDS, two butterflies ahead
template <int SIGN, class LD>
__device__ __forceinline__ void dft32_dif_rt(cd (&lo)[16], cd (&hi)[16], LD load)
{
    constexpr int G = 2;
    cd w[2][2 * G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        if (j > 0)
            w[0][2 * j] = load(j);
        w[0][2 * j + 1] = load(16 + j);
    }
#pragma unroll
    for (int g = 0; g < 16 / G; ++g) {
        if (g + 1 < 16 / G) {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                w[(g + 1) & 1][2 * j] = load(G * (g + 1) + j);
                w[(g + 1) & 1][2 * j + 1] = load(16 + G * (g + 1) + j);
            }
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int n = G * g + j;
            cd a, b;
            bf<SIGN>(lo[n], n == 0 ? pend_none() : pend_w(w[g & 1][2 * j]), hi[n], pend_w(w[g & 1][2 * j + 1]), a, b);
            lo[n] = a;
            hi[n] = b;  // W32^n pending
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    dft16p<SIGN>(lo, 0, 0);
    dft16p<SIGN>(hi, 0, 1);
}

// 32-point DFT, decimation in time, of (load(k) X[k]): in X[2m] = lo[m], X[2m + 1] = hi[m];
// out x[n] = lo[n], x[16 + n] = hi[n].  PLAIN0: X[0] carries no factor.
template <int SIGN, bool PLAIN0, class LD>
__device__ __forceinline__ void dft32_dit_rt(cd (&lo)[16], cd (&hi)[16], LD load)
{
    dft16_rt<SIGN, PLAIN0>(lo, [&](int m) { return load(2 * m); });
    dft16_rt<SIGN, false>(hi, [&](int m) { return load(2 * m + 1); });
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        cd a, b;
        bf<SIGN>(lo[n], pend_none(), hi[n], pend_w32<SIGN>(n), a, b);
        lo[n] = a;
        hi[n] = b;
    }
}

// register k of the 32 (k = 0..31) in the two layouts used below
//   natural : k -> (k < 16 ? lo[k] : hi[k - 16])
//   split   : k -> (k even ? lo[k / 2] : hi[k / 2])      (what dif produces / dit consumes)
#define PH_NAT(k) ((k) < 16 ? lo[(k)&15] : hi[(k)&15])
#define PH_SPL(k) (((k)&1) ? hi[(k) >> 1] : lo[(k) >> 1])
#define PH_COL(k) pa[33 * (k)]  // plane element (row k, own column)
#define PH_ROW(k) pb[(k)]       // plane element (own row, column k)

// In: lo/hi natural = the window, lane l5 register r -> window index l5 + 32 r (re, im = the two
// channels).  Out: lo/hi natural = the circular convolution with the taps at the same indices.
// pa = plane + l5, pb = plane + 33 l5 (the item's exchange plane); twl = table row base for this
// lane (row k at twl[32 k]); hlo = hspec + l5, hhi = hspec - l5.
__device__ __forceinline__ void ols32_transform(cd (&lo)[16], cd (&hi)[16], double *pa, double *pb,
                                                const double2 *__restrict__ twl, const double2 *__restrict__ hlo,
                                                const double2 *__restrict__ hhi)
{
        // one exchange through the item's plane, real parts then imaginary parts:
        // WR(k): address the value of register k goes to;  RD(k): where register k comes from
#if PH_OLS_ABLATE & 1
#define PH_EXCHANGE(WREG, WADDR, RREG, RADDR)                 \
    do {                                                      \
        cd t_[32];                                            \
        _Pragma("unroll") for (int k = 0; k < 32; ++k)        \
            t_[k] = WREG(k);                                  \
        _Pragma("unroll") for (int k = 0; k < 32; ++k)        \
            RREG(k) = t_[k];                                  \
    } while (0)
#else
#define PH_EXCHANGE(WREG, WADDR, RREG, RADDR)                 \
    do {                                                      \
        double re_[32];                                       \
        _Pragma("unroll") for (int k = 0; k < 32; ++k)        \
            WADDR(k) = WREG(k).re;                            \
        wave_fence();                                         \
        _Pragma("unroll") for (int k = 0; k < 32; ++k)        \
            re_[k] = RADDR(k);                                \
        wave_fence();                                         \
        _Pragma("unroll") for (int k = 0; k < 32; ++k)        \
            WADDR(k) = WREG(k).im;                            \
        wave_fence();                                         \
        _Pragma("unroll") for (int k = 0; k < 32; ++k)        \
            RREG(k) = cd{re_[k], RADDR(k)};                   \
        wave_fence();                                         \
    } while (0)
#endif

        // the inter-pass twiddles W1024^(n1 k2) and the tap spectrum are not applied on their own:
        // they ride into the next 32-point DFT as pending factors.  The table is symmetric in
        // (register, lane), so the twiddles may sit on either side of the exchange: after it.
#if PH_OLS_ABLATE & 2
        double2 one_ = twl[32];
        asm volatile("" : "+v"(one_.x), "+v"(one_.y));  // (a register value the compiler knows nothing about)
#define PH_TAB(expr) one_
#else
#define PH_TAB(expr) (expr)
#endif
        const auto tw_fwd = [&](int k) {
            const double2 t = PH_TAB(twl[32 * k]);
            return cd{t.x, t.y};
        };
        const auto tw_inv = [&](int k) {
            const double2 t = PH_TAB(twl[32 * k]);
            return cd{t.x, -t.y};
        };
        // tap spectrum (scaled by 1/M) at k = 32 k1 + l5; upper half read as the conjugate mirror
        const auto taps_at = [&](int k1) {
            if (k1 < 16) {
                const double2 h = PH_TAB(hlo[32 * k1]);
                return cd{h.x, h.y};
            }
            const double2 h = PH_TAB(hhi[1024 - 32 * k1]);
            return cd{h.x, -h.y};
        };

        // ---- forward ------------------------------------------------------------------------
        dft32_dif<-1>(lo, hi);                         // A: over n2 -> k2 (split layout)
        PH_EXCHANGE(PH_SPL, PH_COL, PH_NAT, PH_ROW);   // X: (lane n1, reg k2) -> (lane k2, reg n1)
        dft32_dif_rt<-1>(lo, hi, tw_fwd);              // B + C: W1024^(n1 k2), then over n1 -> k1 (split)

        // ---- inverse: the same steps backwards, conjugate twiddles ---------------------------
        dft32_dit_rt<+1, false>(lo, hi, taps_at);      // tap spectrum, then over k1 -> n1 (natural)
        PH_EXCHANGE(PH_NAT, PH_ROW, PH_SPL, PH_COL);   // (lane k2, reg n1) -> (lane n1, reg k2 split)
        dft32_dit_rt<+1, true>(lo, hi, tw_inv);        // conj W1024^(n1 k2), then over k2 -> n2 (natural)

}

// The two halves of the transform on their own, for the partitioned form (fir_ols32p.hip: filters of
// 513 .. 4096 taps as several <= 512-tap spectra whose products are summed in the frequency domain).
//   forward : window (natural) -> spectrum, lane k2, register k1 in split layout: X[32 k1 + k2]
//   inverse : spectrum (split)  -> circular result (natural); no tap factor rides in
__device__ __forceinline__ void ols32_forward(cd (&lo)[16], cd (&hi)[16], double *pa, double *pb,
                                              const double2 *__restrict__ twl)
{
    const auto tw_fwd = [&](int k) {
        const double2 t = twl[32 * k];
        return cd{t.x, t.y};
    };
    dft32_dif<-1>(lo, hi);
    PH_EXCHANGE(PH_SPL, PH_COL, PH_NAT, PH_ROW);
    dft32_dif_rt<-1>(lo, hi, tw_fwd);
}
__device__ __forceinline__ void ols32_inverse_plain(cd (&lo)[16], cd (&hi)[16], double *pa, double *pb,
                                                    const double2 *__restrict__ twl)
{
    const auto tw_inv = [&](int k) {
        const double2 t = twl[32 * k];
        return cd{t.x, -t.y};
    };
    // over k1 -> n1 without factors: X[2m] = lo[m], X[2m + 1] = hi[m] -> x[n] = lo[n], x[16 + n] = hi[n]
    dft16p<+1>(lo, 0, 0);
    dft16p<+1>(hi, 0, 0);
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        cd a, b;
        bf<+1>(lo[n], pend_none(), hi[n], pend_w32<+1>(n), a, b);
        lo[n] = a;
        hi[n] = b;
    }
    PH_EXCHANGE(PH_NAT, PH_ROW, PH_SPL, PH_COL);
    dft32_dit_rt<+1, true>(lo, hi, tw_inv);
}
#undef PH_EXCHANGE
#undef PH_TAB

}  // namespace ols
}  // namespace pipehip
