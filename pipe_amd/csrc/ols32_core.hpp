// The 1024-point transform of fir_ols32.hip as a reusable core: one transform per HALF-WAVE,
// 1024 = 32 x 32 (see fir_ols32.hip for the decomposition).  Included by fir_ols32.hip (the FIR
// alone) and chain_fused.hip (FIR -> biquad -> gain in one pass).
#pragma once

#include "ols_math.hpp"

namespace pipehip {
namespace ols {

constexpr int kM32 = 1024;
constexpr int kHalf32 = 513;          // H[0..512] (+1 pad)
constexpr int kPlane32 = 32 * 33;     // doubles per item exchange plane (stride 33)

// W32^e (SIGN < 0: forward, = cos - i sin) or its conjugate, e compile-time after unrolling
template <int SIGN>
__device__ __forceinline__ cd tw32(cd v, int e)
{
    constexpr double c[16] = {1.0,
                              0.98078528040323044913,
                              0.92387953251128675613,
                              0.83146961230254523708,
                              0.70710678118654752440,
                              0.55557023301960222474,
                              0.38268343236508977173,
                              0.19509032201612826785,
                              0.0,
                              -0.19509032201612826785,
                              -0.38268343236508977173,
                              -0.55557023301960222474,
                              -0.70710678118654752440,
                              -0.83146961230254523708,
                              -0.92387953251128675613,
                              -0.98078528040323044913};
    constexpr double s[16] = {0.0,
                              0.19509032201612826785,
                              0.38268343236508977173,
                              0.55557023301960222474,
                              0.70710678118654752440,
                              0.83146961230254523708,
                              0.92387953251128675613,
                              0.98078528040323044913,
                              1.0,
                              0.98078528040323044913,
                              0.92387953251128675613,
                              0.83146961230254523708,
                              0.70710678118654752440,
                              0.55557023301960222474,
                              0.38268343236508977173,
                              0.19509032201612826785};
    if (e == 0)
        return v;
    if (e == 8)  // -i forward, +i inverse
        return SIGN < 0 ? cd{v.im, -v.re} : cd{-v.im, v.re};
    const cd w{c[e], SIGN < 0 ? -s[e] : s[e]};
    return cmul(v, w);
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
        hi[n] = tw32<SIGN>(b, n);
    }
    dft16<SIGN>(lo);
    dft16<SIGN>(hi);
}

// 32-point DFT, decimation in time: in X[2m] = lo[m], X[2m + 1] = hi[m];
// out x[n] = lo[n], x[16 + n] = hi[n]
template <int SIGN>
__device__ __forceinline__ void dft32_dit(cd (&lo)[16], cd (&hi)[16])
{
    dft16<SIGN>(lo);
    dft16<SIGN>(hi);
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const cd t = tw32<SIGN>(hi[n], n);
        const cd a{lo[n].re + t.re, lo[n].im + t.im};
        const cd b{lo[n].re - t.re, lo[n].im - t.im};
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

        // twiddles W1024^(k * l5), k = 1..31, applied to register REG(k); reads run G ahead
#define PH_TWIDDLE(REG, CONJ)                                                          \
    do {                                                                               \
        constexpr int G = 5;                                                           \
        double2 w_[2][G];                                                              \
        _Pragma("unroll") for (int j = 0; j < G; ++j) w_[0][j] = twl[32 * (1 + j)];    \
        _Pragma("unroll") for (int g = 0; g < 7; ++g)                                  \
        {                                                                              \
            if (g + 1 < 7) {                                                           \
                _Pragma("unroll") for (int j = 0; j < G; ++j)                          \
                {                                                                      \
                    const int kn = 1 + G * (g + 1) + j;                                \
                    if (kn < 32)                                                       \
                        w_[(g + 1) & 1][j] = twl[32 * kn];                             \
                }                                                                      \
            }                                                                          \
            _Pragma("unroll") for (int j = 0; j < G; ++j)                              \
            {                                                                          \
                const int k = 1 + G * g + j;                                           \
                if (k < 32) {                                                          \
                    const cd ww{w_[g & 1][j].x, w_[g & 1][j].y};                       \
                    REG(k) = (CONJ) ? cmulc(REG(k), ww) : cmul(REG(k), ww);            \
                }                                                                      \
            }                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                         \
        }                                                                              \
    } while (0)

        // ---- forward ------------------------------------------------------------------------
        dft32_dif<-1>(lo, hi);                         // A: over n2 -> k2 (split layout)
        PH_TWIDDLE(PH_SPL, false);                     // B: W1024^(n1 k2), n1 = l5
        PH_EXCHANGE(PH_SPL, PH_COL, PH_NAT, PH_ROW);   // X: (lane n1, reg k2) -> (lane k2, reg n1)
        dft32_dif<-1>(lo, hi);                         // C: over n1 -> k1 (split), k = 32 k1 + l5

        // ---- tap spectrum (scaled by 1/M); upper half read as the conjugate mirror -----------
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) {
            if (k1 < 16) {
                const double2 h = hlo[32 * k1];
                PH_SPL(k1) = cmul(PH_SPL(k1), cd{h.x, h.y});
            } else {
                const double2 h = hhi[1024 - 32 * k1];
                PH_SPL(k1) = cmulc(PH_SPL(k1), cd{h.x, h.y});
            }
        }

        // ---- inverse: the same steps backwards, conjugate twiddles ---------------------------
        dft32_dit<+1>(lo, hi);                         // over k1 -> n1 (natural)
        PH_TWIDDLE(PH_NAT, true);                      // conj W1024^(n1 k2), k2 = l5
        PH_EXCHANGE(PH_NAT, PH_ROW, PH_SPL, PH_COL);   // (lane k2, reg n1) -> (lane n1, reg k2 split)
        dft32_dit<+1>(lo, hi);                         // over k2 -> n2 (natural): y[l5 + 32 n2]

#undef PH_TWIDDLE
#undef PH_EXCHANGE
}

}  // namespace ols
}  // namespace pipehip
