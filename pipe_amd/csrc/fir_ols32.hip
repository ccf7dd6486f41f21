// Overlap-save FIR, second decomposition: ONE 1024-POINT TRANSFORM PER HALF-WAVE, 1024 = 32 x 32.
//
// fir_ols.hip holds 16 complex values per lane (1024 = 16 x 16 x 4): three in-register DFT steps
// and therefore TWO transposes between lane index and register index per transform -- one through
// LDS, one as 64 v_permlane swaps (128 per item at ~7.6 cycles = 18 % of the VALU issue time of an
// item, profiles/r01_fir_ols_headline.txt).  With 32 complex values per lane 32 x 32 = 1024 needs
// only TWO in-register steps and ONE transpose:
//     A : 32-point DFT over n2 in registers (decimation in frequency)   n = n1 + 32 n2, lane = n1
//     B : twiddle W1024^(n1 k2)                                          (LDS table, exact entries)
//     X : LDS exchange (lane n1, reg k2) -> (lane k2, reg n1), stride 33: conflict-free both ways
//     C : 32-point DFT over n1 in registers                              k = 32 k1 + k2
// then the tap spectrum and the same steps backwards (decimation in time, conjugate twiddles).
// A wave's two 32-lane halves work on two different items with one instruction stream; a wave
// needs ~200 VGPRs, so 8 waves (one 512-thread workgroup) live on a CU, two per SIMD.
// Per item: 1068 float64 instructions (1032 before), no swaps, one twiddle table (31 reads per
// transform instead of 27 + the row table), 64 KB through the exchange planes as before.
#include <cstdlib>
#include <vector>

#include <hip/hip_ext.h>

#include "common.hpp"
#include "fir_hist.hpp"
#include "fir_ols.hpp"
#include "fir_ols_impl.hpp"
#include "ols_math.hpp"

namespace pipehip {
namespace ols {
namespace {

constexpr int kM = 1024;
constexpr int kWaves32 = 8;          // waves per workgroup = per CU
constexpr int kHalf = 513;           // H[0..512] (+1 pad)
constexpr int kPlane = 32 * 33;      // doubles per item exchange plane (stride 33)
constexpr unsigned kOut = 0x80000000u;  // a buffer offset beyond any num_records: loads 0, stores dropped

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

struct Args32 {
    int64_t frames;       // frames per Line in this call
    int64_t line_stride;  // elements between Lines
    int C, N, H;
    int L;                // valid outputs per tile
    int pairs, lines, tiles_per_line;
    int ipl, upl;         // items per Line (tiles x pairs); units (item pairs) per Line
    int64_t nunits;
    int d_slot, d_line;   // the wave stride of the launch as (slot, Line) digits
    double *hist_new;
};

// register k of the 32 (k = 0..31) in the two layouts used below
//   natural : k -> (k < 16 ? lo[k] : hi[k - 16])
//   split   : k -> (k even ? lo[k / 2] : hi[k / 2])      (what dif produces / dit consumes)
#define PH_NAT(k) ((k) < 16 ? lo[(k)&15] : hi[(k)&15])
#define PH_SPL(k) (((k)&1) ? hi[(k) >> 1] : lo[(k) >> 1])

template <typename TIn, typename TOut>
__global__ void __launch_bounds__(kWaves32 * 64)
fir_ols32_kernel(const TIn *__restrict__ in_base, TOut *__restrict__ out_base, const double *__restrict__ hist_base,
                 const double2 *__restrict__ tw_g, const double2 *__restrict__ hperm_g, const Args32 a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2 *hspec = reinterpret_cast<double2 *>(smem_raw);       // H[0..512] (+ pad)
    double2 *tws = hspec + kHalf + 1;                              // W1024^(k n), k = 1..31, n = 0..31
    double *planes = reinterpret_cast<double *>(tws + 31 * 32);   // [waves][2][kPlane]

    fir_history_carry(in_base, hist_base, a.hist_new, a.frames, a.line_stride, a.H, a.C, a.lines);
    for (int i = threadIdx.x; i < kHalf; i += kWaves32 * 64)
        hspec[i] = hperm_g[i];
    for (int i = threadIdx.x; i < 31 * 32; i += kWaves32 * 64)
        tws[i] = tw_g[32 + i];
    __syncthreads();

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, l5 = lane & 31;
    double *plane = planes + (wave * 2 + half) * kPlane;
    double *pa = plane + l5;       // (row r, this lane's column): pa[33 r]
    double *pb = plane + 33 * l5;  // (this lane's row, column c): pb[c]
    const double2 *__restrict__ twl = tws + l5 - 32;  // row k at twl[32 k]
    const double2 *__restrict__ hlo = hspec + l5;     // H[32 k1 + l5]
    const double2 *__restrict__ hhi = hspec - l5;     // conj side: H[1024 - 32 k1 - l5]

    using In2 = typename Pair<TIn>::type;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int nb = (int)gridDim.x;
    // consecutive units go to consecutive blocks of the same XCD (block b runs on XCD b % 8):
    // neighbouring tiles share their overlap through that XCD's L2
    const int xb = nb % 8 == 0 ? ((int)blockIdx.x % 8) * (nb / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
    const int64_t wave_global = (int64_t)wave_u * nb + xb;
    const int64_t wave_stride = (int64_t)nb * kWaves32;
    int line = 0, slot = 0;
    if (wave_global < a.nunits) {
        line = __builtin_amdgcn_readfirstlane((int)(wave_global / a.upl));
        slot = __builtin_amdgcn_readfirstlane((int)(wave_global % a.upl));
    }
    const unsigned in_step = (unsigned)(32 * a.C * sizeof(TIn));    // 32 frames
    const unsigned out_step = (unsigned)(32 * a.C * sizeof(TOut));
    auto bytes31 = [](int64_t n) { return (int)(n < 0x7FFFFFFF ? n : 0x7FFFFFFF); };
    const int64_t last = a.frames - 1;

    for (int64_t unit = wave_global; unit < a.nunits; unit += wave_stride) {
        // ---- the unit's two items: item0 = 2 slot (half 0), item0 + 1 (half 1) -----------------
        const int item0 = 2 * slot;
        const int tile0 = __builtin_amdgcn_readfirstlane(item0 / a.pairs);
        const int pair0 = __builtin_amdgcn_readfirstlane(item0 - tile0 * a.pairs);
        int tile = tile0, pair = pair0 + half;
        if (pair >= a.pairs) {
            pair = 0;
            tile = tile0 + 1;
        }
        const bool valid = item0 + half < a.ipl;
        const int c0 = 2 * pair;
        const int64_t fr00 = (int64_t)tile0 * a.L - a.H;  // first window frame of half 0's item

        cd lo[16], hi[16];
        // ---- the window: lane l5, register r -> window index l5 + 32 r ---------------------------
        if (tile0 > 0) {
            // both windows start inside the Line: one buffer resource based at half 0's window,
            // 32-bit lane offsets; frames past the end of the Line read as zero
            const TIn *base = in_base + (int64_t)line * a.line_stride + fr00 * a.C;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<TIn *>(base), 0, bytes31((a.frames - fr00) * a.C * (int64_t)sizeof(TIn)), 0x00020000);
            const unsigned v0 = valid ? (unsigned)((((tile - tile0) * a.L + l5) * a.C + c0) * (int)sizeof(TIn)) : kOut;
            In2 pf[32];
#pragma unroll
            for (int r = 0; r < 32; ++r)
                pf[r] = buf_load_pair<TIn>(rs, v0 + (unsigned)r * in_step);
#pragma unroll
            for (int r = 0; r < 32; ++r)
                PH_NAT(r) = cd{(double)pf[r].x, (double)pf[r].y};
        } else {
            // a Line's first tile: its head is the history
            const TIn *__restrict__ in = in_base + (int64_t)line * a.line_stride;
            const double *__restrict__ hist = hist_base + (int64_t)line * a.H * a.C;
            const int64_t fr0 = (int64_t)tile * a.L - a.H;
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int64_t g = fr0 + l5 + 32 * r;
                double re = 0.0, im = 0.0;
                if (valid) {
                    if (g >= 0) {
                        if (g <= last) {
                            re = (double)in[g * a.C + c0];
                            im = (double)in[g * a.C + c0 + 1];
                        }
                    } else {
                        re = hist[(g + a.H) * a.C + c0];
                        im = hist[(g + a.H) * a.C + c0 + 1];
                    }
                }
                PH_NAT(r) = cd{re, im};
            }
        }
        // the next unit's coordinates (uniform)
        const int cur_line = line;
        slot += a.d_slot;
        if (slot >= a.upl) {
            slot -= a.upl;
            ++line;
        }
        line += a.d_line;

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
#define PH_COL(k) pa[33 * (k)]  // (row k, own column)
#define PH_ROW(k) pb[(k)]       // (own row, column k)

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

        // ---- store the valid part: window index i >= H is frame t0 + i - H --------------------
        {
            const int64_t t00 = (int64_t)tile0 * a.L;
            TOut *base = out_base + (int64_t)cur_line * a.line_stride + t00 * a.C;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                base, 0, bytes31((a.frames - t00) * a.C * (int64_t)sizeof(TOut)), 0x00020000);
            const int o0 = (((tile - tile0) * a.L + l5 - a.H) * a.C + c0) * (int)sizeof(TOut);
            const int i0 = valid ? l5 - a.H : -2048;  // window index - H of register 0: outputs need >= 0
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int off = o0 + r * (int)out_step;
                buf_store_pair<TOut>(rs, i0 + 32 * r >= 0 ? (unsigned)off : kOut, PH_NAT(r).re, PH_NAT(r).im);
            }
        }
    }
#undef PH_TWIDDLE
#undef PH_EXCHANGE
#undef PH_COL
#undef PH_ROW
}
#undef PH_NAT
#undef PH_SPL

template <typename TIn, typename TOut>
int launch32(const Plan::Impl &I, const void *d_in, void *d_out, const double *hist, Args32 a, hipStream_t s,
             KernelTimer *timer)
{
    auto kfn = fir_ols32_kernel<TIn, TOut>;
    const size_t lds = sizeof(double2) * (kHalf + 1 + 31 * 32) + sizeof(double) * (size_t)kPlane * 2 * kWaves32;
    PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    const int64_t resident = I.cus;  // one 512-thread workgroup per CU
    const int64_t wanted = (a.nunits + kWaves32 - 1) / kWaves32;
    const unsigned grid = (unsigned)(wanted < resident ? wanted : resident);
    const int64_t stride = (int64_t)grid * kWaves32;
    a.d_slot = (int)(stride % a.upl);
    a.d_line = (int)(stride / a.upl);
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    if (timer)
        PH_TRY(timer->pair(&ev_a, &ev_b));
    hipExtLaunchKernelGGL(kfn, dim3(grid), dim3(kWaves32 * 64), lds, s, ev_a, ev_b, 0, static_cast<const TIn *>(d_in),
                          static_cast<TOut *>(d_out), hist, static_cast<const double2 *>(I.tw32.p),
                          static_cast<const double2 *>(I.hperm[I.cur].p), a);
    PH_HIP(hipGetLastError());
    return PIPE_HIP_OK;
}

}  // namespace

int init_ols32_tables(Plan::Impl *I)
{
    std::vector<double> t(2 * 32 * 32);
    for (int k = 0; k < 32; ++k)
        for (int n = 0; n < 32; ++n) {
            const long double ang = -2.0L * (long double)kPi * (k * n) / kM;
            t[2 * (k * 32 + n)] = (double)cosl(ang);
            t[2 * (k * 32 + n) + 1] = (double)sinl(ang);
        }
    PH_TRY(I->tw32.alloc(sizeof(double) * t.size()));
    PH_HIP(hipMemcpy(I->tw32.p, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
    return PIPE_HIP_OK;
}

int run_ols32(const Plan::Impl &I, const void *d_in, int in_dtype, void *d_out, int out_dtype, const double *hist,
              double *hist_new, int64_t frames, int channels, int lines, hipStream_t s, const char **kernel_name,
              KernelTimer *timer)
{
    Args32 a{};
    a.frames = frames;
    a.hist_new = hist_new;
    a.line_stride = frames * channels;
    a.C = channels;
    a.N = I.N;
    a.H = I.N - 1;
    a.L = kM - a.H;
    a.pairs = channels / 2;
    a.lines = lines;
    a.tiles_per_line = (int)((frames + a.L - 1) / a.L);
    a.ipl = a.tiles_per_line * a.pairs;
    a.upl = (a.ipl + 1) / 2;
    a.nunits = (int64_t)a.upl * lines;
    if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32) {
        *kernel_name = "fir_ols_kernel<f32,f32,32x32>";
        return launch32<float, float>(I, d_in, d_out, hist, a, s, timer);
    }
    if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F32) {
        *kernel_name = "fir_ols_kernel<f64,f32,32x32>";
        return launch32<double, float>(I, d_in, d_out, hist, a, s, timer);
    }
    if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F64) {
        *kernel_name = "fir_ols_kernel<f32,f64,32x32>";
        return launch32<float, double>(I, d_in, d_out, hist, a, s, timer);
    }
    *kernel_name = "fir_ols_kernel<f64,f64,32x32>";
    return launch32<double, double>(I, d_in, d_out, hist, a, s, timer);
}

}  // namespace ols
}  // namespace pipehip
