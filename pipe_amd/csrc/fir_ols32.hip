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
#include "ols32_kernel.hpp"

namespace pipehip {
namespace ols {
namespace {

constexpr int kM = kM32;
constexpr int kHalf = kHalf32;
constexpr int kPlane = kPlane32;

template <typename TIn, typename TOut, bool MONO = false>
int launch32(const Plan::Impl &I, const void *d_in, void *d_out, const double *hist, Args32 a, hipStream_t s,
             KernelTimer *timer)
{
    auto kfn = fir_ols32_kernel<TIn, TOut, 0, false, false, MONO>;
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
                          static_cast<const double2 *>(I.hperm[I.cur].p), a, FuseArgs{}, FuseConst<1>{});
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
    a.HP = a.H;
    a.L = kM - a.HP;
    a.pairs = (channels + 1) / 2;
    a.odd = channels & 1;
    a.lines = lines;
    a.tiles_per_line = (int)((frames + a.L - 1) / a.L);
    // one channel: two tiles of the Line per complex sequence (tile t and tile t + half the tiles) -- the same
    // efficiency as a channel pair; a Line of one tile, or PIPE_HIP_OLS_MONO_ALONE (A/B), keeps the channel alone
    const bool mono = channels == 1 && a.tiles_per_line >= 2 && !PH_ENV_AB("PIPE_HIP_OLS_MONO_ALONE");
    if (mono) {
        a.tiles_per_line = (a.tiles_per_line + 1) / 2;
        a.mono_shift = (int64_t)a.tiles_per_line * a.L;
        a.odd = 0;
    }
    a.ipl = a.tiles_per_line * a.pairs;
    a.upl = (a.ipl + 1) / 2;
    a.nunits = (int64_t)a.upl * lines;
    if (mono) {
        if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32) {
            *kernel_name = "fir_ols_kernel<f32,f32,32x32>";
            return launch32<float, float, true>(I, d_in, d_out, hist, a, s, timer);
        }
        if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F32) {
            *kernel_name = "fir_ols_kernel<f64,f32,32x32>";
            return launch32<double, float, true>(I, d_in, d_out, hist, a, s, timer);
        }
        if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F64) {
            *kernel_name = "fir_ols_kernel<f32,f64,32x32>";
            return launch32<float, double, true>(I, d_in, d_out, hist, a, s, timer);
        }
        *kernel_name = "fir_ols_kernel<f64,f64,32x32>";
        return launch32<double, double, true>(I, d_in, d_out, hist, a, s, timer);
    }
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
