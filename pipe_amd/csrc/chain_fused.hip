// FIR -> biquad -> gain as ONE kernel: the epilogue of ols32_kernel.hpp behind the 32 x 32
// overlap-save transform.  One read of the float32 input, one write of the float32 result; the
// float64 intermediates of the staged chain (chain.hip: 4x the algorithmic traffic) never exist.
//
// This file is the host side: the matrices the epilogue needs (zero-input state transitions of the
// biquad cascade, in long double), the look-back records, and the launch.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <hip/hip_ext.h>

#include "chain_fused.hpp"
#include "fir_ols_impl.hpp"
#include "ols32_kernel.hpp"

namespace pipehip {
namespace fused {
namespace {

using ols::Args32;
using ols::FuseArgs;
using ols::FuseConst;
using ols::kWaves32;

constexpr int kMaxN2 = 4;  // 2 sections

typedef long double ld;
struct Mat {
    int n;
    ld m[kMaxN2][kMaxN2];
};
Mat identity(int n)
{
    Mat r{};
    r.n = n;
    for (int i = 0; i < n; ++i)
        r.m[i][i] = 1.0L;
    return r;
}
Mat mul(const Mat &a, const Mat &b)
{
    Mat r{};
    r.n = a.n;
    for (int i = 0; i < a.n; ++i)
        for (int j = 0; j < a.n; ++j) {
            ld acc = 0;
            for (int k = 0; k < a.n; ++k)
                acc += a.m[i][k] * b.m[k][j];
            r.m[i][j] = acc;
        }
    return r;
}
Mat power(Mat b, long e)
{
    Mat r = identity(b.n);
    while (e > 0) {
        if (e & 1)
            r = mul(r, b);
        b = mul(b, b);
        e >>= 1;
    }
    return r;
}
// one zero-input step of the cascade: column j = the state after one frame of silence started from
// unit state j (state order s1_0, s2_0, s1_1, s2_1, ...), the recurrence of oracle/dsp_oracle.h
Mat one_step(const double *c, int S)
{
    Mat r{};
    r.n = 2 * S;
    for (int j = 0; j < 2 * S; ++j) {
        ld st[kMaxN2] = {0};
        st[j] = 1.0L;
        ld x = 0.0L;
        for (int s = 0; s < S; ++s) {
            const ld b0 = c[5 * s], b1 = c[5 * s + 1], b2 = c[5 * s + 2], a1 = c[5 * s + 3], a2 = c[5 * s + 4];
            const ld y = b0 * x + st[2 * s];
            st[2 * s] = -a1 * y + (b1 * x + st[2 * s + 1]);
            st[2 * s + 1] = -a2 * y + b2 * x;
            x = y;
        }
        for (int i = 0; i < 2 * S; ++i)
            r.m[i][j] = st[i];
    }
    return r;
}
void store_flat(double *dst, const Mat &m)
{
    for (int i = 0; i < m.n; ++i)
        for (int j = 0; j < m.n; ++j)
            dst[i * m.n + j] = (double)m.m[i][j];
}

// The state after a Line's LAST frame.  The fused kernel leaves the true start state of the
// 32-position segment that holds that frame (seg_state); here one half-wave per (Line, channel)
// series stages the segment's input window in LDS, computes the FIR output of its <= 32 frames
// in the direct form (the oracle's own ordered sum, one frame per lane, taps as scalar operands)
// and walks the recurrence over them.  Runs behind the fused kernel on the same stream -- only when
// the Line does not end on a segment boundary (frames % 32 != 0); otherwise the fused kernel has the
// state itself -- and writes the series' state slot in its place.
struct TailArgs {
    int64_t frames, line_stride;
    int C, pairs, N, H, HP, L, tiles_per_line, nseries;  // nseries = lines * C; pairs = ceil(C / 2)
    const double *seg_state;                       // [lines][pairs][2][2S]
    unsigned long long *own;                       // [lines][pairs][2 slots][...]: ols32_kernel.hpp
    unsigned epoch;
};
constexpr int kTailSeries = 8;  // series (half-waves) per workgroup
typedef const __attribute__((address_space(4))) double *tail_const_f64;
template <int S, typename T>
__global__ void __launch_bounds__(32 * kTailSeries)
chain_tail_kernel(const T *__restrict__ in_base, const T *__restrict__ hist_base,
                  const double *__restrict__ taps, const TailArgs a, const FuseConst<S> fc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
    constexpr int N2 = 2 * S;
    const int local = threadIdx.x >> 5, l5 = threadIdx.x & 31;
    const int sid = (int)blockIdx.x * kTailSeries + local;
    const int wlen = 32 + a.H;  // window: frames f0 - H .. f0 + 31
    double *xs = reinterpret_cast<double *>(tail_smem) + (size_t)local * wlen;
    const bool live = sid < a.nseries;
    const int line = live ? sid / a.C : 0, ch = live ? sid - line * a.C : 0;
    const int tile = a.tiles_per_line - 1;
    const int64_t t0 = (int64_t)tile * a.L;
    const int len = (int)(a.frames - t0);
    const int plast = a.HP + len - 1, kl = plast >> 5, jl = plast & 31;
    const int64_t f0 = t0 + 32 * kl - a.HP;  // first frame of the segment
    if (live) {
        const T *__restrict__ in = in_base + (int64_t)line * a.line_stride + ch;
        const T *__restrict__ hist = hist_base + (int64_t)line * a.H * a.C + ch;
        // all loads of a lane first, then the LDS stores: one memory round trip, not one per trip
        constexpr int kMaxTrips = (32 + 511 + 31) / 32;  // taps <= 512
        double v[kMaxTrips];
#pragma unroll
        for (int t = 0; t < kMaxTrips; ++t) {
            const int i = l5 + 32 * t;
            const int64_t g = f0 - a.H + i;
            v[t] = 0.0;
            if (i < wlen) {
                if (g >= 0) {
                    if (g < a.frames)
                        v[t] = (double)in[g * a.C];
                } else if (g >= -(int64_t)a.H) {
                    v[t] = (double)hist[(g + a.H) * a.C];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < kMaxTrips; ++t) {
            const int i = l5 + 32 * t;
            if (i < wlen)
                xs[i] = v[t];
        }
    }
    __syncthreads();
    if (!live)
        return;
    // FIR output of frame f0 + l5: acc = fma(h[k], x[f - k], acc), k = 0..N-1
    double y = 0.0;
    {
        // (four interleaved partial sums: the dependent fma chain would otherwise be 256 deep;
        // this value only seeds a state, nothing compares it bit for bit)
        const double *w = xs + a.H + l5;
        const tail_const_f64 h = (tail_const_f64)taps;
        double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
        int k = 0;
#pragma unroll 2
        for (; k + 4 <= a.N; k += 4) {
            y0 = __builtin_fma(h[k], w[-k], y0);
            y1 = __builtin_fma(h[k + 1], w[-k - 1], y1);
            y2 = __builtin_fma(h[k + 2], w[-k - 2], y2);
            y3 = __builtin_fma(h[k + 3], w[-k - 3], y3);
        }
        for (; k < a.N; ++k)
            y0 = __builtin_fma(h[k], w[-k], y0);
        y = (y0 + y1) + (y2 + y3);
    }
    double st[N2];
    const double *sp = a.seg_state + ((int64_t)(line * a.pairs + ch / 2) * 2 + (ch & 1)) * N2;
#pragma unroll
    for (int j = 0; j < N2; ++j)
        st[j] = sp[j];
    for (int c = 0; c <= jl; ++c)
        (void)ols::biquad_step<S>(__shfl(y, c, 32), st, fc);
    if (l5 == 0) {
        constexpr int NV = 2 * N2;
        unsigned long long *slot =
            ols::own_write_slot<NV>(a.own + (int64_t)(line * a.pairs + ch / 2) * (2 * 2 * NV), a.epoch);
#pragma unroll
        for (int j = 0; j < N2; ++j)
            ols::own_store(slot + 2 * ((ch & 1) * N2 + j), a.epoch, st[j]);
    }
}

// The biquad stage's own state array <-> the state slots, when a chain changes between its staged
// and its fused form (one thread per channel pair).
// (C channels in `pairs` = ceil(C / 2) pairs per Line: an odd count's last pair holds ONE channel -- the other half of
// its slots is a channel that does not exist, zero on the way in and dropped on the way out)
template <int S>
__global__ void chain_state_import_kernel(const double *__restrict__ state, unsigned long long *own, int nseries2,
                                          unsigned epoch, int C, int pairs)
{
    constexpr int N2 = 2 * S, NV = 4 * S;
    const int sid = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (sid >= nseries2)
        return;
    const int line = sid / pairs, c0 = 2 * (sid - line * pairs);
    unsigned long long *o = own + (int64_t)sid * (2 * 2 * NV);
    for (int slot = 0; slot < 2; ++slot)  // equal tags: the reader takes slot 0, the writer slot 1
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = c0 + j / N2;
            ols::own_store(o + slot * (2 * NV) + 2 * j, epoch, c < C ? state[((int64_t)line * C + c) * N2 + j % N2] : 0.0);
        }
}
template <int S>
__global__ void chain_state_export_kernel(double *__restrict__ state, const unsigned long long *own, int nseries2,
                                          unsigned epoch, int C, int pairs)
{
    constexpr int N2 = 2 * S, NV = 4 * S;
    const int sid = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (sid >= nseries2)
        return;
    const int line = sid / pairs, c0 = 2 * (sid - line * pairs);
    double pay[NV];
    ols::own_read<NV>(own + (int64_t)sid * (2 * 2 * NV), epoch, pay);  // (epoch: one no launch has used)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = c0 + j / N2;
        if (c < C)
            state[((int64_t)line * C + c) * N2 + j % N2] = pay[j];
    }
}

}  // namespace

// S = 1 .. 4 as a compile-time constant
#define PH_FOR_SECTIONS(Sv, STMT)                  \
    switch (Sv) {                                  \
    case 1: { constexpr int SC = 1; STMT; } break; \
    case 2: { constexpr int SC = 2; STMT; } break; \
    case 3: { constexpr int SC = 3; STMT; } break; \
    default: { constexpr int SC = 4; STMT; } break; \
    }

struct Plan::Impl {
    DevBuf rec, mats[2], seg_state, own, prof;
    PinnedBuf err;            // host-resident, device-visible: the host reads it after any synchronisation
    int *err_dev = nullptr;   // the kernel's alias of it
    int own_series = 0;       // channel pairs the slots were sized for
    int own_C = 0, own_pairs = 1;  // ... as channels and pairs per Line
    bool in_slots = false;    // the cascade's state lives in the slots (else in the biquad stage's array)
    double *bq_state = nullptr;
    int cur_mats = 0;
    AsyncUpload upload;  // pinned staging of the matrices
    std::vector<double> coeffs;
    int S = 0, H = -1;
    bool has_gain = false;
    double gain = 1.0;
    FuseConst<1> c1{};
    FuseConst<2> c2{};
    FuseConst<3> c3{};
    FuseConst<4> c4{};
    template <int SC>
    FuseConst<SC> &fc()
    {
        if constexpr (SC == 1)
            return c1;
        else if constexpr (SC == 2)
            return c2;
        else if constexpr (SC == 3)
            return c3;
        else
            return c4;
    }
    int D = 1 << 30;
    unsigned epoch = 0;
    size_t rec_granules = 0;
    bool err_checked = true;
    hipStream_t last_stream = nullptr;
    int debug_withhold = -1;       // the next launch only
    double debug_limit_us = 0.0;
};

Plan::Plan() : impl_(new Impl) {}
Plan::~Plan() { delete impl_; }

bool Plan::enabled()
{
    static const bool on = [] {
        const char *e = std::getenv("PIPE_HIP_CHAIN_FUSED");
        return !(e && e[0] == '0');
    }();
    return on;
}

// (re)build everything that depends on the coefficients or the tap count.  Every section gets its own
// table of 2 x 2 matrices (the kernel runs the sections one after the other, ols32_kernel.hpp).
int Plan::prepare(const double *coeffs, int S, int ntaps, hipStream_t s)
{
    Impl &I = *impl_;
    const int H = (ntaps - 1 + 31) / 32 * 32, L = ols::kM32 - H;  // (the padded history: Plan::run)
    if (I.S == S && I.H == H && I.coeffs.size() == (size_t)5 * S &&
        std::memcmp(I.coeffs.data(), coeffs, sizeof(double) * 5 * S) == 0)
        return PIPE_HIP_OK;
    if (S < 1 || S > kMaxFusedSections)
        return PIPE_HIP_EINVAL;
    constexpr int kNever = 1 << 30;  // the filter does not forget within a look-back window
    // the matrices are double-buffered on the device and staged through pinned memory: a
    // coefficient mutation uploads them on the launch stream, nothing waits for the device
    const size_t mm = 4, per_section = (size_t)ols::kMatCount * mm;
    const size_t bytes = sizeof(double) * per_section * kMaxFusedSections;
    if (!I.mats[0].p) {
        PH_TRY(I.mats[0].alloc(bytes));
        PH_TRY(I.mats[1].alloc(bytes));
        PH_TRY(I.err.alloc(sizeof(int)));
        *static_cast<volatile int *>(I.err.p) = 0;
        void *alias = nullptr;
        PH_HIP(hipHostGetDevicePointer(&alias, I.err.p, 0));
        I.err_dev = static_cast<int *>(alias);
    }
    void *host = nullptr;
    PH_TRY(I.upload.stage(bytes, &host));
    int Dmax = 1;
    for (int sec = 0; sec < S; ++sec) {
        const double *c = coeffs + 5 * sec;
        double *h = static_cast<double *>(host) + per_section * sec;
        const Mat M = one_step(c, 1);
        const Mat M32 = power(M, 32);
        const Mat ML = power(M, L);
        std::vector<Mat> T(33);
        T[0] = identity(2);
        for (int j = 1; j <= 32; ++j)
            T[j] = mul(T[j - 1], ML);
        int D = kNever;
        for (int j = 1; j <= 32 && D == kNever; ++j) {
            ld big = 0;
            for (int i = 0; i < 2; ++i)
                for (int k = 0; k < 2; ++k)
                    big = std::fmax(big, std::fabs(T[j].m[i][k]));
            if (big < 0x1p-60L)
                D = j;
        }
        Dmax = D > Dmax ? D : Dmax;
        Mat ak = identity(2);
        for (int j = 0; j <= 16; ++j) {
            store_flat(h + (ols::kMatAk + j) * mm, ak);
            ak = mul(ak, M32);
        }
        store_flat(h + ols::kMatML * mm, ML);
        store_flat(h + ols::kMatT32 * mm, T[32]);
        for (int j = 0; j <= 32; ++j)
            store_flat(h + (ols::kMatTj + j) * mm, T[j]);
        const int k0 = H / 32;
        for (int k = 0; k < 32; ++k)
            store_flat(h + (ols::kMatPk + k) * mm, k > k0 ? power(M, 32L * (k - k0)) : identity(2));
        {
            Mat w = identity(2);
            for (int k = 0; k <= 32; ++k) {
                store_flat(h + (ols::kMatWw + k) * mm, w);
                w = mul(w, T[32]);
            }
        }
        {
            // g_i = M^(31 - i) c, c = (b1 - a1 b0, b2 - a2 b0): sample i of a segment in its zero-start end state
            const ld b0 = c[0], b1 = c[1], b2 = c[2], a1 = c[3], a2 = c[4];
            ld g[2] = {b1 - a1 * b0, b2 - a2 * b0};
            double *gz = h + ols::kMatGz * mm;
            for (int i = 31; i >= 0; --i) {
                gz[2 * i] = (double)g[0];
                gz[2 * i + 1] = (double)g[1];
                const ld t0 = M.m[0][0] * g[0] + M.m[0][1] * g[1], t1 = M.m[1][0] * g[0] + M.m[1][1] * g[1];
                g[0] = t0;
                g[1] = t1;
            }
        }
    }
    I.D = Dmax;  // (the slowest section decides)
    std::memcpy(I.c1.c, coeffs, sizeof(double) * 5);
    std::memcpy(I.c2.c, coeffs, sizeof(double) * 5 * (S < 2 ? 1 : 2));
    std::memcpy(I.c3.c, coeffs, sizeof(double) * 5 * (S < 3 ? S : 3));
    std::memcpy(I.c4.c, coeffs, sizeof(double) * 5 * (S < 4 ? S : 4));
    I.c1.D = I.c2.D = I.c3.D = I.c4.D = Dmax;
    I.cur_mats ^= 1;
    PH_TRY(I.upload.commit(I.mats[I.cur_mats].p, bytes, s));
    I.coeffs.assign(coeffs, coeffs + 5 * S);
    I.S = S;
    I.H = H;
    return PIPE_HIP_OK;
}

// Can this cascade run fused on a call of `frames` frames?  One section: always (slow filters take the
// general look-back, a ragged end the tail kernel).  Two to four sections: forgetful filters only (every section
// forgets within a look-back window).
bool Plan::accepts(const double *coeffs, int S, int ntaps, int64_t frames, hipStream_t s)
{
    if (S == 1)
        return true;
    (void)frames;
    if (S < 2 || S > kMaxFusedSections || PH_ENV_AB("PIPE_HIP_CHAIN_GENERAL") || PH_ENV_AB("PIPE_HIP_CHAIN_ONE_SECTION"))
        return false;
    if (prepare(coeffs, S, ntaps, s) != PIPE_HIP_OK)
        return false;
    return impl_->D <= 32;
}

// Precondition: the stream the last launch went to has been synchronised (every caller has just
// waited for the buffer it hands back).  The flag lives in pinned host memory, so this is a plain
// read -- cheap enough for every synchronous entry point, not only FlushFunc.
int Plan::poll_error()
{
    Impl &I = *impl_;
    if (!I.err.p || I.err_checked)
        return PIPE_HIP_OK;
    volatile int *e = static_cast<volatile int *>(I.err.p);
    I.err_checked = true;
    if (*e != 0) {
        *e = 0;
        return PIPE_HIP_EHIP;
    }
    return PIPE_HIP_OK;
}

// Every workgroup of a fused launch must be resident at once (tiles wait for their predecessors):
// one 512-thread workgroup with this much LDS has to fit a CU.  Asked once per kernel form; a
// device (or a runtime LDS carve-out) where it does not fit takes the staged chain instead.
// tables + exchange planes + (LOCAL) one ring of records per section and the round counters
template <int S, bool LOCAL>
static constexpr size_t fused_lds_bytes()
{
    return sizeof(double2) * (ols::kHalf32 + 1 + 31 * 32) + sizeof(double) * (size_t)ols::kPlane32 * 2 * kWaves32 +
           (LOCAL ? (S < 2 ? sizeof(ols::LocalRec<4>) * ols::kLocalRing : S * sizeof(ols::LocalRec<4>) * ols::kLocalRing2) +
                        4 * sizeof(unsigned)
                  : 0);
}

template <typename T, int S, bool GENERAL, bool LOCAL>
static bool form_fits()
{
    auto kfn = ols::fir_ols32_kernel<T, T, S, GENERAL, LOCAL>;
    const size_t lds = fused_lds_bytes<S, LOCAL>();
    int per_cu = 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kfn), kWaves32 * 64, lds) !=
            hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return per_cu >= 1;
}

bool Plan::launchable()
{
    static std::mutex mu;
    static int known[64] = {};  // per device: 0 = not asked, 1 = fits, 2 = does not
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    dev &= 63;
    std::lock_guard<std::mutex> lock(mu);
    if (!known[dev])
        known[dev] = form_fits<float, 1, true, false>() && form_fits<float, 1, false, true>() && form_fits<float, 1, false, false>() &&
                             form_fits<float, 2, false, true>() && form_fits<float, 2, false, false>() &&
                             form_fits<double, 1, true, false>() && form_fits<double, 1, false, true>() &&
                             form_fits<double, 1, false, false>() && form_fits<double, 2, false, true>() &&
                             form_fits<double, 2, false, false>() && form_fits<float, 3, false, false>() &&
                             form_fits<float, 4, false, false>() && form_fits<double, 3, false, false>() &&
                             form_fits<double, 4, false, false>()
                         ? 1
                         : 2;
    return known[dev] == 1;
}

// the cascade's state back into the biquad stage's own array: before anything but the fused
// kernel touches it
int Plan::export_state(hipStream_t s)
{
    Impl &I = *impl_;
    if (!I.in_slots)
        return PIPE_HIP_OK;
    PH_FOR_SECTIONS(I.S, hipLaunchKernelGGL(chain_state_export_kernel<SC>, dim3((unsigned)((I.own_series + 255) / 256)), dim3(256), 0, s,
                                            I.bq_state, static_cast<const unsigned long long *>(I.own.p), I.own_series, I.epoch + 1,
                                            I.own_C, I.own_pairs))
    PH_HIP(hipGetLastError());
    I.in_slots = false;
    return PIPE_HIP_OK;
}
// the stage's own array has been reset (StartFunc): what the slots hold is void
void Plan::drop_state() { impl_->in_slots = false; }

int Plan::rollback(hipStream_t s)
{
    Impl &I = *impl_;
    if (!I.in_slots)
        return PIPE_HIP_OK;
    // own_read takes the newer slot that is NOT tagged with the epoch it is given: given the failed launch's own
    // epoch it skips whatever that launch wrote
    PH_FOR_SECTIONS(I.S, hipLaunchKernelGGL(chain_state_export_kernel<SC>, dim3((unsigned)((I.own_series + 255) / 256)), dim3(256), 0, s,
                                            I.bq_state, static_cast<const unsigned long long *>(I.own.p), I.own_series, I.epoch, I.own_C,
                                            I.own_pairs))
    PH_HIP(hipGetLastError());
    I.in_slots = false;
    ++I.epoch;  // (the failed launch's tags stay behind in the records and slots: never reused)
    return PIPE_HIP_OK;
}

void Plan::debug_fail_next(int tile, double limit_us)
{
    impl_->debug_withhold = tile;
    impl_->debug_limit_us = limit_us;
}

template <typename T, int S, bool GENERAL, bool LOCAL>
static int launch_t(const ols::Plan::Impl &P, const void *d_in, void *d_out, const void *hist, Args32 a, const FuseArgs &fa,
                    const FuseConst<S> &fc, hipStream_t s, KernelTimer *timer)
{
    auto kfn = ols::fir_ols32_kernel<T, T, S, GENERAL, LOCAL>;
    const size_t lds = fused_lds_bytes<S, LOCAL>();
    PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    // every workgroup of the grid must be resident (tiles wait for their predecessors): one
    // 512-thread workgroup per CU, never more
    const int64_t resident = P.cus;
    const int64_t wanted = LOCAL ? a.lines : (a.nunits + kWaves32 - 1) / kWaves32;
    const unsigned grid = (unsigned)(wanted < resident ? wanted : resident);
    const int64_t stride = (int64_t)grid * kWaves32;
    a.d_slot = (int)(stride % a.upl);
    a.d_line = (int)(stride / a.upl);
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    if (timer)
        PH_TRY(timer->pair(&ev_a, &ev_b));
    // Tiles of the global look-back wait for tiles of OTHER workgroups of the same launch, so the
    // whole grid must be resident at once.  Two such launches on two streams could each get half the
    // CUs and wait for their other halves: launches of this form on one device therefore run one
    // after the other (an event chain; the block-local form has no such waits and is not chained).
    static std::mutex chain_mu;
    static hipEvent_t chain_done[64] = {};
    std::unique_lock<std::mutex> chain_lock(chain_mu, std::defer_lock);
    int dev = 0;
    if constexpr (!LOCAL) {
        PH_HIP(hipGetDevice(&dev));
        dev &= 63;
        chain_lock.lock();  // wait + launch + record as one step: no other launch of this form slips in between
        if (!chain_done[dev])
            PH_HIP(hipEventCreateWithFlags(&chain_done[dev], hipEventDisableTiming));
        else
            PH_HIP(hipStreamWaitEvent(s, chain_done[dev], 0));
    }
    hipExtLaunchKernelGGL(kfn, dim3(grid), dim3(kWaves32 * 64), lds, s, ev_a, ev_b, 0, static_cast<const T *>(d_in),
                          static_cast<T *>(d_out), static_cast<const T *>(hist), static_cast<const double2 *>(P.tw32.p),
                          static_cast<const double2 *>(P.hperm[P.cur].p), a, fa, fc);
    PH_HIP(hipGetLastError());
    if constexpr (!LOCAL)
        PH_HIP(hipEventRecord(chain_done[dev], s));
#ifdef PH_FUSE_PROF
    {
        static int launches = 0;
        if (++launches == 30 && fa.prof) {
            PH_HIP(hipStreamSynchronize(s));
            std::vector<unsigned long long> h((size_t)ols::kFuseProfPhases * kWaves32 * grid);
            PH_HIP(hipMemcpy(h.data(), fa.prof, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
            static const char *names[ols::kFuseProfPhases] = {
                "window + FIR transform", "to segments + pass 1", "scan", "publish A", "look-back", "publish P",
                "pass 3", "back to natural", "last-tile pass 3 + back", "stores", "lookback: own state", "lookback: wait", "-"};
            double sum[ols::kFuseProfPhases] = {}, tot = 0;
            for (size_t w = 0; w < (size_t)kWaves32 * grid; ++w)
                for (int i = 0; i < ols::kFuseProfPhases; ++i)
                    sum[i] += (double)h[w * ols::kFuseProfPhases + i];
            for (double v : sum)
                tot += v;
            std::fprintf(stderr, "[fused prof] s_memtime ticks per unit (wave time), %lld units, grid %u\n",
                         (long long)a.nunits, grid);
            for (int i = 0; i < ols::kFuseProfPhases; ++i)
                std::fprintf(stderr, "[fused prof]   %-26s %9.1f  %5.1f %%\n", names[i], sum[i] / (double)a.nunits,
                             100.0 * sum[i] / tot);
            std::fprintf(stderr, "[fused prof]   %-26s %9.1f\n", "total", tot / (double)a.nunits);
            // per wave index of the workgroup (averaged over the workgroups): who waits for whom
            for (int w = 0; w < kWaves32; ++w) {
                double ph[ols::kFuseProfPhases] = {};
                for (unsigned b = 0; b < grid; ++b)
                    for (int i = 0; i < ols::kFuseProfPhases; ++i)
                        ph[i] += (double)h[((size_t)b * kWaves32 + w) * ols::kFuseProfPhases + i];
                double t = 0;
                for (double v : ph)
                    t += v;
                std::fprintf(stderr, "[fused prof]   wave %d: transform %8.0f  segments %6.0f  wait %7.0f  pass3 %6.0f  stores %6.0f  total %8.0f\n", w,
                             ph[0] / grid, ph[1] / grid, ph[11] / grid, ph[6] / grid, ph[9] / grid, t / grid);
            }
#ifdef PH_FUSE_TIMELINE
            // timeline of two workgroups: ticks since the workgroup's first event
            std::vector<unsigned long long> tl((size_t)ols::kTlEvents * ols::kTlUnits * kWaves32 * grid);
            PH_HIP(hipMemcpy(tl.data(), fa.prof + ols::kTlOffset, sizeof(unsigned long long) * tl.size(), hipMemcpyDeviceToHost));
            for (unsigned b : {0u, grid / 2}) {
                const unsigned long long *row = tl.data() + (size_t)b * kWaves32 * ols::kTlUnits * ols::kTlEvents;
                unsigned long long t0 = ~0ull;
                for (int i = 0; i < kWaves32 * ols::kTlUnits * ols::kTlEvents; ++i)
                    if (row[i] && row[i] < t0)
                        t0 = row[i];
                std::fprintf(stderr, "[fused prof] timeline of workgroup %u (gate | start | transform done | epilogue done | stores issued):\n", b);
                for (int w = 0; w < kWaves32; ++w) {
                    std::fprintf(stderr, "[fused prof]   wave %d:", w);
                    for (int u = 0; u < ols::kTlUnits; ++u) {
                        const unsigned long long *e = row + ((size_t)w * ols::kTlUnits + u) * ols::kTlEvents;
                        if (!e[1])
                            continue;
                        std::fprintf(stderr, "  [%6lld %6lld %6lld %6lld %6lld]", (long long)(e[0] - t0), (long long)(e[1] - t0),
                                     (long long)(e[2] - t0), (long long)(e[3] - t0), (long long)(e[4] - t0));
                    }
                    std::fprintf(stderr, "\n");
                }
            }
#endif
        }
    }
#endif
    return PIPE_HIP_OK;
}

// (the stream's element type: float32, or float64 on a handle switched with PIPE_HIP_PARAM_RELAXED_F64)
template <int S, bool GENERAL, bool LOCAL>
static int launch(bool f64, const ols::Plan::Impl &P, const void *d_in, void *d_out, const void *hist, const Args32 &a,
                  const FuseArgs &fa, const FuseConst<S> &fc, hipStream_t s, KernelTimer *timer)
{
    return f64 ? launch_t<double, S, GENERAL, LOCAL>(P, d_in, d_out, hist, a, fa, fc, s, timer)
               : launch_t<float, S, GENERAL, LOCAL>(P, d_in, d_out, hist, a, fa, fc, s, timer);
}

int Plan::run(const pipe_hip_processor::FirFuseView &fir, const pipe_hip_processor::BiquadFuseView &bq, bool has_gain,
              double gain, const void *d_in, void *d_out, bool f64, int64_t frames, int channels, int lines, hipStream_t s,
              KernelTimer *timer, const char **kernel_name)
{
    Impl &I = *impl_;
    const ols::Plan::Impl &P = *static_cast<const ols::Plan::Impl *>(fir.plan);
    const int S = bq.sections;
    PH_TRY(prepare(bq.coeffs, S, fir.ntaps, s));
    Args32 a{};
    a.frames = frames;
    a.hist_new = fir.hist_new;
    a.line_stride = frames * channels;
    a.C = channels;
    a.N = fir.ntaps;
    a.H = fir.ntaps - 1;
    a.HP = (a.H + 31) / 32 * 32;  // a tile's first output opens a segment (ols32_kernel.hpp)
    a.L = ols::kM32 - a.HP;
    a.pairs = (channels + 1) / 2;  // (an odd count: the last channel alone in its pair, as in the FIR alone)
    a.odd = channels & 1;
    a.lines = lines;
    a.tiles_per_line = (int)((frames + a.L - 1) / a.L);
    a.ipl = a.tiles_per_line * a.pairs;
    a.upl = (a.ipl + 1) / 2;
    a.nunits = (int64_t)a.upl * lines;
    const int N2 = 2 * S, NV = 2 * N2;
    // records: [series][tile][A | P][2 NV granules]; tags are the launch epoch, so nothing is
    // cleared between launches (a fresh or regrown array is zeroed once; epochs start at 1)
    const size_t need = (size_t)lines * a.pairs * a.tiles_per_line * (size_t)(2 * 2 * NV);
    if (I.rec_granules < need) {
        PH_HIP(hipStreamSynchronize(s));
        PH_TRY(I.rec.alloc(sizeof(unsigned long long) * need));
        PH_HIP(hipMemsetAsync(I.rec.p, 0, sizeof(unsigned long long) * need, s));
        I.rec_granules = need;
    }
    if (I.epoch >= 0xFFFFFFF0u) {  // 2^32 launches: tags would repeat
        PH_TRY(export_state(s));
        PH_HIP(hipMemsetAsync(I.rec.p, 0, sizeof(unsigned long long) * I.rec_granules, s));
        I.epoch = 0;
    }
    // the cascade's state: in the tagged slots while the chain stays fused (ols32_kernel.hpp)
    const int nseries2 = lines * a.pairs;
    if (I.own_series != nseries2 || I.own_C != channels) {
        PH_TRY(export_state(s));
        PH_HIP(hipStreamSynchronize(s));
        PH_TRY(I.own.alloc(sizeof(unsigned long long) * (size_t)nseries2 * (2 * 2 * NV)));
        I.own_series = nseries2;
        I.own_C = channels;
        I.own_pairs = a.pairs;
    }
    I.bq_state = bq.state;
    if (!I.in_slots) {
        ++I.epoch;
        PH_FOR_SECTIONS(S, hipLaunchKernelGGL(chain_state_import_kernel<SC>, dim3((unsigned)((nseries2 + 255) / 256)), dim3(256), 0, s,
                                              static_cast<const double *>(bq.state), static_cast<unsigned long long *>(I.own.p),
                                              nseries2, I.epoch, channels, a.pairs))
        PH_HIP(hipGetLastError());
        I.in_slots = true;
    }
    ++I.epoch;
    FuseArgs fa{};
    fa.k0 = a.HP / 32;
    fa.epoch = I.epoch;
    fa.rec = static_cast<unsigned long long *>(I.rec.p);
    fa.own = static_cast<unsigned long long *>(I.own.p);
    const size_t seg_bytes = sizeof(double) * (size_t)lines * a.pairs * NV;
    if (I.seg_state.bytes < seg_bytes)
        PH_TRY(I.seg_state.alloc(seg_bytes));
    fa.seg_state = static_cast<double *>(I.seg_state.p);
    fa.mats = static_cast<const double *>(I.mats[I.cur_mats].p);
    fa.err = I.err_dev;
    // a wait for a predecessor's record gives up after ~4 s of the shader clock (2^33 ticks): a predecessor that a
    // context switch took away is back long before that; the caller then runs the call again on the staged chain
    fa.spin_ticks = 1ull << 33;
    fa.withhold = -1;
    if (I.debug_withhold >= 0) {
        fa.withhold = I.debug_withhold;
        fa.spin_ticks = (unsigned long long)(I.debug_limit_us * 2000.0);  // (~2 ticks a nanosecond)
        I.debug_withhold = -1;
    }
#ifdef PH_FUSE_PROF
    if (!I.prof.p)
        PH_TRY(I.prof.alloc(sizeof(unsigned long long) * (ols::kTlOffset + (size_t)ols::kTlEvents * ols::kTlUnits * kWaves32 * 4096)));
    fa.prof = static_cast<unsigned long long *>(I.prof.p);
#ifdef PH_FUSE_TIMELINE
    PH_HIP(hipMemsetAsync(fa.prof + ols::kTlOffset, 0, sizeof(unsigned long long) * ols::kTlEvents * ols::kTlUnits * kWaves32 * 4096, s));
#endif
#endif
    I.err_checked = false;
    I.last_stream = s;
    // a filter that forgets within one look-back window (D <= 32) takes the kernel without P
    // records and windows; PIPE_HIP_CHAIN_GENERAL=1 forces the general one (tests)
    static const bool force_general = PH_ENV_AB("PIPE_HIP_CHAIN_GENERAL") != nullptr;
    const bool general = force_general || I.D > 32;
    if (S < 1 || S > kMaxFusedSections || (S >= 2 && general))
        return PIPE_HIP_EINVAL;  // (Plan::accepts said otherwise: the caller did not ask)
    I.c1.gain = I.c2.gain = I.c3.gain = I.c4.gain = has_gain ? gain : 1.0;
    // Block-local look-back (ols32_kernel.hpp): at least as many Lines as CUs -- a workgroup per CU,
    // whole Lines per workgroup, balanced to within one Line -- and predecessors within the record
    // ring's reach.  PIPE_HIP_CHAIN_LOCAL=0 switches it off (tests, A/B).
    const char *local_env = PH_ENV_AB("PIPE_HIP_CHAIN_LOCAL");
    // (three and four sections: a ring of records per section does not fit the LDS next to the tap spectrum -- the global look-back)
    const bool local = !general && S <= 2 && !(local_env && local_env[0] == '0') && lines >= P.cus &&
                       (int64_t)I.D * a.pairs <= (S == 2 ? ols::kLocalReach2 : ols::kLocalReach) &&
                       (lines % P.cus == 0 || lines >= 8 * P.cus);
    a.local = local ? 1 : 0;
    {
        const char *e = PH_ENV_AB("PIPE_HIP_CHAIN_STAGGER");  // A/B knob
        a.stagger = e ? std::atoi(e) : 0;
    }
    if (S == 3) {
        I.c3.D = I.D;
        *kernel_name = f64 ? "chain_fused_kernel<f64,f64,fir+biquad3+gain>" : "chain_fused_kernel<f32,f32,fir+biquad3+gain>";
        PH_TRY((launch<3, false, false>(f64, P, d_in, d_out, fir.hist, a, fa, I.c3, s, timer)));
    } else if (S == 4) {
        I.c4.D = I.D;
        *kernel_name = f64 ? "chain_fused_kernel<f64,f64,fir+biquad4+gain>" : "chain_fused_kernel<f32,f32,fir+biquad4+gain>";
        PH_TRY((launch<4, false, false>(f64, P, d_in, d_out, fir.hist, a, fa, I.c4, s, timer)));
    } else
    if (S == 2) {
        // two sections: the sections one after the other over the tile in segment layout (ols32_kernel.hpp)
        I.c2.D = I.D;
        if (local) {
            *kernel_name = f64 ? "chain_fused_kernel<f64,f64,fir+biquad2+gain,local>" : "chain_fused_kernel<f32,f32,fir+biquad2+gain,local>";
            PH_TRY((launch<2, false, true>(f64, P, d_in, d_out, fir.hist, a, fa, I.c2, s, timer)));
        } else {
            *kernel_name = f64 ? "chain_fused_kernel<f64,f64,fir+biquad2+gain>" : "chain_fused_kernel<f32,f32,fir+biquad2+gain>";
            PH_TRY((launch<2, false, false>(f64, P, d_in, d_out, fir.hist, a, fa, I.c2, s, timer)));
        }
    } else
    if (general) {
        I.c1.D = force_general && I.D <= 32 ? I.D : (1 << 30);
        *kernel_name = f64 ? "chain_fused_kernel<f64,f64,fir+biquad1+gain,general>" : "chain_fused_kernel<f32,f32,fir+biquad1+gain,general>";
        PH_TRY((launch<1, true, false>(f64, P, d_in, d_out, fir.hist, a, fa, I.c1, s, timer)));
    } else if (local) {
        I.c1.D = I.D;
        *kernel_name = f64 ? "chain_fused_kernel<f64,f64,fir+biquad1+gain,local>" : "chain_fused_kernel<f32,f32,fir+biquad1+gain,local>";
        PH_TRY((launch<1, false, true>(f64, P, d_in, d_out, fir.hist, a, fa, I.c1, s, timer)));
    } else {
        I.c1.D = I.D;
        *kernel_name = f64 ? "chain_fused_kernel<f64,f64,fir+biquad1+gain>" : "chain_fused_kernel<f32,f32,fir+biquad1+gain>";
        PH_TRY((launch<1, false, false>(f64, P, d_in, d_out, fir.hist, a, fa, I.c1, s, timer)));
    }
    // the state after every Line's last frame: the fused kernel's own work when the Line ends on a
    // segment boundary (PIPE_HIP_CHAIN_NO_TAIL: debug switch)
    if (frames % 32 != 0 && !PH_ENV_AB("PIPE_HIP_CHAIN_NO_TAIL")) {
        TailArgs ta{};
        ta.frames = frames;
        ta.line_stride = a.line_stride;
        ta.C = channels;
        ta.pairs = a.pairs;
        ta.N = a.N;
        ta.H = a.H;
        ta.HP = a.HP;
        ta.L = a.L;
        ta.tiles_per_line = a.tiles_per_line;
        ta.nseries = lines * channels;
        ta.seg_state = static_cast<const double *>(I.seg_state.p);
        ta.own = static_cast<unsigned long long *>(I.own.p);
        ta.epoch = I.epoch;
        const unsigned tgrid = (unsigned)((ta.nseries + kTailSeries - 1) / kTailSeries);
        const size_t tlds = sizeof(double) * (size_t)(32 + a.H) * kTailSeries;
        if (f64) {
            PH_FOR_SECTIONS(S, hipLaunchKernelGGL((chain_tail_kernel<SC, double>), dim3(tgrid), dim3(32 * kTailSeries), tlds, s,
                                                  static_cast<const double *>(d_in), static_cast<const double *>(fir.hist), fir.taps, ta,
                                                  I.fc<SC>()))
        } else {
            PH_FOR_SECTIONS(S, hipLaunchKernelGGL((chain_tail_kernel<SC, float>), dim3(tgrid), dim3(32 * kTailSeries), tlds, s,
                                                  static_cast<const float *>(d_in), static_cast<const float *>(fir.hist), fir.taps, ta,
                                                  I.fc<SC>()))
        }
        PH_HIP(hipGetLastError());
    }
    return PIPE_HIP_OK;
}

}  // namespace fused
}  // namespace pipehip
