// FIR -> biquad -> gain as ONE kernel: the epilogue of ols32_kernel.hpp behind the 32 x 32
// overlap-save transform.  One read of the float32 input, one write of the float32 result; the
// float64 intermediates of the staged chain (chain.hip: 4x the algorithmic traffic) never exist.
//
// This file is the host side: the matrices the epilogue needs (zero-input state transitions of the
// biquad cascade, in long double), the look-back records, and the launch.
#include <cmath>
#include <cstdlib>
#include <cstring>
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
template <int N>
void store(double (&dst)[N][N], const Mat &m)
{
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
            dst[i][j] = (double)m.m[i][j];
}
void store_flat(double *dst, const Mat &m)
{
    for (int i = 0; i < m.n; ++i)
        for (int j = 0; j < m.n; ++j)
            dst[i * m.n + j] = (double)m.m[i][j];
}

}  // namespace

struct Plan::Impl {
    DevBuf rec, tj, pk, state_out, err;
    PinnedBuf h_tab;  // staging of the two tables
    std::vector<double> coeffs;
    int S = 0, H = -1;
    bool has_gain = false;
    double gain = 1.0;
    FuseConst<1> c1{};
    FuseConst<2> c2{};
    int D = 1 << 30;
    unsigned epoch = 0;
    size_t rec_granules = 0;
    bool err_checked = true;
    hipStream_t last_stream = nullptr;
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

template <int S>
static void fill_const(FuseConst<S> *fc, const double *coeffs, const Mat &M, int L, const Mat &ML, const Mat &T32, int D)
{
    std::memcpy(fc->c, coeffs, sizeof(double) * 5 * S);
    for (int i = 0; i < 5; ++i)
        store(fc->A[i], power(M, 32L << i));
    (void)L;
    store(fc->ML, ML);
    store(fc->T32, T32);
    fc->D = D;
}

// (re)build everything that depends on the coefficients or the tap count
int Plan::prepare(const double *coeffs, int S, int ntaps, hipStream_t s)
{
    Impl &I = *impl_;
    const int H = ntaps - 1, L = ols::kM32 - H;
    if (I.S == S && I.H == H && I.coeffs.size() == (size_t)5 * S &&
        std::memcmp(I.coeffs.data(), coeffs, sizeof(double) * 5 * S) == 0)
        return PIPE_HIP_OK;
    const int n = 2 * S;
    const Mat M = one_step(coeffs, S);
    const Mat ML = power(M, L);
    std::vector<Mat> T(33);
    T[0] = identity(n);
    for (int j = 1; j <= 32; ++j)
        T[j] = mul(T[j - 1], ML);
    constexpr int kNever = 1 << 30;  // the filter does not forget within a look-back window
    int D = kNever;
    for (int j = 1; j <= 32 && D == kNever; ++j) {
        ld big = 0;
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < n; ++k)
                big = std::fmax(big, std::fabs(T[j].m[i][k]));
        if (big < 0x1p-90L)
            D = j;
    }
    I.D = D;
    if (S == 1)
        fill_const<1>(&I.c1, coeffs, M, L, ML, T[32], D);
    else
        fill_const<2>(&I.c2, coeffs, M, L, ML, T[32], D);
    // tables: Tj [33][n][n], Pk [32][n][n]
    const size_t tj_n = (size_t)33 * n * n, pk_n = (size_t)32 * n * n;
    if (!I.tj.p) {
        PH_TRY(I.tj.alloc(sizeof(double) * 33 * kMaxN2 * kMaxN2));
        PH_TRY(I.pk.alloc(sizeof(double) * 32 * kMaxN2 * kMaxN2));
        PH_TRY(I.h_tab.alloc(sizeof(double) * 65 * kMaxN2 * kMaxN2));
        PH_TRY(I.err.alloc(sizeof(int)));
        PH_HIP(hipMemsetAsync(I.err.p, 0, sizeof(int), s));
    } else {
        // the staging block may still be in flight for an earlier upload
        PH_HIP(hipStreamSynchronize(s));
    }
    double *h = static_cast<double *>(I.h_tab.p);
    for (int j = 0; j <= 32; ++j)
        store_flat(h + (size_t)j * n * n, T[j]);
    const int k0 = H / 32;
    for (int k = 0; k < 32; ++k)
        store_flat(h + tj_n + (size_t)k * n * n, k > k0 ? power(M, 32L * k - H) : identity(n));
    PH_HIP(hipMemcpyAsync(I.tj.p, h, sizeof(double) * tj_n, hipMemcpyHostToDevice, s));
    PH_HIP(hipMemcpyAsync(I.pk.p, h + tj_n, sizeof(double) * pk_n, hipMemcpyHostToDevice, s));
    I.coeffs.assign(coeffs, coeffs + 5 * S);
    I.S = S;
    I.H = H;
    return PIPE_HIP_OK;
}

int Plan::poll_error(hipStream_t s)
{
    Impl &I = *impl_;
    if (!I.err.p || I.err_checked)
        return PIPE_HIP_OK;
    if (I.last_stream)
        s = I.last_stream;  // the stream the last launch went to
    int e = 0;
    PH_HIP(hipMemcpyAsync(&e, I.err.p, sizeof(int), hipMemcpyDeviceToHost, s));
    PH_HIP(hipStreamSynchronize(s));
    I.err_checked = true;
    if (e != 0) {
        PH_HIP(hipMemsetAsync(I.err.p, 0, sizeof(int), s));
        return PIPE_HIP_EHIP;
    }
    return PIPE_HIP_OK;
}

template <int S>
static int launch(const ols::Plan::Impl &P, const void *d_in, void *d_out, const double *hist, Args32 a, const FuseArgs &fa,
                  const FuseConst<S> &fc, hipStream_t s, KernelTimer *timer)
{
    auto kfn = ols::fir_ols32_kernel<float, float, S>;
    const size_t lds =
        sizeof(double2) * (ols::kHalf32 + 1 + 31 * 32) + sizeof(double) * (size_t)ols::kPlane32 * 2 * kWaves32;
    PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    // every workgroup of the grid must be resident (tiles wait for their predecessors): one
    // 512-thread workgroup per CU, never more
    const int64_t resident = P.cus;
    const int64_t wanted = (a.nunits + kWaves32 - 1) / kWaves32;
    const unsigned grid = (unsigned)(wanted < resident ? wanted : resident);
    const int64_t stride = (int64_t)grid * kWaves32;
    a.d_slot = (int)(stride % a.upl);
    a.d_line = (int)(stride / a.upl);
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    if (timer)
        PH_TRY(timer->pair(&ev_a, &ev_b));
    hipExtLaunchKernelGGL(kfn, dim3(grid), dim3(kWaves32 * 64), lds, s, ev_a, ev_b, 0, static_cast<const float *>(d_in),
                          static_cast<float *>(d_out), hist, static_cast<const double2 *>(P.tw32.p),
                          static_cast<const double2 *>(P.hperm[P.cur].p), a, fa, fc);
    PH_HIP(hipGetLastError());
    return PIPE_HIP_OK;
}

int Plan::run(const pipe_hip_processor::FirFuseView &fir, const pipe_hip_processor::BiquadFuseView &bq, bool has_gain,
              double gain, const void *d_in, void *d_out, int64_t frames, int channels, int lines, hipStream_t s,
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
    a.L = ols::kM32 - a.H;
    a.pairs = channels / 2;
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
    const size_t state_bytes = sizeof(double) * (size_t)lines * channels * N2;
    if (I.state_out.bytes < state_bytes)
        PH_TRY(I.state_out.alloc(state_bytes));
    if (++I.epoch == 0) {  // 2^32 launches: tags would repeat
        PH_HIP(hipMemsetAsync(I.rec.p, 0, sizeof(unsigned long long) * I.rec_granules, s));
        I.epoch = 1;
    }
    FuseArgs fa{};
    fa.k0 = a.H / 32;
    fa.n00 = a.H % 32;
    fa.epoch = I.epoch;
    fa.rec = static_cast<unsigned long long *>(I.rec.p);
    fa.state = bq.state;
    fa.state_out = static_cast<double *>(I.state_out.p);
    fa.Tj = static_cast<const double *>(I.tj.p);
    fa.Pk = static_cast<const double *>(I.pk.p);
    fa.err = static_cast<int *>(I.err.p);
    I.err_checked = false;
    I.last_stream = s;
    if (S == 1) {
        I.c1.has_gain = has_gain ? 1 : 0;
        I.c1.gain = gain;
        *kernel_name = "chain_fused_kernel<f32,f32,fir+biquad1+gain>";
        PH_TRY(launch<1>(P, d_in, d_out, fir.hist, a, fa, I.c1, s, timer));
    } else {
        I.c2.has_gain = has_gain ? 1 : 0;
        I.c2.gain = gain;
        *kernel_name = "chain_fused_kernel<f32,f32,fir+biquad2+gain>";
        PH_TRY(launch<2>(P, d_in, d_out, fir.hist, a, fa, I.c2, s, timer));
    }
    // the biquad stage's own state <- the state after this call (stream-ordered)
    PH_HIP(hipMemcpyAsync(bq.state, I.state_out.p, state_bytes, hipMemcpyDeviceToDevice, s));
    return PIPE_HIP_OK;
}

}  // namespace fused
}  // namespace pipehip
