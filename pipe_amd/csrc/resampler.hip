// Rational polyphase resampler Processor (up/down, T taps per phase) for gfx950.
//
// Contract (oracle/dsp_oracle.h), binary64: output m (counted from Start) reads
// input frame n = floor(m*down/up) with phase p = (m*down) mod up:
//     acc = +0.0;  for j = 0..T-1:  acc = fma(proto[p + j*up], x[n-j], acc)
// and is emitted as soon as frame n has been consumed, so a call that brings the
// total input to I frames emits outputs m < ceil(I*up/down).
//
// Every output is independent: one lane per (Line, output frame, channel).  The
// polyphase table (up*T doubles, 30 KiB at 160x24) and the input window are served
// by L1/L2; per scalar output the kernel moves 4-8 B of HBM and does T fma, so it
// is HBM/L2-bound, not ALU-bound.  An up-sampler emits more frames than it
// consumes, which ProcessFunc cannot express with full buffers:
// hence the explicit (in_frames, out_cap) -> out_frames ABI.
#include "common.hpp"

namespace pipehip {
namespace {

constexpr int kThreads = 256;

struct ResampleArgs {
    const void *in;
    void *out;
    const double *hist;   // [lines][T-1][C]
    const double *proto;  // [up*T]
    int64_t in_frames, out_frames, out_cap;
    int64_t in_total, out_total;  // consumed / produced before this call
    int C, T, up, down, lines;
};

template <typename TIn, typename TOut>
__global__ void __launch_bounds__(kThreads) resample_kernel(const ResampleArgs a)
{
    const int64_t per_line = a.out_frames * a.C;
    const int64_t total = per_line * a.lines;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    const TIn *__restrict__ in = reinterpret_cast<const TIn *>(a.in);
    TOut *__restrict__ out = reinterpret_cast<TOut *>(a.out);
    const int H = a.T - 1;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += stride) {
        const int line = (int)(e / per_line);
        const int64_t r = e - (int64_t)line * per_line;
        const int64_t i = r / a.C;
        const int c = (int)(r - i * a.C);
        const int64_t m = a.out_total + i;
        const int64_t t = m * a.down;
        const int64_t n = t / a.up - a.in_total;  // relative to this call's input
        const int p = (int)(t % a.up);
        const TIn *__restrict__ x = in + (int64_t)line * a.in_frames * a.C;
        const double *__restrict__ h = a.hist + (int64_t)line * H * a.C;
        double acc = 0.0;
        for (int j = 0; j < a.T; ++j) {
            const int64_t idx = n - j;
            const double v = idx >= 0 ? (double)x[idx * a.C + c] : h[(idx + H) * a.C + c];
            acc = __builtin_fma(a.proto[p + (int64_t)j * a.up], v, acc);
        }
        out[((int64_t)line * a.out_cap + i) * a.C + c] = (TOut)acc;
    }
}

template <typename TIn>
__global__ void resample_hist_kernel(const TIn *__restrict__ in, const double *__restrict__ hist_old,
                                     double *__restrict__ hist_new, int64_t frames, int H, int C)
{
    const int line = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * C)
        return;
    const int j = i / C;
    const int c = i - j * C;
    const int64_t s = frames - H + j;
    double v;
    if (s >= 0)
        v = (double)in[((int64_t)line * frames + s) * C + c];
    else
        v = hist_old[((int64_t)line * H + (s + H)) * C + c];
    hist_new[((int64_t)line * H + j) * C + c] = v;
}

class Resampler final : public pipe_hip_processor {
public:
    int init(const double *proto, int32_t T, int32_t up, int32_t down)
    {
        T_ = T;
        up_ = up;
        down_ = down;
        const size_t n = (size_t)up * (size_t)T;
        PH_TRY(proto_.alloc(sizeof(double) * n));
        PH_HIP(hipMemcpy(proto_.p, proto, sizeof(double) * n, hipMemcpyHostToDevice));
        hist_bytes_ = sizeof(double) * (size_t)cfg.lines * (size_t)(T - 1) * (size_t)cfg.channels;
        PH_TRY(hist_[0].alloc(hist_bytes_));
        PH_TRY(hist_[1].alloc(hist_bytes_));
        return start(stream);
    }
    void rate(int32_t *up, int32_t *down) const override
    {
        *up = up_;
        *down = down_;
    }
    int64_t out_frames_for(int64_t in_frames) const override
    {
        return ((in_total_ + in_frames) * up_ + down_ - 1) / down_ - out_total_;
    }
    int64_t max_out_frames(int64_t in_frames) const override
    {
        return (in_frames * up_ + down_ - 1) / down_ + 1;
    }
    int start(hipStream_t s) override
    {
        if (hist_bytes_)
            PH_HIP(hipMemsetAsync(hist_[cur_].p, 0, hist_bytes_, s));
        in_total_ = 0;
        out_total_ = 0;
        return PIPE_HIP_OK;
    }
    // the generic single-rate entry is not meaningful for a rate changer
    int run(const void *, int, void *, int, int64_t, hipStream_t) override { return PIPE_HIP_EINVAL; }

    bool fixed_rate() const override { return false; }
    int run_var(const void *d_in, int in_dtype, int64_t in_frames, void *d_out, int out_dtype,
                int64_t out_cap, int64_t *out_frames, hipStream_t s) override
    {
        if (in_frames < 0)
            return PIPE_HIP_EINVAL;
        const int64_t n_out = out_frames_for(in_frames);
        if (n_out > out_cap)
            return PIPE_HIP_ECAP;  // nothing consumed (pipe.go:437: out is bufferSize frames)
        if (out_frames)
            *out_frames = n_out;
        if (in_frames == 0)
            return PIPE_HIP_OK;
        ResampleArgs a{};
        a.in = d_in;
        a.out = d_out;
        a.hist = static_cast<const double *>(hist_[cur_].p);
        a.proto = static_cast<const double *>(proto_.p);
        a.in_frames = in_frames;
        a.out_frames = n_out;
        a.out_cap = out_cap;
        a.in_total = in_total_;
        a.out_total = out_total_;
        a.C = cfg.channels;
        a.T = T_;
        a.up = up_;
        a.down = down_;
        a.lines = cfg.lines;
        const int64_t total = n_out * cfg.channels * cfg.lines;
        if (total > 0) {
            int64_t b = (total + kThreads - 1) / kThreads;
            if (b > 4096)
                b = 4096;
            const dim3 grid((unsigned)b);
            PH_TRY(timer.begin(s));
            if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32) {
                hipLaunchKernelGGL((resample_kernel<float, float>), grid, dim3(kThreads), 0, s, a);
                last_kernel = "resample_kernel<f32,f32>";
            } else if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F64) {
                hipLaunchKernelGGL((resample_kernel<double, double>), grid, dim3(kThreads), 0, s, a);
                last_kernel = "resample_kernel<f64,f64>";
            } else if (in_dtype == PIPE_HIP_F32) {
                hipLaunchKernelGGL((resample_kernel<float, double>), grid, dim3(kThreads), 0, s, a);
                last_kernel = "resample_kernel<f32,f64>";
            } else {
                hipLaunchKernelGGL((resample_kernel<double, float>), grid, dim3(kThreads), 0, s, a);
                last_kernel = "resample_kernel<f64,f32>";
            }
            PH_HIP(hipGetLastError());
            PH_TRY(timer.end(s));
        }
        const int H = T_ - 1;
        if (H > 0) {
            const int n = H * cfg.channels;
            const dim3 hg((unsigned)((n + 255) / 256), (unsigned)cfg.lines);
            double *hn = static_cast<double *>(hist_[cur_ ^ 1].p);
            if (in_dtype == PIPE_HIP_F32)
                hipLaunchKernelGGL(resample_hist_kernel<float>, hg, dim3(256), 0, s,
                                   static_cast<const float *>(d_in), a.hist, hn, in_frames, H,
                                   cfg.channels);
            else
                hipLaunchKernelGGL(resample_hist_kernel<double>, hg, dim3(256), 0, s,
                                   static_cast<const double *>(d_in), a.hist, hn, in_frames, H,
                                   cfg.channels);
            PH_HIP(hipGetLastError());
            cur_ ^= 1;
        }
        in_total_ += in_frames;
        out_total_ += n_out;
        return PIPE_HIP_OK;
    }

private:
    int T_ = 1, up_ = 1, down_ = 1;
    DevBuf proto_;
    DevBuf hist_[2];
    size_t hist_bytes_ = 0;
    int cur_ = 0;
    int64_t in_total_ = 0, out_total_ = 0;
};

}  // namespace

int make_resampler(const pipe_hip_config *cfg, const double *proto, int32_t taps_per_phase,
                   int32_t up, int32_t down, pipe_hip_processor **out)
{
    if (!proto || taps_per_phase < 1 || taps_per_phase > 1024 || up < 1 || down < 1 || up > 4096 ||
        down > 4096)
        return PIPE_HIP_EINVAL;
    auto p = std::make_unique<Resampler>();
    PH_TRY(p->init_common(cfg));
    PH_TRY(p->init(proto, taps_per_phase, up, down));
    *out = p.release();
    return PIPE_HIP_OK;
}

}  // namespace pipehip
