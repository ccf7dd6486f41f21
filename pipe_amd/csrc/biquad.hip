// DF2T biquad cascade Processor for gfx950.
//
// Contract (oracle/dsp_oracle.h), per section, binary64:
//     y  = fma(b0, x, s1)
//     s1 = fma(-a1, y, fma(b1, x, s2))
//     s2 = fma(-a2, y, b2 * x)
// An IIR is a serial recurrence along time, so the only parallelism that keeps
// the float64 result bit-exact is across (Line, channel) series: one lane per
// series.  A wave stages kChunk frames of its series through LDS (coalesced
// global access, all lanes of the workgroup cooperating), then each lane walks
// its own recurrence out of LDS.  The kernel is latency-bound by the dependent
// fma chain (2 fma on the critical path per section and sample), not by HBM.
#include "common.hpp"

namespace pipehip {
namespace {

constexpr int kMaxSections = 8;
constexpr int kThreads = 64;  // one wave per workgroup: series are scarce, spread them over CUs
constexpr int kChunk = 64;    // frames staged per trip

struct BiquadCoeffs {
    double c[kMaxSections][5];
};

struct BiquadArgs {
    const void *in;
    void *out;
    double *state;  // [lines][C][S][2]
    int64_t frames;
    int C, S, lines;
    int spb;        // series per workgroup (<= kThreads), a multiple of C or a divisor arrangement
};

// Workgroup b owns series [b*spb, b*spb + spb) where series id = line*C + c.
// Because the layout is (line, frame, channel), the series of one Line are
// contiguous within a frame: a chunk of one Line is a dense [kChunk][C] block.
template <typename TIn, typename TOut>
__global__ void __launch_bounds__(kThreads) biquad_kernel(const BiquadArgs a, const BiquadCoeffs q)
{
    __shared__ double tile[kThreads * (kChunk + 1)];
    const int series0 = blockIdx.x * a.spb;
    const int nseries_total = a.lines * a.C;
    const int nser = min(a.spb, nseries_total - series0);
    const int lane = threadIdx.x;
    const bool owner = lane < nser;
    const int my = series0 + lane;
    const int my_line = owner ? my / a.C : 0;
    const int my_c = owner ? my - my_line * a.C : 0;

    double s1[kMaxSections], s2[kMaxSections];
#pragma unroll
    for (int s = 0; s < kMaxSections; ++s) {
        s1[s] = 0.0;
        s2[s] = 0.0;
    }
    if (owner) {
        for (int s = 0; s < a.S; ++s) {
            const double *st = a.state + (((int64_t)my_line * a.C + my_c) * a.S + s) * 2;
            s1[s] = st[0];
            s2[s] = st[1];
        }
    }
    const TIn *__restrict__ in = reinterpret_cast<const TIn *>(a.in);
    TOut *__restrict__ out = reinterpret_cast<TOut *>(a.out);

    for (int64_t f0 = 0; f0 < a.frames; f0 += kChunk) {
        const int nf = (int)min((int64_t)kChunk, a.frames - f0);
        // stage: element e -> (series j, frame f) with the channel index fastest
        // in memory: for a Line, [f][c] is dense
        for (int e = lane; e < nser * nf; e += kThreads) {
            // walk memory order: per line segment [nf][cseg]
            const int j = e % nser;       // series within the workgroup
            const int f = e / nser;
            const int sid = series0 + j;
            const int l = sid / a.C, c = sid - l * a.C;
            tile[j * (kChunk + 1) + f] = (double)in[((int64_t)l * a.frames + f0 + f) * a.C + c];
        }
        __syncthreads();
        if (owner) {
            double *row = tile + lane * (kChunk + 1);
            for (int f = 0; f < nf; ++f) {
                double x = row[f];
#pragma unroll
                for (int s = 0; s < kMaxSections; ++s) {
                    if (s < a.S) {
                        const double y = __builtin_fma(q.c[s][0], x, s1[s]);
                        const double t = __builtin_fma(q.c[s][1], x, s2[s]);
                        s1[s] = __builtin_fma(-q.c[s][3], y, t);
                        const double u = q.c[s][2] * x;
                        s2[s] = __builtin_fma(-q.c[s][4], y, u);
                        x = y;
                    }
                }
                row[f] = x;
            }
        }
        __syncthreads();
        for (int e = lane; e < nser * nf; e += kThreads) {
            const int j = e % nser;
            const int f = e / nser;
            const int sid = series0 + j;
            const int l = sid / a.C, c = sid - l * a.C;
            out[((int64_t)l * a.frames + f0 + f) * a.C + c] = (TOut)tile[j * (kChunk + 1) + f];
        }
        __syncthreads();
    }
    if (owner) {
        for (int s = 0; s < a.S; ++s) {
            double *st = a.state + (((int64_t)my_line * a.C + my_c) * a.S + s) * 2;
            st[0] = s1[s];
            st[1] = s2[s];
        }
    }
}

class Biquad final : public pipe_hip_processor {
public:
    int init(const double *coeffs, int32_t nsections)
    {
        S_ = nsections;
        std::memset(&q_, 0, sizeof q_);
        std::memcpy(q_.c, coeffs, sizeof(double) * 5u * (size_t)S_);
        state_bytes_ = sizeof(double) * (size_t)cfg.lines * (size_t)cfg.channels * (size_t)S_ * 2u;
        PH_TRY(state_.alloc(state_bytes_));
        return start(stream);
    }
    int start(hipStream_t s) override
    {
        PH_HIP(hipMemsetAsync(state_.p, 0, state_bytes_, s));
        return PIPE_HIP_OK;
    }
    int set_param(int32_t param, const double *values, int32_t count) override
    {
        if (param != PIPE_HIP_PARAM_COEFFS || count != 5 * S_ || !values)
            return PIPE_HIP_EINVAL;
        std::memcpy(q_.c, values, sizeof(double) * 5u * (size_t)S_);  // kernel argument
        return PIPE_HIP_OK;
    }
    int run(const void *d_in, int in_dtype, void *d_out, int out_dtype, int64_t frames,
            hipStream_t s) override
    {
        if (frames <= 0)
            return PIPE_HIP_OK;
        BiquadArgs a{};
        a.in = d_in;
        a.out = d_out;
        a.state = static_cast<double *>(state_.p);
        a.frames = frames;
        a.C = cfg.channels;
        a.S = S_;
        a.lines = cfg.lines;
        const int nseries = cfg.lines * cfg.channels;
        // spread series over the chip: aim at >= 256 workgroups before packing lanes
        int spb = (nseries + 255) / 256;
        if (spb < 1)
            spb = 1;
        if (spb > kThreads)
            spb = kThreads;
        a.spb = spb;
        const dim3 grid((unsigned)((nseries + spb - 1) / spb));
        PH_TRY(timer.begin(s));
        if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32) {
            hipLaunchKernelGGL((biquad_kernel<float, float>), grid, dim3(kThreads), 0, s, a, q_);
            last_kernel = "biquad_kernel<f32,f32>";
        } else if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F64) {
            hipLaunchKernelGGL((biquad_kernel<double, double>), grid, dim3(kThreads), 0, s, a, q_);
            last_kernel = "biquad_kernel<f64,f64>";
        } else if (in_dtype == PIPE_HIP_F32) {
            hipLaunchKernelGGL((biquad_kernel<float, double>), grid, dim3(kThreads), 0, s, a, q_);
            last_kernel = "biquad_kernel<f32,f64>";
        } else {
            hipLaunchKernelGGL((biquad_kernel<double, float>), grid, dim3(kThreads), 0, s, a, q_);
            last_kernel = "biquad_kernel<f64,f32>";
        }
        PH_HIP(hipGetLastError());
        PH_TRY(timer.end(s));
        return PIPE_HIP_OK;
    }

private:
    int S_ = 1;
    BiquadCoeffs q_{};
    DevBuf state_;
    size_t state_bytes_ = 0;
};

}  // namespace

int make_biquad(const pipe_hip_config *cfg, const double *coeffs, int32_t nsections,
                pipe_hip_processor **out)
{
    if (!coeffs || nsections < 1 || nsections > kMaxSections)
        return PIPE_HIP_EINVAL;
    auto p = std::make_unique<Biquad>();
    PH_TRY(p->init_common(cfg));
    PH_TRY(p->init(coeffs, nsections));
    *out = p.release();
    return PIPE_HIP_OK;
}

}  // namespace pipehip
