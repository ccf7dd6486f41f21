// DF2T biquad cascade Processor for gfx950.
//
// Contract (oracle/dsp_oracle.h), per section, binary64:
//     y  = fma(b0, x, s1)
//     s1 = fma(-a1, y, fma(b1, x, s2))
//     s2 = fma(-a2, y, b2 * x)
// An IIR is a serial recurrence along time, so the only parallelism that keeps
// the float64 result bit-exact is across (Line, channel) series: one lane per
// series, state in registers for the whole call.  The kernel is bound by the
// latency of the dependent fma chain (2 fma on the critical path per section and
// sample), not by HBM: what matters is that memory never adds to that chain, so
// each lane issues kChunk independent loads before it starts the kChunk dependent
// steps, and stores the results afterwards.  The channels of a Line are adjacent
// in memory, so a wave's accesses are C-element runs that L2 merges into lines.
// An optional gain (the stage that follows a biquad in BASELINE config 3) is
// applied to the float64 result in the same pass: y_out = y * g, exactly the
// arithmetic of a separate gain stage reading float64.
#include "common.hpp"

namespace pipehip {
namespace {

constexpr int kMaxSections = 8;
constexpr int kThreads = 64;
constexpr int kChunk = 32;  // frames per chunk; two chunks in flight per lane

struct BiquadCoeffs {
    double c[kMaxSections][5];
};

struct BiquadArgs {
    double *state;  // [lines][C][S][2]
    int64_t frames;
    int C, S, nseries;
    double gain;
    int64_t in_bytes, out_bytes;  // extent of the call's buffers (< 4 GiB)
};

template <int NS>
__device__ __forceinline__ double biquad_step(double x, double (&s1)[kMaxSections],
                                              double (&s2)[kMaxSections], const BiquadCoeffs &q)
{
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const double y = __builtin_fma(q.c[s][0], x, s1[s]);
        const double t = __builtin_fma(q.c[s][1], x, s2[s]);
        s1[s] = __builtin_fma(-q.c[s][3], y, t);
        const double u = q.c[s][2] * x;
        s2[s] = __builtin_fma(-q.c[s][4], y, u);
        x = y;
    }
    return x;
}

// Element access through a buffer resource: the per-lane part of the address (which
// series) is a 32-bit VGPR offset computed once, the per-frame part is a wave-uniform SGPR
// offset -- so the dependent fma chain shares the lane's single VALU issue slot (one
// instruction per ~4.4 cycles for a lone wave) with no address arithmetic at all.
template <typename T>
__device__ __forceinline__ T buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff);
template <>
__device__ __forceinline__ float buf_load<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <>
__device__ __forceinline__ double buf_load<double>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double v)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, voff, soff, 0);
}

// NS = compile-time section count (1, 2) or 0 = runtime a.S; GAIN = a gain is folded in
template <typename TIn, typename TOut, int NS, bool GAIN>
__global__ void __launch_bounds__(kThreads)
biquad_kernel(const TIn *__restrict__ in_base, TOut *__restrict__ out_base, const BiquadArgs a,
              const BiquadCoeffs q)
{
    const int sid = blockIdx.x * kThreads + threadIdx.x;
    const bool live = sid < a.nseries;
    const int sidc = live ? sid : 0;
    const int line = sidc / a.C;
    const int c = sidc - line * a.C;
    double *__restrict__ st = a.state + (int64_t)sidc * a.S * 2;
    // whole call through 32-bit offsets (the launcher guarantees the buffers are < 4 GiB)
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TIn *>(in_base), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out_base, 0, (int)a.out_bytes, 0x00020000);
    // dead lanes point past the end: their loads return 0 and their stores are dropped
    const unsigned vin = live ? (unsigned)(((int64_t)line * a.frames * a.C + c) * sizeof(TIn)) : 0xFFFFFFFFu;
    const unsigned vout = live ? (unsigned)(((int64_t)line * a.frames * a.C + c) * sizeof(TOut)) : 0xFFFFFFFFu;
    const unsigned sin_step = (unsigned)(a.C * sizeof(TIn));    // bytes per frame
    const unsigned sout_step = (unsigned)(a.C * sizeof(TOut));

    double s1[kMaxSections], s2[kMaxSections];
#pragma unroll
    for (int s = 0; s < kMaxSections; ++s) {
        s1[s] = 0.0;
        s2[s] = 0.0;
        if (live && s < a.S) {
            s1[s] = st[2 * s];
            s2[s] = st[2 * s + 1];
        }
    }
    auto step = [&](double x) -> double {
        double y;
        if constexpr (NS > 0) {
            y = biquad_step<NS>(x, s1, s2, q);
        } else {
            y = x;
#pragma unroll
            for (int s = 0; s < kMaxSections; ++s) {
                if (s < a.S) {
                    const double v = __builtin_fma(q.c[s][0], y, s1[s]);
                    const double t = __builtin_fma(q.c[s][1], y, s2[s]);
                    s1[s] = __builtin_fma(-q.c[s][3], v, t);
                    const double u = q.c[s][2] * y;
                    s2[s] = __builtin_fma(-q.c[s][4], v, u);
                    y = v;
                }
            }
        }
        if constexpr (GAIN)
            y = y * a.gain;
        return y;
    };

    // two chunks in flight: chunk k+1 is loading while the dependent chain walks chunk k
    // (series are scarce -- 64 waves for 4096 series -- so no other wave hides the latency)
    const int64_t nchunks = a.frames / kChunk;
    TIn xa[kChunk], xb[kChunk];
    if (nchunks > 0) {
#pragma unroll
        for (int u = 0; u < kChunk; ++u)
            xa[u] = buf_load<TIn>(rin, vin, (unsigned)u * sin_step);
    }
    auto run_chunk = [&](const TIn (&x)[kChunk], int64_t f0) {
        TOut y[kChunk];
#pragma unroll
        for (int u = 0; u < kChunk; ++u)
            y[u] = (TOut)step((double)x[u]);
        const unsigned so = (unsigned)f0 * sout_step;
#pragma unroll
        for (int u = 0; u < kChunk; ++u)
            buf_store(rout, vout, so + (unsigned)u * sout_step, y[u]);
    };
    int64_t k = 0;
    for (; k + 2 <= nchunks; k += 2) {
        const unsigned sb = (unsigned)((k + 1) * kChunk) * sin_step;
#pragma unroll
        for (int u = 0; u < kChunk; ++u)
            xb[u] = buf_load<TIn>(rin, vin, sb + (unsigned)u * sin_step);
        run_chunk(xa, k * kChunk);
        if (k + 2 < nchunks) {
            const unsigned sa = (unsigned)((k + 2) * kChunk) * sin_step;
#pragma unroll
            for (int u = 0; u < kChunk; ++u)
                xa[u] = buf_load<TIn>(rin, vin, sa + (unsigned)u * sin_step);
        }
        run_chunk(xb, (k + 1) * kChunk);
    }
    if (k < nchunks)
        run_chunk(xa, k * kChunk);
    for (int64_t f = nchunks * kChunk; f < a.frames; ++f) {
        const double y = step((double)buf_load<TIn>(rin, vin, (unsigned)f * sin_step));
        buf_store(rout, vout, (unsigned)f * sout_step, (TOut)y);
    }

    if (live) {
#pragma unroll
        for (int s = 0; s < kMaxSections; ++s) {
            if (s < a.S) {
                st[2 * s] = s1[s];
                st[2 * s + 1] = s2[s];
            }
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
    // a gain stage that directly follows this biquad in a chain is folded into the
    // store of the result (same float64 arithmetic as the separate stage)
    void set_post_gain(bool on, double g)
    {
        has_gain_ = on;
        gain_ = g;
    }

    int run(const void *d_in, int in_dtype, void *d_out, int out_dtype, int64_t frames,
            hipStream_t s) override
    {
        if (frames <= 0)
            return PIPE_HIP_OK;
        BiquadArgs a{};
        a.state = static_cast<double *>(state_.p);
        a.frames = frames;
        a.C = cfg.channels;
        a.S = S_;
        a.nseries = cfg.lines * cfg.channels;
        a.gain = gain_;
        a.in_bytes = (int64_t)dtype_size(in_dtype) * frames * cfg.channels * cfg.lines;
        a.out_bytes = (int64_t)dtype_size(out_dtype) * frames * cfg.channels * cfg.lines;
        if (a.in_bytes >= ((int64_t)1 << 32) - 4096 || a.out_bytes >= ((int64_t)1 << 32) - 4096)
            return PIPE_HIP_EINVAL;  // 32-bit buffer offsets: split the call (never reached by buffer_size*max_batch in practice)
        const dim3 grid((unsigned)((a.nseries + kThreads - 1) / kThreads));
        PH_TRY(timer.begin(s));
#define PH_BQ2(TI, TO, G)                                                                           \
    do {                                                                                            \
        if (S_ == 1)                                                                                \
            hipLaunchKernelGGL((biquad_kernel<TI, TO, 1, G>), grid, dim3(kThreads), 0, s,           \
                               static_cast<const TI *>(d_in), static_cast<TO *>(d_out), a, q_);     \
        else if (S_ == 2)                                                                           \
            hipLaunchKernelGGL((biquad_kernel<TI, TO, 2, G>), grid, dim3(kThreads), 0, s,           \
                               static_cast<const TI *>(d_in), static_cast<TO *>(d_out), a, q_);     \
        else                                                                                        \
            hipLaunchKernelGGL((biquad_kernel<TI, TO, 0, G>), grid, dim3(kThreads), 0, s,           \
                               static_cast<const TI *>(d_in), static_cast<TO *>(d_out), a, q_);     \
    } while (0)
#define PH_BQ(TI, TO, NAME)                                                                         \
    do {                                                                                            \
        if (has_gain_)                                                                              \
            PH_BQ2(TI, TO, true);                                                                   \
        else                                                                                        \
            PH_BQ2(TI, TO, false);                                                                  \
        last_kernel = NAME;                                                                         \
    } while (0)
        if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32)
            PH_BQ(float, float, "biquad_kernel<f32,f32>");
        else if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F64)
            PH_BQ(double, double, "biquad_kernel<f64,f64>");
        else if (in_dtype == PIPE_HIP_F32)
            PH_BQ(float, double, "biquad_kernel<f32,f64>");
        else
            PH_BQ(double, float, "biquad_kernel<f64,f32>");
#undef PH_BQ
#undef PH_BQ2
        PH_HIP(hipGetLastError());
        PH_TRY(timer.end(s));
        return PIPE_HIP_OK;
    }

private:
    int S_ = 1;
    bool has_gain_ = false;
    double gain_ = 1.0;
    BiquadCoeffs q_{};
    DevBuf state_;
    size_t state_bytes_ = 0;
};

}  // namespace

bool biquad_set_post_gain(pipe_hip_processor *p, bool on, double g)
{
    auto *b = dynamic_cast<Biquad *>(p);
    if (!b)
        return false;
    b->set_post_gain(on, g);
    return true;
}

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
