// HBM-bound element-wise Processors for gfx950: gain (also the pass-through copy
// of mock.Processor, mock/mock.go:147-154, at gain == 1), the n-input mix, and the
// SplitMix64 synthetic source used by the bench.
//
// Contract (oracle/dsp_oracle.h): y = (T_out)((double)x * g);  mix = ((a+b)+c)...
// in binary64.  One 16-byte vector per lane per trip, grid capped at 2048
// workgroups with a grid-stride loop.
#include <cstdlib>

#include <type_traits>

#include <hip/hip_ext.h>

#include "common.hpp"

namespace pipehip {
namespace {

constexpr int kThreads = 256;
// One vector per lane and as many workgroups as that takes (the kernels keep their grid-stride
// loops only for the > 2^31-vector case): the dispatcher hands out workgroups in address order, so
// the chip streams through DRAM pages front to back.  A persistent 2048-workgroup grid striding
// over the buffer touches 2048 pages at once and loses a quarter of the bandwidth on multi-GiB
// passes (4.9 vs 6.3 TB/s at 4 GiB in + 4 GiB out; equal at 64 MiB).
constexpr int kMaxBlocks = 0x7FFFFFFF;

template <typename T, int V>
struct alignas(sizeof(T) * V) Pack {
    T v[V];
};

// the same for pointers aligned to their element only (a view at an odd element offset): gfx950 loads and stores
// 16 bytes from any 4-byte address, so such streams keep the vector path (a straddled cache line costs less
// than four times the instructions: mix of the configs[4] size 0.0229 -> 0.0165 ms)
template <typename T, int V>
struct __attribute__((packed, aligned(alignof(T)))) PackU {
    T v[V];
};

// elements handled per lane per trip: 4 (16 B of f32, 32 B of f64)
constexpr int kPer = 4;

template <typename T>
using V4 = T __attribute__((ext_vector_type(4)));
// streaming (non-temporal) vector access: a pass over more data than the caches hold should not
// displace what they do hold
template <typename P, typename T>
__device__ __forceinline__ P nt_load4(const P *p)
{
    const V4<T> v = __builtin_nontemporal_load(reinterpret_cast<const V4<T> *>(p));
    return __builtin_bit_cast(P, v);
}
template <typename P, typename T>
__device__ __forceinline__ void nt_store4(const P &v, P *p)
{
    __builtin_nontemporal_store(__builtin_bit_cast(V4<T>, v), reinterpret_cast<V4<T> *>(p));
}

template <typename TIn, typename TOut>
__global__ void __launch_bounds__(kThreads) gain_kernel(const TIn *__restrict__ in,
                                                        TOut *__restrict__ out, int64_t n, double g,
                                                        int vec_ok)
{
    const int64_t nvec = n / kPer;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    if (!vec_ok) {  // pointers aligned to their element only: 16-byte accesses from 4- / 8-byte addresses (PackU)
        using UI = PackU<TIn, kPer>;
        using UO = PackU<TOut, kPer>;
        for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += stride) {
            const UI x = reinterpret_cast<const UI *>(in)[i];
            UO y;
#pragma unroll
            for (int k = 0; k < kPer; ++k)
                y.v[k] = (TOut)((double)x.v[k] * g);
            reinterpret_cast<UO *>(out)[i] = y;
        }
    }
    using PI = Pack<TIn, kPer>;
    using PO = Pack<TOut, kPer>;
    const PI *vin = reinterpret_cast<const PI *>(in);
    PO *vout = reinterpret_cast<PO *>(out);
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; vec_ok && i < nvec; i += stride) {
        const PI x = vec_ok > 1 ? nt_load4<PI, TIn>(vin + i) : vin[i];
        PO y;
#pragma unroll
        for (int k = 0; k < kPer; ++k)
            y.v[k] = (TOut)((double)x.v[k] * g);
        if (vec_ok > 1)
            nt_store4<PO, TOut>(y, vout + i);
        else
            vout[i] = y;
    }
    // tail (or everything, when a pointer is not vector-aligned)
    for (int64_t t = nvec * kPer + (int64_t)blockIdx.x * kThreads + threadIdx.x; t < n; t += stride)
        out[t] = (TOut)((double)in[t] * g);
}

struct MixPtrs {
    const void *p[8];
};

template <typename T, bool ALIGNED>
__global__ void __launch_bounds__(kThreads) mix_kernel(const MixPtrs ins, int n_inputs,
                                                       T *__restrict__ out, int64_t n)
{
    const int64_t nvec = n / kPer;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    using P = typename std::conditional<ALIGNED, Pack<T, kPer>, PackU<T, kPer>>::type;
    P *vout = reinterpret_cast<P *>(out);
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += stride) {
        double acc[kPer];
        const P a = reinterpret_cast<const P *>(ins.p[0])[i];
#pragma unroll
        for (int k = 0; k < kPer; ++k)
            acc[k] = (double)a.v[k];
        for (int j = 1; j < n_inputs; ++j) {
            const P b = reinterpret_cast<const P *>(ins.p[j])[i];
#pragma unroll
            for (int k = 0; k < kPer; ++k)
                acc[k] = acc[k] + (double)b.v[k];
        }
        P y;
#pragma unroll
        for (int k = 0; k < kPer; ++k)
            y.v[k] = (T)acc[k];
        vout[i] = y;
    }
    for (int64_t t = nvec * kPer + (int64_t)blockIdx.x * kThreads + threadIdx.x; t < n; t += stride) {
        double acc = (double)reinterpret_cast<const T *>(ins.p[0])[t];
        for (int j = 1; j < n_inputs; ++j)
            acc = acc + (double)reinterpret_cast<const T *>(ins.p[j])[t];
        out[t] = (T)acc;
    }
}

__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <typename T>
__global__ void __launch_bounds__(kThreads) synth_kernel(T *__restrict__ out, uint64_t seed,
                                                         int64_t first, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const uint64_t u = splitmix64_at(seed, (uint64_t)(first + i));
        out[i] = (T)((double)(u >> 40) * 0x1p-23 - 1.0);
    }
}

inline bool aligned_to(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

inline int grid_for(int64_t n_vec)
{
    int64_t b = (n_vec + kThreads - 1) / kThreads;
    if (b < 1)
        b = 1;
    if (b > kMaxBlocks)
        b = kMaxBlocks;
    return (int)b;
}

class Gain final : public pipe_hip_processor {
public:
    double gain = 1.0;
    int start(hipStream_t) override { return PIPE_HIP_OK; }
    int start_lines(int, int, hipStream_t) override { return PIPE_HIP_OK; }  // stateless
    bool armable() const override { return true; }  // (stateless: a queued launch that is dropped leaves nothing behind)
    int set_param(int32_t param, const double *values, int32_t count) override
    {
        if (param != PIPE_HIP_PARAM_GAIN || count != 1 || !values)
            return PIPE_HIP_EINVAL;
        gain = values[0];  // a kernel argument: queued launches keep their own copy
        return PIPE_HIP_OK;
    }
    int run(const void *d_in, int in_dtype, void *d_out, int out_dtype, int64_t frames,
            hipStream_t s) override
    {
        const int64_t n = frames * cfg.channels * active_lines();
        if (n <= 0)
            return PIPE_HIP_OK;
        int vec_ok = aligned_to(d_in, dtype_size(in_dtype) * kPer) &&
                     aligned_to(d_out, dtype_size(out_dtype) * kPer);
        // 2 = vectors + streaming access: a pass larger than the 256 MB Infinity Cache gains 8 %
        // from not allocating in it (measured 5.57 -> 6.03 TB/s on 512 MB), a smaller one loses
        // the reuse it would have had (4.82 -> 4.55 TB/s on 128 MB)
        static const int64_t nt_min = std::getenv("PIPE_HIP_GAIN_NT_MIN_BYTES")
                                          ? std::atoll(std::getenv("PIPE_HIP_GAIN_NT_MIN_BYTES"))
                                          : (int64_t)384 << 20;
        if (vec_ok && n * (int64_t)(dtype_size(in_dtype) + dtype_size(out_dtype)) >= nt_min)
            vec_ok = 2;
        const dim3 grid(grid_for(n / kPer > 0 ? n / kPer : n));
        PH_TRY(timer.begin(s));
        // a ProcessFunc-form buffer: this launch is the call's last operation and signals its completion
        hipEvent_t done = completion;
        completion = nullptr;
        if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32) {
            hipExtLaunchKernelGGL((gain_kernel<float, float>), grid, dim3(kThreads), 0, s, nullptr, done, 0,
                               (const float *)d_in, (float *)d_out, n, gain, vec_ok);
            last_kernel = "gain_kernel<f32,f32>";
        } else if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F64) {
            hipExtLaunchKernelGGL((gain_kernel<double, double>), grid, dim3(kThreads), 0, s, nullptr, done, 0,
                               (const double *)d_in, (double *)d_out, n, gain, vec_ok);
            last_kernel = "gain_kernel<f64,f64>";
        } else if (in_dtype == PIPE_HIP_F32) {
            hipExtLaunchKernelGGL((gain_kernel<float, double>), grid, dim3(kThreads), 0, s, nullptr, done, 0,
                               (const float *)d_in, (double *)d_out, n, gain, vec_ok);
            last_kernel = "gain_kernel<f32,f64>";
        } else {
            hipExtLaunchKernelGGL((gain_kernel<double, float>), grid, dim3(kThreads), 0, s, nullptr, done, 0,
                               (const double *)d_in, (float *)d_out, n, gain, vec_ok);
            last_kernel = "gain_kernel<f64,f32>";
        }
        PH_HIP(hipGetLastError());
        PH_TRY(timer.end(s));
        return PIPE_HIP_OK;
    }
};

class Mix final : public pipe_hip_processor {
public:
    int inputs = 2;
    bool single_input() const override { return false; }
    int start(hipStream_t) override { return PIPE_HIP_OK; }
    // a mix has no single-input form: ProcessFunc carries one input (pipe.go:62-64)
    int run(const void *, int, void *, int, int64_t, hipStream_t) override { return PIPE_HIP_EINVAL; }
    int run_n(const void *const *d_ins, int32_t n_inputs, void *d_out, int64_t frames, hipStream_t s)
    {
        if (n_inputs != inputs)
            return PIPE_HIP_EINVAL;
        const int64_t n = frames * cfg.channels * cfg.lines;
        if (n <= 0)
            return PIPE_HIP_OK;
        MixPtrs ptrs{};
        int vec_ok = aligned_to(d_out, dtype_size(cfg.dtype) * kPer);
        for (int i = 0; i < n_inputs; ++i) {
            if (!d_ins[i])
                return PIPE_HIP_EINVAL;
            ptrs.p[i] = d_ins[i];
            vec_ok = vec_ok && aligned_to(d_ins[i], dtype_size(cfg.dtype) * kPer);
        }
        const dim3 grid(grid_for(n / kPer > 0 ? n / kPer : n));
        hipEvent_t ev_a = nullptr, ev_b = nullptr;
        PH_TRY(timer.pair(&ev_a, &ev_b));
        if (cfg.dtype == PIPE_HIP_F32) {
            if (vec_ok)
                hipExtLaunchKernelGGL((mix_kernel<float, true>), grid, dim3(kThreads), 0, s, ev_a, ev_b, 0, ptrs, n_inputs, (float *)d_out, n);
            else
                hipExtLaunchKernelGGL((mix_kernel<float, false>), grid, dim3(kThreads), 0, s, ev_a, ev_b, 0, ptrs, n_inputs, (float *)d_out, n);
            last_kernel = "mix_kernel<f32>";
        } else {
            if (vec_ok)
                hipExtLaunchKernelGGL((mix_kernel<double, true>), grid, dim3(kThreads), 0, s, ev_a, ev_b, 0, ptrs, n_inputs, (double *)d_out, n);
            else
                hipExtLaunchKernelGGL((mix_kernel<double, false>), grid, dim3(kThreads), 0, s, ev_a, ev_b, 0, ptrs, n_inputs, (double *)d_out, n);
            last_kernel = "mix_kernel<f64>";
        }
        PH_HIP(hipGetLastError());
        return PIPE_HIP_OK;
    }
};

}  // namespace

bool gain_value(const pipe_hip_processor *p, double *g)
{
    auto *q = dynamic_cast<const Gain *>(p);
    if (!q)
        return false;
    *g = q->gain;
    return true;
}

int make_gain(const pipe_hip_config *cfg, double gain, pipe_hip_processor **out)
{
    auto p = std::make_unique<Gain>();
    PH_TRY(p->init_common(cfg));
    p->gain = gain;
    *out = p.release();
    return PIPE_HIP_OK;
}

int make_mix(const pipe_hip_config *cfg, int32_t inputs, pipe_hip_processor **out)
{
    if (inputs < 2 || inputs > 8)
        return PIPE_HIP_EINVAL;
    auto p = std::make_unique<Mix>();
    PH_TRY(p->init_common(cfg));
    p->inputs = inputs;
    *out = p.release();
    return PIPE_HIP_OK;
}

int mix_run(pipe_hip_processor *p, const void *const *d_ins, int32_t n_inputs, void *d_out,
            int64_t frames, hipStream_t s)
{
    auto *m = dynamic_cast<Mix *>(p);
    if (!m)
        return PIPE_HIP_EINVAL;
    return m->run_n(d_ins, n_inputs, d_out, frames, s);
}

// Rows of L Lines between scattered (pinned host, device-visible) buffers and one packed
// device block, 8 bytes per lane: the gather / scatter of pipe_hip_process_lines_pinned.
// tab[l] = buffer of Line l (nullptr: idle slot), words[l] = its length in 8-byte words;
// a packed row is row_words long, the tail past words[l] is zero-filled on gather.
__global__ void gather_rows_kernel(const uint64_t *const *__restrict__ tab, const int *__restrict__ words,
                                   uint64_t *__restrict__ dst, int row_words)
{
    const int l = blockIdx.y;
    const uint64_t *__restrict__ src = tab[l];
    const int have = src ? words[l] : 0;
    uint64_t *__restrict__ row = dst + (int64_t)l * row_words;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < row_words; i += gridDim.x * blockDim.x)
        row[i] = i < have ? src[i] : 0ull;
}

__global__ void scatter_rows_kernel(uint64_t *const *__restrict__ tab, const int *__restrict__ words,
                                    const uint64_t *__restrict__ src, int row_words)
{
    const int l = blockIdx.y;
    uint64_t *__restrict__ dst = tab[l];
    const int have = dst ? words[l] : 0;
    const uint64_t *__restrict__ row = src + (int64_t)l * row_words;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < have; i += gridDim.x * blockDim.x)
        dst[i] = row[i];
}

int launch_gather_rows(const void *const *tab, const int *words, void *dst, int row_words, int lines, hipStream_t s)
{
    if (row_words <= 0 || lines <= 0)
        return PIPE_HIP_OK;
    const int bx = (row_words + 4 * kThreads - 1) / (4 * kThreads);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(bx < 64 ? bx : 64), (unsigned)lines), dim3(kThreads), 0, s,
                       reinterpret_cast<const uint64_t *const *>(tab), words, static_cast<uint64_t *>(dst), row_words);
    PH_HIP(hipGetLastError());
    return PIPE_HIP_OK;
}

int launch_scatter_rows(void *const *tab, const int *words, const void *src, int row_words, int lines, hipStream_t s)
{
    if (row_words <= 0 || lines <= 0)
        return PIPE_HIP_OK;
    const int bx = (row_words + 4 * kThreads - 1) / (4 * kThreads);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)(bx < 64 ? bx : 64), (unsigned)lines), dim3(kThreads), 0, s,
                       reinterpret_cast<uint64_t *const *>(tab), words, static_cast<const uint64_t *>(src), row_words);
    PH_HIP(hipGetLastError());
    return PIPE_HIP_OK;
}

int launch_synth_fill(void *d_out, int dtype, uint64_t seed, int64_t first, int64_t n, hipStream_t s)
{
    if (n <= 0)
        return PIPE_HIP_OK;
    const dim3 grid(grid_for(n));
    if (dtype == PIPE_HIP_F32)
        hipLaunchKernelGGL(synth_kernel<float>, grid, dim3(kThreads), 0, s, (float *)d_out, seed,
                           first, n);
    else
        hipLaunchKernelGGL(synth_kernel<double>, grid, dim3(kThreads), 0, s, (double *)d_out, seed,
                           first, n);
    PH_HIP(hipGetLastError());
    return PIPE_HIP_OK;
}

}  // namespace pipehip
