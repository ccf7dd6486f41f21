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
#include <cstdint>
#include <cstdlib>
#include <utility>

#include <hip/hip_ext.h>

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

// LDS-tiled form (used whenever the polyphase table fits): a workgroup owns kOutTile
// consecutive output frames of one Line.  The table (tap-major: h[j*up + p], so lanes of
// consecutive outputs -- phases 'down' apart -- hit distinct banks) and the input window
// (per-channel float64 planes) are staged once; one lane = one output FRAME for all
// channels, so every tap is read once per frame and the result leaves as one contiguous
// C-element vector per lane (coalesced).  Same fma order as the gather kernel: bit-exact.
constexpr int kOutTile = 1024;

struct TiledArgs {
    ResampleArgs r;
    int win;       // staged input frames per tile (upper bound)
    int plane;     // plane stride (elements)
    int cx_log;    // log2 of staging columns (pow2 >= C)
    int tiles_per_line;
    int tile_out;  // output frames per tile (kOutTile, or a multiple of q when the taps sit in registers)
    int q;         // lanes that compute: the outputs of lane l are l, l + q, l + 2q, ... of the tile
};

// CH channels of one output frame: acc[c] = fma(h[j], x_c[n-j], acc[c]), j ascending
// The same with the lane's T = TT taps in registers: when every output of a lane has the same
// phase (its outputs are a multiple of `up` apart) the taps are read from the table once per
// launch instead of once per output -- a third of the bytes this kernel moves through LDS.
// One ds_read_b64 with an immediate offset.  Written as an instruction because the compiler merges
// neighbouring window reads into ds_read2_b64, which the LDS serves at HALF the rate of two
// ds_read_b64 (8 cycles against 2 x 2 per wave-instruction, MI355X_MICROARCH.md "LDS") -- and the
// window reads are what bounds this kernel.  The value is only valid behind lds_wait below.
template <int OFF>
__device__ __forceinline__ double lds_read_f64(unsigned addr)
{
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N, int... I>
__device__ __forceinline__ void lds_read_run(double (&v)[N], unsigned addr, std::integer_sequence<int, I...>)
{
    ((v[I] = lds_read_f64<8 * (N - 1 - I)>(addr)), ...);  // v[i] = element (N - 1 - i) above addr
}
// all but the CNT most recent LDS reads have landed (the LDS returns in order); the values pass
// through so that their uses stay behind the wait
template <int CNT>
__device__ __forceinline__ void lds_wait(double (&a)[4])
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(CNT));
}
template <int CNT>
__device__ __forceinline__ void lds_wait(double (&a)[4], double (&b)[4])
{
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 : "n"(CNT));
}

template <int CH, int TT, typename TOut>
__device__ __forceinline__ void resample_taps_reg(const double (&h)[TT > 0 ? TT : 1], const double *__restrict__ xp,
                                                  int plane, TOut *__restrict__ o)
{
    double acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
        acc[c] = 0.0;
    if constexpr (CH == 2 && TT % 4 == 0) {
        // four taps of both channels per group (x[-j0 - 3 .. -j0], plain ds_read_b64 each), the next
        // group requested before the current one is waited for: eight reads stay in flight
        typedef __attribute__((address_space(3))) const double *lds_ptr;
        const unsigned a0 = (unsigned)(uintptr_t)(lds_ptr)xp;  // LDS byte address of x[0], channel 0
        const unsigned a1 = a0 + 8u * (unsigned)plane;
        double v0[2][4], v1[2][4];  // v[i] = x[-j0 - i]
        lds_read_run<4>(v0[0], a0 - 8u * 3u, std::make_integer_sequence<int, 4>{});
        lds_read_run<4>(v1[0], a1 - 8u * 3u, std::make_integer_sequence<int, 4>{});
#pragma unroll
        for (int g = 0; g < TT / 4; ++g) {
            const int j0 = 4 * g;
            if (g + 1 < TT / 4) {
                lds_read_run<4>(v0[(g + 1) & 1], a0 - 8u * (unsigned)(j0 + 7), std::make_integer_sequence<int, 4>{});
                lds_read_run<4>(v1[(g + 1) & 1], a1 - 8u * (unsigned)(j0 + 7), std::make_integer_sequence<int, 4>{});
                lds_wait<8>(v0[g & 1], v1[g & 1]);
            } else {
                lds_wait<0>(v0[g & 1], v1[g & 1]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[0] = __builtin_fma(h[j0 + i], v0[g & 1][i], acc[0]);
                acc[1] = __builtin_fma(h[j0 + i], v1[g & 1][i], acc[1]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < TT; ++j) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                acc[c] = __builtin_fma(h[j], xp[c * plane - j], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
        o[c] = (TOut)acc[c];
}

// float32 stream, window kept as float32 {channel 2p, channel 2p + 1} pairs: ONE ds_read_b64 brings a
// frame of both channels (the float64 planes need two), widened in registers -- the kernel is bound by
// the number of LDS reads in flight, not by the VALU.  Eight taps a group, the next group requested
// before the current one is waited for.
template <int TT, typename TOut>
__device__ __forceinline__ void resample_taps_reg_f32pair(const double (&h)[TT], const double *__restrict__ xp,
                                                          TOut *__restrict__ o)
{
    static_assert(TT % 4 == 0, "four taps a group");
    typedef __attribute__((address_space(3))) const double *lds_ptr;
    const unsigned a0 = (unsigned)(uintptr_t)(lds_ptr)xp;  // LDS byte address of the pair's frame x[0]
    double acc0 = 0.0, acc1 = 0.0;
    double v[2][4];
    lds_read_run<4>(v[0], a0 - 8u * 3u, std::make_integer_sequence<int, 4>{});
#pragma unroll
    for (int g = 0; g < TT / 4; ++g) {
        const int j0 = 4 * g;
        if (g + 1 < TT / 4) {
            lds_read_run<4>(v[(g + 1) & 1], a0 - 8u * (unsigned)(j0 + 7), std::make_integer_sequence<int, 4>{});
            lds_wait<4>(v[g & 1]);
        } else {
            lds_wait<0>(v[g & 1]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 f = __builtin_bit_cast(float2, v[g & 1][i]);
            acc0 = __builtin_fma(h[j0 + i], (double)f.x, acc0);
            acc1 = __builtin_fma(h[j0 + i], (double)f.y, acc1);
        }
    }
    o[0] = (TOut)acc0;
    o[1] = (TOut)acc1;
}

template <int CH, typename TOut>
__device__ __forceinline__ void resample_taps(const double *__restrict__ hp, const double *__restrict__ xp,
                                              int T, int up, int plane, TOut *__restrict__ o)
{
    double acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
        acc[c] = 0.0;
#pragma unroll 8
    for (int j = 0; j < T; ++j) {
        const double hj = *hp;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            acc[c] = __builtin_fma(hj, xp[c * plane], acc[c]);
        hp += up;
        xp -= 1;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
        o[c] = (TOut)acc[c];
}

template <typename TIn, typename TOut, int TT, bool F32WIN = false>
__global__ void __launch_bounds__(kThreads) resample_tiled_kernel(const TiledArgs t)  // (launched with 64..256 threads)
{
    static_assert(!F32WIN || (TT > 0 && sizeof(TIn) == 4), "the float32 pair window: float32 streams, taps in registers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const ResampleArgs &a = t.r;
    const int nthr = (int)blockDim.x;
    // taps in registers (TT > 0): no table in LDS at all -- the lane's T taps come from global memory
    // once per launch -- so a workgroup is its planes only and more of them share a CU
    double *tab = reinterpret_cast<double *>(smem_raw);           // [T][up]   (TT == 0)
    double *xs = TT > 0 ? tab : tab + (size_t)a.T * a.up;          // [C][plane]
    const int H = a.T - 1;
    if constexpr (TT == 0) {   // proto is already [p + j*up]; staged once per (persistent) workgroup, 8 loads in flight
        const int ntab = a.T * a.up;
        for (int k0 = threadIdx.x; k0 < ntab; k0 += 8 * nthr) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * nthr;
                v[u] = a.proto[k < ntab ? k : 0];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * nthr;
                if (k < ntab)
                    tab[k] = v[u];
            }
        }
    }

    // TT > 0: this lane's outputs all have phase p = (r0 + lane*down) mod up, where r0 is the
    // phase of the call's first output (tiles are a multiple of `up` outputs long): its taps
    // h[j] = proto[p + j*up] are fetched once
    double hreg[TT > 0 ? TT : 1];
    if constexpr (TT > 0) {
        const unsigned r_call = (unsigned)((a.out_total * a.down) % a.up);
        const unsigned p = (r_call + (unsigned)threadIdx.x * (unsigned)a.down) % (unsigned)a.up;
#pragma unroll
        for (int j = 0; j < TT; ++j)
            hreg[j] = a.proto[p + j * a.up];
    }

    const int ntiles = t.tiles_per_line * a.lines;
    for (int tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
    const int line = tile_id / t.tiles_per_line;
    const int tile = tile_id - line * t.tiles_per_line;
    const int64_t m0 = a.out_total + (int64_t)tile * t.tile_out;  // first output (global index)
    const int64_t i0 = (int64_t)tile * t.tile_out;                // ... relative to this call
    const int nout = (int)min((int64_t)t.tile_out, a.out_frames - i0);

    // input frame (relative to this call's input) read by the tile's first output, minus history
    const int64_t t0 = m0 * a.down;
    const int64_t nfirst = t0 / a.up - a.in_total;
    const unsigned r0 = (unsigned)(t0 % a.up);
    const int64_t base = nfirst - H;  // frame of plane element 0
    const TIn *__restrict__ in = reinterpret_cast<const TIn *>(a.in) + (int64_t)line * a.in_frames * a.C;
    const double *__restrict__ hist = a.hist + (int64_t)line * H * a.C;
    {
        const int tx = threadIdx.x & ((1 << t.cx_log) - 1);
        const int ty = threadIdx.x >> t.cx_log;
        const int FY = nthr >> t.cx_log;
        const bool ok = tx < a.C;
        const int64_t last = a.in_frames - 1;
        const TIn *__restrict__ src = in + (ok ? tx : 0);
        for (int f0 = ty; f0 < t.win; f0 += 8 * FY) {
            TIn v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int64_t g = base + f0 + u * FY;
                g = g < 0 ? 0 : (g > last ? last : g);
                v[u] = src[g * a.C];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int f = f0 + u * FY;
                const int64_t g = base + f;
                const double loaded = (double)v[u];
                if (ok && f < t.win) {
                    double w = 0.0;
                    if (g >= 0)
                        w = g <= last ? loaded : 0.0;
                    else if (g >= -(int64_t)H)
                        w = hist[(g + H) * a.C + tx];
                    if constexpr (F32WIN)  // {channel 2p, 2p + 1} of a frame side by side, plane p
                        reinterpret_cast<float *>(xs)[((size_t)(tx >> 1) * t.plane + f) * 2 + (tx & 1)] = (float)w;
                    else
                        xs[tx * t.plane + f] = w;
                }
            }
        }
    }
    __syncthreads();

    TOut *__restrict__ out = reinterpret_cast<TOut *>(a.out) + ((int64_t)line * a.out_cap + i0) * a.C;
    for (int ml = threadIdx.x; ml < nout && (int)threadIdx.x < t.q; ml += t.q) {
        const unsigned tt = r0 + (unsigned)ml * (unsigned)a.down;  // < up + 1024*down: fits 32 bits
        const unsigned nrel = tt / (unsigned)a.up;                 // input frame relative to nfirst
        const unsigned p = tt - nrel * (unsigned)a.up;
        const double *__restrict__ x = xs + nrel + H;  // element of frame (nfirst + nrel)
        const double *__restrict__ h = tab + p;
        // channels in compile-time groups of 8/4/2/1: the tap loop carries no predicates
        int c0 = 0;
        while (c0 < a.C) {
            const int left = a.C - c0;
            const double *__restrict__ xp = x + c0 * t.plane;
            TOut *__restrict__ o = out + (int64_t)ml * a.C + c0;
            if constexpr (F32WIN) {
                resample_taps_reg_f32pair<TT>(hreg, xs + (size_t)(c0 >> 1) * t.plane + nrel + H, o);
                c0 += 2;
            } else if constexpr (TT > 0) {
                if (left >= 4 && TT % 4 != 0) {  // (channel pairs take the ds_read_b64 form: resample_taps_reg)
                    resample_taps_reg<4, TT>(hreg, xp, t.plane, o);
                    c0 += 4;
                } else if (left >= 2) {
                    resample_taps_reg<2, TT>(hreg, xp, t.plane, o);
                    c0 += 2;
                } else {
                    resample_taps_reg<1, TT>(hreg, xp, t.plane, o);
                    c0 += 1;
                }
            } else if (left >= 8) {
                resample_taps<8>(h, xp, a.T, a.up, t.plane, o);
                c0 += 8;
            } else if (left >= 4) {
                resample_taps<4>(h, xp, a.T, a.up, t.plane, o);
                c0 += 4;
            } else if (left >= 2) {
                resample_taps<2>(h, xp, a.T, a.up, t.plane, o);
                c0 += 2;
            } else {
                resample_taps<1>(h, xp, a.T, a.up, t.plane, o);
                c0 += 1;
            }
        }
    }
    __syncthreads();  // the planes are rewritten by the next tile
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
        // taps in registers: T one of the specialised sizes and `up` small enough that a
        // workgroup's lanes cover whole periods of the phase pattern
        const bool reg_taps = (T_ == 8 || T_ == 12 || T_ == 16 || T_ == 24 || T_ == 32) && up_ <= kThreads &&
                              !std::getenv("PIPE_HIP_RESAMPLE_LDS_TAPS");
        const int q = reg_taps ? up_ * (kThreads / up_) : kThreads;
        // (taps in registers: as many waves as hold the q computing lanes, e.g. 3 for 160)
        const int threads = reg_taps ? (q + 63) / 64 * 64 : kThreads;
        // outputs per tile: kOutTile, halved until the workgroup's LDS -- planes, and the table unless
        // the taps sit in registers -- fits 64 KB (wide Lines: 8 channels at 160 x 24 stay on this
        // kernel: 74 us where the gather kernel takes 213) -- never below one output per computing lane
        // float32 streams with the taps in registers keep the window as float32 channel pairs
        // (from four channels on: with two the float64 planes are faster, 51.5 against 59.7 us)
        const bool f32win = reg_taps && in_dtype == PIPE_HIP_F32 && cfg.channels % 2 == 0 && cfg.channels >= 4 && T_ % 4 == 0 &&
                            !std::getenv("PIPE_HIP_RESAMPLE_F64_PLANES");
        int tile_out = 0, win = 0, plane = 0;
        size_t lds = 0;
        for (int cap = kOutTile; cap >= q; cap /= 2) {
            tile_out = reg_taps ? q * (cap / q) : cap;
            // staged window of a tile: frames read by tile_out outputs, plus history, plus slack
            win = (int)(((int64_t)tile_out * down_ + up_ - 1) / up_) + T_ + 1;
            plane = win + 1;
            plane += (16 - plane % 32 + 32) % 32;  // plane stride == 16 (mod 32): channel planes on distinct banks
            lds = sizeof(double) * ((reg_taps ? 0 : (size_t)T_ * up_) + (size_t)plane * (f32win ? cfg.channels / 2 : cfg.channels));
            if (lds <= 64 * 1024)
                break;
        }
        // A tap table that leaves no room for the planes in 64 KB (160 x 48 taps are 61 KB) still beats
        // the gather kernel from a larger LDS allocation, one workgroup per CU: the largest tile that
        // fits 144 KB.
        constexpr size_t kBigLds = 144 * 1024;
        if (lds > 64 * 1024 && !reg_taps) {
            for (int cap = kOutTile; cap >= q; cap /= 2) {
                tile_out = cap;
                win = (int)(((int64_t)tile_out * down_ + up_ - 1) / up_) + T_ + 1;
                plane = win + 1;
                plane += (16 - plane % 32 + 32) % 32;
                lds = sizeof(double) * ((size_t)T_ * up_ + (size_t)plane * cfg.channels);
                if (lds <= kBigLds)
                    break;
            }
        }
        const bool big_lds = lds > 64 * 1024;
        const bool tiled = lds <= (reg_taps ? (size_t)64 * 1024 : kBigLds) && n_out > 0 && !std::getenv("PIPE_HIP_RESAMPLE_GATHER");
        if (total > 0 && tiled) {
            TiledArgs t{};
            t.r = a;
            t.win = win;
            t.plane = plane;
            t.cx_log = 0;
            while ((1 << t.cx_log) < cfg.channels)
                ++t.cx_log;
            t.tile_out = tile_out;
            t.q = q;
            t.tiles_per_line = (int)((n_out + tile_out - 1) / tile_out);
            const int64_t ntiles = (int64_t)t.tiles_per_line * cfg.lines;
            const int64_t slots = (reg_taps ? 4 : 3) * 256;  // workgroups resident per CU (registers bound them: 12 waves)
            const int64_t per = (ntiles + slots - 1) / slots;
            const dim3 grid((unsigned)((ntiles + per - 1) / per));
            // events attached to the kernel's own dispatch (what rocprofv3 reports)
            hipEvent_t ev_a = nullptr, ev_b = nullptr;
            PH_TRY(timer.pair(&ev_a, &ev_b));
#define PH_RS(TI, TO, NAME)                                                                              \
    do {                                                                                                 \
        switch (reg_taps ? T_ : 0) {                                                                     \
        case 8: hipExtLaunchKernelGGL((resample_tiled_kernel<TI, TO, 8>), grid, dim3(threads), lds, s, ev_a, ev_b, 0, t); break;   \
        case 12: hipExtLaunchKernelGGL((resample_tiled_kernel<TI, TO, 12>), grid, dim3(threads), lds, s, ev_a, ev_b, 0, t); break; \
        case 16: hipExtLaunchKernelGGL((resample_tiled_kernel<TI, TO, 16>), grid, dim3(threads), lds, s, ev_a, ev_b, 0, t); break; \
        case 24: hipExtLaunchKernelGGL((resample_tiled_kernel<TI, TO, 24>), grid, dim3(threads), lds, s, ev_a, ev_b, 0, t); break; \
        case 32: hipExtLaunchKernelGGL((resample_tiled_kernel<TI, TO, 32>), grid, dim3(threads), lds, s, ev_a, ev_b, 0, t); break; \
        default:                                                                                         \
            if (big_lds)                                                                                 \
                PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(resample_tiled_kernel<TI, TO, 0>),         \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
            hipExtLaunchKernelGGL((resample_tiled_kernel<TI, TO, 0>), grid, dim3(threads), lds, s, ev_a, ev_b, 0, t);      \
            break;                                                                                       \
        }                                                                                                \
        last_kernel = NAME;                                                                              \
    } while (0)
#define PH_RS32(TO, NAME)                                                                                         \
    do {                                                                                                          \
        switch (T_) {                                                                                             \
        case 8: hipExtLaunchKernelGGL((resample_tiled_kernel<float, TO, 8, true>), grid, dim3(threads), lds, s, ev_a, ev_b, 0, t); break;   \
        case 12: hipExtLaunchKernelGGL((resample_tiled_kernel<float, TO, 12, true>), grid, dim3(threads), lds, s, ev_a, ev_b, 0, t); break; \
        case 16: hipExtLaunchKernelGGL((resample_tiled_kernel<float, TO, 16, true>), grid, dim3(threads), lds, s, ev_a, ev_b, 0, t); break; \
        case 24: hipExtLaunchKernelGGL((resample_tiled_kernel<float, TO, 24, true>), grid, dim3(threads), lds, s, ev_a, ev_b, 0, t); break; \
        default: hipExtLaunchKernelGGL((resample_tiled_kernel<float, TO, 32, true>), grid, dim3(threads), lds, s, ev_a, ev_b, 0, t); break; \
        }                                                                                                         \
        last_kernel = NAME;                                                                                       \
    } while (0)
            if (f32win && out_dtype == PIPE_HIP_F32)
                PH_RS32(float, "resample_tiled_kernel<f32,f32,pairs>");
            else if (f32win)
                PH_RS32(double, "resample_tiled_kernel<f32,f64,pairs>");
            else if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32)
                PH_RS(float, float, "resample_tiled_kernel<f32,f32>");
            else if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F64)
                PH_RS(double, double, "resample_tiled_kernel<f64,f64>");
            else if (in_dtype == PIPE_HIP_F32)
                PH_RS(float, double, "resample_tiled_kernel<f32,f64>");
            else
                PH_RS(double, float, "resample_tiled_kernel<f64,f32>");
#undef PH_RS32
#undef PH_RS
            PH_HIP(hipGetLastError());
        } else if (total > 0) {
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
            // a ProcessFunc-form buffer: the call's last launch signals its completion
            hipEvent_t done = completion;
            completion = nullptr;
            if (in_dtype == PIPE_HIP_F32)
                hipExtLaunchKernelGGL(resample_hist_kernel<float>, hg, dim3(256), 0, s, nullptr, done, 0,
                                      static_cast<const float *>(d_in), a.hist, hn, in_frames, H,
                                      cfg.channels);
            else
                hipExtLaunchKernelGGL(resample_hist_kernel<double>, hg, dim3(256), 0, s, nullptr, done, 0,
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
