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
#include <vector>

#include <hip/hip_ext.h>

#include "common.hpp"
#include "resampler_rows.hpp"

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

// ---- two adjacent outputs per lane: 2-channel streams, taps in registers ----------------------------
// The tiled kernel above reads the window once per OUTPUT: 2 T ds_read_b64 per output frame, and the
// LDS is what bounds it.  Two ADJACENT outputs m, m + 1 read nearly the same frames: output m reads
// x[n - j], output m + 1 reads x[n + d - j] with d = floor((t + down) / up) - floor(t / up), which is
// DMIN = down / up or DMIN + 1 -- fixed per lane when a lane's outputs are a multiple of `up` apart
// (the same condition that lets it keep its taps in registers).  A lane therefore takes outputs
// (2 l, 2 l + 1) of every step of 2 ql outputs (2 ql a multiple of up), reads the W = T + DMIN + 1
// frames X[i] = x[n + DMIN + 1 - i] once per channel and feeds both sums:
//     A (output m)    : sum_j hA[j]  X[DMIN + 1 + j]
//     B (output m + 1): sum_j hB[j]  X[1 - e + j],      e = d - DMIN in {0, 1}
// B's taps are stored shifted by the lane's e (hB'[i] = hB[i - 1 + e]), so every register index is a
// compile-time constant; the two end positions that one value of e does not use are skipped with a
// select, not multiplied by a zero tap (0 x Inf would be NaN where the oracle never touches that
// sample).  Same operations in the same order per output as the oracle: bit-exact.  LDS reads per
// output frame: 48 -> 25 (160/147) or 26 (147/160).
// Staging moves 16 bytes per lane (two float32 frames of both channels, or one float64 frame) and is
// double-buffered: the next tile's window is requested before this tile's outputs are computed and
// written to the other pair of planes afterwards, one barrier per tile.
struct PairArgs {
    ResampleArgs r;
    int win;        // staged frames per tile (even)
    int plane;      // plane stride (doubles, 16-byte aligned planes)
    int tiles_per_line;
    int tile_out;   // outputs per tile = 2 * ql * steps
    int ql;         // computing lanes
    int steps;
    int adv;        // input frames a step advances: 2 * ql * down / up
    int vec_ok;     // every Line's input starts 16-byte aligned
    int lead;       // virtual outputs ahead of the call's first one: tiles start where the phase is 0
    int64_t nb;     // input frame (relative to this call) the first virtual output reads
    const double *ptaps;  // [2 T + 1][ql]: the lanes' taps, hA rows then the shifted hB rows
    unsigned long long *prof;  // PH_RS_PROF builds: [wave][5] s_memtime ticks per phase
};
#ifdef PH_RS_PROF
#define PH_RS_STAMP(i)                                                  \
    do {                                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
        rsprof[i] += now_ - rslast;                                     \
        rslast = now_;                                                  \
    } while (0)
#else
#define PH_RS_STAMP(i) \
    do {               \
    } while (0)
#endif
constexpr int kPairPad = 4;   // frames staged below the oldest one a tile's first output reads (grouped reads overshoot)
constexpr int kPairVecs = 3;  // pieces of the window a thread stages per tile, at most (registers: three waves per SIMD)

template <int CNT>
__device__ __forceinline__ void lds_wait_cnt()
{
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CNT) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_pin(double (&v)[N])  // uses of v stay behind the wait above
{
#pragma unroll
    for (int i = 0; i < N; ++i)
        asm volatile("" : "+v"(v[i]));
}
// v[i] = X[G0 + i], X[pos] at byte offset 8 * (NPOS - 1 - pos) above `lo`
template <int G0, int NPOS, int N, int... I>
__device__ __forceinline__ void lds_read_pos(double (&v)[N], unsigned lo, std::integer_sequence<int, I...>)
{
    ((v[I] = lds_read_f64<8 * (NPOS - 1 - (G0 + I))>(lo)), ...);
}
template <typename F, int... Gs>
__device__ __forceinline__ void for_each_const(std::integer_sequence<int, Gs...>, F &&f)
{
    (f(std::integral_constant<int, Gs>{}), ...);
}

template <typename T>
struct PairRaw;  // one 16-byte (float32: two frames) / 32-byte (float64: two frames) piece of the window
template <>
struct PairRaw<float> {
    // (element-aligned: gfx950 loads 16 bytes from any 4-byte address, so a stream that starts at an odd
    // frame of its buffer keeps this kernel)
    struct __attribute__((packed, aligned(4))) U4 {
        float x, y, z, w;
    };
    U4 v;
    __device__ __forceinline__ void load(const float *p) { v = *reinterpret_cast<const U4 *>(p); }
    // the piece at byte soff + voff of a raw buffer: out of range reads as zero (silence past the end of the input)
    __device__ __forceinline__ void load_buf(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
    {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        v = __builtin_bit_cast(U4, (v4u)__builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    }
    static constexpr unsigned kBytes = 16;
    __device__ __forceinline__ void widen(double (&c0)[2], double (&c1)[2]) const
    {
        c0[0] = (double)v.x;
        c1[0] = (double)v.y;
        c0[1] = (double)v.z;
        c1[1] = (double)v.w;
    }
};
template <>
struct PairRaw<double> {
    struct __attribute__((packed, aligned(8))) U2 {
        double x, y;
    };
    U2 a, b;
    __device__ __forceinline__ void load(const double *p)
    {
        a = *reinterpret_cast<const U2 *>(p);
        b = *reinterpret_cast<const U2 *>(p + 2);
    }
    __device__ __forceinline__ void load_buf(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
    {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        a = __builtin_bit_cast(U2, (v4u)__builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
        b = __builtin_bit_cast(U2, (v4u)__builtin_amdgcn_raw_buffer_load_b128(r, voff + 16u, soff, 0));
    }
    static constexpr unsigned kBytes = 32;
    __device__ __forceinline__ void widen(double (&c0)[2], double (&c1)[2]) const
    {
        c0[0] = a.x;
        c1[0] = a.y;
        c0[1] = b.x;
        c1[1] = b.y;
    }
};

// Ablation builds (scripts/build_ablate_lib.sh; never the shipped library): 1 = no tap loops
// (staging and stores only), 2 = tap loops without their LDS reads, 3 = no staging, 4 = no stores
#ifndef PH_RS_ABLATE
#define PH_RS_ABLATE 0
#endif

template <typename TIn, typename TOut, int TT, int DMIN>
__global__ void __launch_bounds__(kThreads) resample_pair_kernel(const PairArgs t)  // (launched with 64..256 threads)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *bufs = reinterpret_cast<double *>(smem_raw);  // [2 tiles][2 channels][plane]
    const ResampleArgs &a = t.r;
    const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
    constexpr int H = TT - 1;
    constexpr int W = TT + DMIN + 1;  // frames a pair reads
    constexpr int G = 4, NG = (W + G - 1) / G, NPOS = NG * G;

#ifdef PH_RS_PROF
    unsigned long long rsprof[5] = {}, rslast = __builtin_amdgcn_s_memtime();
#endif
    // ---- the lane's two outputs (fixed for the whole launch).  Tiles start at outputs whose phase is 0
    // (multiples of `up` counted from Start), so a lane's phases, offset and taps do not depend on the
    // call: they come from a table made once, one coalesced load per tap.
    const bool computing = tid < t.ql;
    const unsigned ltid = computing ? (unsigned)tid : 0u;
    const unsigned ttA = 2u * ltid * (unsigned)a.down, nA = ttA / (unsigned)a.up;
    const unsigned nB = (ttA + (unsigned)a.down) / (unsigned)a.up;
    const bool e1 = (int)(nB - nA) - DMIN != 0;  // e = 1: output B reads X[0 .. T-1]; e = 0: X[1 .. T]
    double hA[TT], hB[TT + 1];
    auto load_taps = [&]() {
#pragma unroll
        for (int j = 0; j < TT; ++j)
            hA[j] = t.ptaps[j * t.ql + ltid];
#pragma unroll
        for (int i = 0; i <= TT; ++i)
            hB[i] = t.ptaps[(TT + i) * t.ql + ltid];
    };

    const int ntiles = t.tiles_per_line * a.lines;
    const int npairs = t.win / 2;

    // window geometry of a tile (no division beyond the Line split: tiles are whole periods)
    auto geometry = [&](int tile_id, int &line, int64_t &i0, int64_t &base_e, int &odd) {
        line = a.lines == 1 ? 0 : tile_id / t.tiles_per_line;
        const int tile = tile_id - line * t.tiles_per_line;
        i0 = (int64_t)tile * t.tile_out - t.lead;        // (negative: the virtual outputs ahead of the call)
        const int64_t nfirst = t.nb + (int64_t)tile * (t.steps * t.adv);  // frame the tile's first output reads
        const int64_t base = nfirst - H - kPairPad;
        base_e = base & ~(int64_t)1;  // even: 16-byte pieces of the input, 16-byte plane cells
        odd = (int)(base - base_e);
    };
    // frame g (relative to this call's input) of channel c, the slow way: history below 0, silence past the end
    auto frame_value = [&](const TIn *__restrict__ in, const double *__restrict__ hist, int64_t g, int c) -> double {
        if (g >= 0)
            return g < a.in_frames ? (double)in[g * 2 + c] : 0.0;
        return g >= -(int64_t)H ? hist[(g + H) * 2 + c] : 0.0;
    };

    PairRaw<TIn> pre[kPairVecs];
    unsigned fast = 0;  // bit k: vector k of the prefetched tile is a plain 16-byte piece of the input
    auto request = [&](int tile_id) {
        int line, odd;
        int64_t i0, base_e;
        geometry(tile_id, line, i0, base_e, odd);
        const TIn *__restrict__ in = reinterpret_cast<const TIn *>(a.in) + (int64_t)line * a.in_frames * 2;
        fast = 0;
#pragma unroll
        for (int k = 0; k < kPairVecs; ++k) {
            const int pr = tid + k * nthr;
            const int64_t g = base_e + 2 * pr;
            if (pr < npairs && t.vec_ok && g >= 0 && g + 1 < a.in_frames) {
                pre[k].load(in + g * 2);
                fast |= 1u << k;
            }
        }
    };
    auto deposit = [&](int tile_id, double *dst) {
        int line, odd;
        int64_t i0, base_e;
        geometry(tile_id, line, i0, base_e, odd);
        const TIn *__restrict__ in = reinterpret_cast<const TIn *>(a.in) + (int64_t)line * a.in_frames * 2;
        const double *__restrict__ hist = a.hist + (int64_t)line * H * 2;
#pragma unroll
        for (int k = 0; k < kPairVecs; ++k) {
            const int pr = tid + k * nthr;
            if (pr >= npairs)
                continue;
            double c0[2], c1[2];
            if (fast & (1u << k)) {
                pre[k].widen(c0, c1);
            } else {
                const int64_t g = base_e + 2 * pr;
                c0[0] = frame_value(in, hist, g, 0);
                c1[0] = frame_value(in, hist, g, 1);
                c0[1] = frame_value(in, hist, g + 1, 0);
                c1[1] = frame_value(in, hist, g + 1, 1);
            }
            *reinterpret_cast<double2 *>(dst + 2 * pr) = double2{c0[0], c0[1]};
            *reinterpret_cast<double2 *>(dst + t.plane + 2 * pr) = double2{c1[0], c1[1]};
        }
    };

    int cur = 0;
    if ((int)blockIdx.x < ntiles && PH_RS_ABLATE != 3)
        request((int)blockIdx.x);
    load_taps();  // (behind the first window's requests: both fly together)
    if ((int)blockIdx.x < ntiles && PH_RS_ABLATE != 3)
        deposit((int)blockIdx.x, bufs);
    // (Measured and not kept: results parked in LDS and stored after the next window's deposit, so
    // that no store sits between the window's requests and their use -- vmcnt counts stores too and
    // the compiler otherwise waits for the requests before the tap loops.  33.9 us against 31.9: the
    // four workgroups of a CU already cover each other's loads.)
    __syncthreads();
    PH_RS_STAMP(0);
    for (int tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
        const int next = tile_id + (int)gridDim.x;
        if (next < ntiles && PH_RS_ABLATE != 3)
            request(next);  // flies under this tile's tap loops
        PH_RS_STAMP(1);

        int line, odd;
        int64_t i0, base_e;
        geometry(tile_id, line, i0, base_e, odd);
        // outputs [lo, nout) of the tile are the call's (lo > 0 only in a Line's first tile)
        const int nout = (int)min((int64_t)t.tile_out, a.out_frames - i0);
        const int lo = i0 < 0 ? (int)-i0 : 0;
        TOut *__restrict__ out = reinterpret_cast<TOut *>(a.out) + ((int64_t)line * a.out_cap + i0) * 2;
        const double *P0 = bufs + (size_t)cur * 2 * t.plane;
        typedef __attribute__((address_space(3))) const double *lds_ptr;
        if (computing) {
            for (int s = 0; s < t.steps; ++s) {
                const int mlA = s * 2 * t.ql + 2 * tid;
                if (mlA >= nout)
                    break;
                if (mlA + 1 < lo)
                    continue;
                // plane cell of X[0] = x[nfirst + nrelA + DMIN + 1]; X[pos] sits pos cells below
                const int idx0 = H + kPairPad + odd + (int)nA + s * t.adv + DMIN + 1;
                const unsigned lo0 = (unsigned)(uintptr_t)(lds_ptr)(P0 + idx0 - (NPOS - 1));
                const unsigned lo1 = lo0 + 8u * (unsigned)t.plane;
                double accA0 = 0.0, accA1 = 0.0, accB0 = 0.0, accB1 = 0.0;
                double v0[2][G], v1[2][G];
#if PH_RS_ABLATE == 2
                for (int i = 0; i < G; ++i)
                    v0[0][i] = v0[1][i] = v1[0][i] = v1[1][i] = (double)(lo0 + lo1 + i);
#else
                lds_read_pos<0, NPOS>(v0[0], lo0, std::make_integer_sequence<int, G>{});
                lds_read_pos<0, NPOS>(v1[0], lo1, std::make_integer_sequence<int, G>{});
#endif
                for_each_const(std::make_integer_sequence<int, NG>{}, [&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    if constexpr (PH_RS_ABLATE == 1 || PH_RS_ABLATE == 2) {
                        if constexpr (PH_RS_ABLATE == 1)
                            return;
                    } else if constexpr (g + 1 < NG) {
                        lds_read_pos<(g + 1) * G, NPOS>(v0[(g + 1) & 1], lo0, std::make_integer_sequence<int, G>{});
                        lds_read_pos<(g + 1) * G, NPOS>(v1[(g + 1) & 1], lo1, std::make_integer_sequence<int, G>{});
                        lds_wait_cnt<2 * G>();
                    } else {
                        lds_wait_cnt<0>();
                    }
                    lds_pin(v0[g & 1]);
                    lds_pin(v1[g & 1]);
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        const int pos = g * G + i;
                        const double x0 = v0[g & 1][i], x1 = v1[g & 1][i];
                        if (pos <= TT) {  // output B
                            const double b0 = __builtin_fma(hB[pos < TT + 1 ? pos : 0], x0, accB0);
                            const double b1 = __builtin_fma(hB[pos < TT + 1 ? pos : 0], x1, accB1);
                            if (pos == 0) {
                                accB0 = e1 ? b0 : accB0;
                                accB1 = e1 ? b1 : accB1;
                            } else if (pos == TT) {
                                accB0 = e1 ? accB0 : b0;
                                accB1 = e1 ? accB1 : b1;
                            } else {
                                accB0 = b0;
                                accB1 = b1;
                            }
                        }
                        if (pos >= DMIN + 1 && pos <= DMIN + TT) {  // output A
                            accA0 = __builtin_fma(hA[pos - DMIN - 1 < TT && pos >= DMIN + 1 ? pos - DMIN - 1 : 0], x0, accA0);
                            accA1 = __builtin_fma(hA[pos - DMIN - 1 < TT && pos >= DMIN + 1 ? pos - DMIN - 1 : 0], x1, accA1);
                        }
                    }
                    // this group's fma stay ahead of the reads of the group after next (the sums are
                    // otherwise sunk below every read: all W x 2 values live at once): the accumulators
                    // pass through the ordered asm stream, and the scheduler may not cross the barrier
                    asm volatile("" : "+v"(accA0), "+v"(accA1), "+v"(accB0), "+v"(accB1));
                    __builtin_amdgcn_sched_barrier(0);
                });
                TOut *__restrict__ o = out + (int64_t)mlA * 2;
                if (PH_RS_ABLATE == 4 && accA0 + accA1 + accB0 + accB1 != 12345.678)
                    continue;
                if (mlA >= lo) {
                    o[0] = (TOut)accA0;
                    o[1] = (TOut)accA1;
                }
                if (mlA + 1 < nout) {
                    o[2] = (TOut)accB0;
                    o[3] = (TOut)accB1;
                }
            }
        }
        PH_RS_STAMP(2);
        if (next < ntiles && PH_RS_ABLATE != 3)
            deposit(next, bufs + (size_t)(cur ^ 1) * 2 * t.plane);
        PH_RS_STAMP(3);
        __syncthreads();
        PH_RS_STAMP(4);
        cur ^= 1;
    }
#ifdef PH_RS_PROF
    if (t.prof && (threadIdx.x & 63) == 0) {
        unsigned long long *dst = t.prof + ((size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 5;
        for (int i = 0; i < 5; ++i)
            dst[i] = rsprof[i];
    }
#endif
}

// ---- the same two-outputs-per-lane form with NO workgroup: every wave on its own --------------------------------
// What the pair kernel spends outside its tap loops (profiles/r03_resampler_phase_profile.txt: 47 % of a wave's time
// -- tap table and first window per workgroup, request / deposit / barrier per tile, the third wave of every
// workgroup half empty: 160 computing lanes in 192) comes from tiling the Line into workgroup-sized tiles.  Nothing in
// the arithmetic needs a workgroup: a lane's two outputs read 26 frames, a wave's 64 lanes read ~142 consecutive
// frames (1.1 KB of a float32 stereo stream).  Here a WAVE stages exactly that window for itself -- two or three
// 16-byte pieces per lane, converted once, into its own two-channel float64 planes in LDS -- and runs the pair
// kernel's tap loop on it.  LDS operations of one wave execute in order, so no barrier exists anywhere; the next
// step's pieces are requested before the tap loop and deposited after it into the other pair of planes.
// All 64 lanes compute.  A lane's taps must not change from step to step: G waves side by side cover 128 G outputs, a
// whole number of periods of the phase pattern (G = up / gcd(up, 128): 5 for 160), wave w takes slot w mod G of
// groups w / G, w / G + W / G, ... (W waves in the launch, a multiple of G), and the [G][2 T + 1][64] tap table gives
// every slot its lanes' taps in one coalesced load per tap.  Same operations in the same order per output as every
// other form: bit for bit the oracle's.
struct WaveArgs {
    ResampleArgs r;
    int G;          // waves per group (slots)
    int adv;        // input frames a group advances: 128 G down / up
    int plane;      // plane stride (doubles)
    int pieces;     // 2-frame pieces a wave's window holds (<= kPairVecs * 64)
    int nvec;       // pieces every lane stages per step: ceil(pieces / 64)
    int lead;       // virtual outputs ahead of the call's first one: groups start where the phase is 0
    int groups_per_line;
    int64_t nb;     // input frame (relative to this call) the first virtual output reads
    int small_in;   // (unused since round 6: a step's raw buffer begins at its window, so a Line may be of any length)
    const double *ptaps;  // [G][2 T + 1][64]
    unsigned long long *prof;  // PH_RS_PROF builds: [wave][5] s_memtime ticks per phase
};

template <typename TIn, typename TOut, int TT, int DMIN>
__global__ void __launch_bounds__(64) resample_wave_kernel(const WaveArgs t)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *bufs = reinterpret_cast<double *>(smem_raw);  // [2 steps][2 channels][plane]
    const ResampleArgs &a = t.r;
    const int lane = (int)threadIdx.x;
    constexpr int H = TT - 1;
    constexpr int W = TT + DMIN + 1;  // frames a pair reads
    constexpr int G4 = 4, NG = (W + G4 - 1) / G4, NPOS = NG * G4;
    const int nwaves = (int)gridDim.x;
    // Waves that stand side by side in the stream on ONE XCD (workgroup b runs on XCD b % 8): wave xb of the launch's
    // order is the (b / 8)-th workgroup of XCD b % 8, so that XCD x holds the waves [x n / 8, (x + 1) n / 8) -- the G slots of
    // a group and the groups next to it, whose windows overlap by the filter's length, share their overlap through that
    // XCD's L2 (round 5 dealt them round-robin over the XCDs: every overlap was fetched twice, 1.21 x the algorithmic bytes)
    const int nb8 = (int)gridDim.x;
    const int xb = nb8 % 8 == 0 ? ((int)blockIdx.x % 8) * (nb8 / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
    const int slot = xb % t.G;
    const int ngroups = t.groups_per_line * a.lines, gstride = nwaves / t.G;
#ifdef PH_RS_PROF
    unsigned long long rsprof[5] = {}, rslast = __builtin_amdgcn_s_memtime();
#endif

    // ---- the lane's two outputs (fixed for the whole launch): pair `q` of the group
    const unsigned q = 64u * (unsigned)slot + (unsigned)lane;
    const unsigned ttA = 2u * q * (unsigned)a.down, nA = ttA / (unsigned)a.up;
    const unsigned nB = (ttA + (unsigned)a.down) / (unsigned)a.up;
    const unsigned nA0 = (2u * 64u * (unsigned)slot * (unsigned)a.down) / (unsigned)a.up;  // lane 0's: the window's anchor
    const bool e1 = (int)(nB - nA) - DMIN != 0;  // e = 1: output B reads X[0 .. T-1]; e = 0: X[1 .. T]
    const int nrel = (int)(nA - nA0);
    double hA[TT], hB[TT + 1];
    auto load_taps = [&]() {
        const double *__restrict__ tp = t.ptaps + (size_t)slot * (2 * TT + 1) * 64 + lane;
#pragma unroll
        for (int j = 0; j < TT; ++j)
            hA[j] = tp[j * 64];
#pragma unroll
        for (int i = 0; i <= TT; ++i)
            hB[i] = tp[(TT + i) * 64];
    };
    // Everything about a step that does not depend on the lane -- scalar registers, computed ONCE per step (for the
    // step after the one being computed) and carried: the PMC counts of the first version of this kernel
    // (profiles/r05_resampler_pmc.txt: 234 vector and 161 scalar instructions per wave and step around 98 fma) said
    // that the vector pipe was the bound and that more than half of its instructions were not the filter's.
    // (plain scalars, not a struct: a struct carried around the loop went through 20 bytes of scratch per lane)
    //   line, odd;  m0: the wave's first output, relative to the call's first (negative: virtual outputs);
    //   base_e: the window's first staged frame, relative to the call's input (even);
    //   interior: the window starts inside the call's input -- staged by raw buffer loads, no per-lane tests (a Line of
    //             any length: the step's buffer begins at the window's first frame);
    //   full: every output of the wave belongs to the call -- one 16-byte store per lane, no per-lane tests
#define PH_RW_STEP(P, GID)                                                                            \
    do {                                                                                              \
        P##line = a.lines == 1 ? 0 : (GID) / t.groups_per_line;                                       \
        const int g_ = (GID)-P##line * t.groups_per_line;                                             \
        P##m0 = (int64_t)g_ * (128 * t.G) + 128 * slot - t.lead;                                      \
        const int64_t base_ = t.nb + (int64_t)g_ * t.adv + (int64_t)nA0 - H - kPairPad;              \
        P##base_e = base_ & ~(int64_t)1; /* even: 16-byte pieces of the input, 16-byte plane cells */ \
        P##odd = (int)(base_ - P##base_e);                                                            \
        P##interior = P##base_e >= 0;                                                                 \
        P##full = P##m0 >= 0 && P##m0 + 128 <= a.out_frames;                                          \
    } while (0)
    int cs_line, cs_odd, ns_line = 0, ns_odd = 0;
    int64_t cs_m0, cs_base_e, ns_m0 = 0, ns_base_e = 0;
    bool cs_interior, cs_full, ns_interior = false, ns_full = false;
    auto bytes31 = [](int64_t n) { return (int)(n < 0x7FFFFFFF ? n : 0x7FFFFFFF); };

    // ---- staging, interior steps: what is left of the Line's input from the window's start on is ONE raw buffer (made
    // per step from scalars), a lane's offset is the same every step (k * 64 pieces further: the immediate offset), and a piece past
    // the end of the input reads as zero by the buffer's own range check (silence past the end, as the slow path's).
    // Every lane stages t.nvec pieces, no masks: the planes hold 128 t.nvec cells.
    PairRaw<TIn> pre[kPairVecs];
    // (a lane whose k-th piece lies beyond the window asks for an offset no buffer has: the range check answers zero
    // and nothing is fetched -- without it every wave read 128 pieces where its window holds 73: 1.65 x the input)
    unsigned voffk[kPairVecs];
#pragma unroll
    for (int k = 0; k < kPairVecs; ++k)
        voffk[k] = lane + 64 * k < t.pieces ? (unsigned)(lane + 64 * k) * PairRaw<TIn>::kBytes : 0x7FFFFF00u;
    auto request = [&](int line, int64_t base_e) {
        // The buffer BEGINS at the window's first frame and holds what is left of the Line's input: the range check
        // compares the vector offset (+ immediate) alone with the record count -- a scalar offset is added to the address
        // unchecked, so a window start carried there would let a step's last pieces read past the Line's input (the next
        // Line's samples, or past the allocation).  Scalar arithmetic per step, nothing per lane.
        const int64_t left = a.in_frames - base_e;
        const TIn *in = reinterpret_cast<const TIn *>(a.in) + ((int64_t)line * a.in_frames + (left > 0 ? base_e : 0)) * 2;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<TIn *>(in), 0, left > 0 ? bytes31(left * 2 * (int64_t)sizeof(TIn)) : 0, 0x00020000);
#pragma unroll
        for (int k = 0; k < kPairVecs; ++k)
            if (k < t.nvec)
                pre[k].load_buf(rs, voffk[k], 0u);
    };
    auto deposit = [&](double *dst) {
#pragma unroll
        for (int k = 0; k < kPairVecs; ++k)
            if (k < t.nvec) {
                double c0[2], c1[2];
                pre[k].widen(c0, c1);
                *reinterpret_cast<double2 *>(dst + 2 * (lane + k * 64)) = double2{c0[0], c0[1]};
                *reinterpret_cast<double2 *>(dst + t.plane + 2 * (lane + k * 64)) = double2{c1[0], c1[1]};
            }
    };
    // ---- staging, a Line's first steps (the window reaches into the history): frame by frame, on the spot
    auto frame_value = [&](const TIn *__restrict__ in, const double *__restrict__ hist, int64_t g, int c) -> double {
        if (g >= 0)
            return g < a.in_frames ? (double)in[g * 2 + c] : 0.0;
        return g >= -(int64_t)H ? hist[(g + H) * 2 + c] : 0.0;
    };
    auto stage_slow = [&](int line, int64_t base_e, double *dst) {
        const TIn *__restrict__ in = reinterpret_cast<const TIn *>(a.in) + (int64_t)line * a.in_frames * 2;
        const double *__restrict__ hist = a.hist + (int64_t)line * H * 2;
        for (int k = 0; k < t.nvec; ++k) {
            const int pr = lane + k * 64;
            const int64_t g = base_e + 2 * pr;
            const double c00 = frame_value(in, hist, g, 0), c10 = frame_value(in, hist, g, 1);
            const double c01 = frame_value(in, hist, g + 1, 0), c11 = frame_value(in, hist, g + 1, 1);
            *reinterpret_cast<double2 *>(dst + 2 * pr) = double2{c00, c01};
            *reinterpret_cast<double2 *>(dst + t.plane + 2 * pr) = double2{c10, c11};
        }
    };
    // the planes of a step were written by this wave's own LDS stores: the LDS serves a wave's operations in order,
    // the compiler must keep them in order too (a compiler barrier only: a fence would also wait for the step's global
    // stores, a thousand cycles, every step)
    auto order = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };

    int gid = xb / t.G;
    if (gid >= ngroups)
        return;
    PH_RW_STEP(cs_, gid);
    int cur = 0;
    if (cs_interior)
        request(cs_line, cs_base_e);
    load_taps();  // (behind the first window's requests: both fly together)
    if (cs_interior)
        deposit(bufs);
    else
        stage_slow(cs_line, cs_base_e, bufs);
    order();
    // The taps are IN their registers before the loop begins.  (Left to the compiler, the wait for their loads sits at
    // their first use -- inside the loop, in the middle of the tap loop, as a wait for EVERY outstanding load: each
    // step then waited there for the window it had just requested instead of computing under it.)
#pragma unroll
    for (int j = 0; j < TT; ++j)
        asm volatile("" : "+v"(hA[j]));
#pragma unroll
    for (int i = 0; i <= TT; ++i)
        asm volatile("" : "+v"(hB[i]));
    PH_RS_STAMP(0);
    typedef __attribute__((address_space(3))) const double *lds_ptr;
    // A step's results are STORED AT THE TOP OF THE NEXT STEP, ahead of that step's window requests: the memory counter
    // counts stores and loads alike and in order, so the wait for the requested pieces never waits for a younger store.
    struct __attribute__((packed, aligned(sizeof(TOut)))) Quad {
        TOut v[4];
    };
    Quad pend{};
    TOut *po = nullptr;
    int pflags = 0;  // bit 0: output A belongs to the call, bit 1: output B does; 4: every lane's both do (wave-uniform)
    auto flush = [&]() {
        if (pflags & 4) {  // (all but a Line's first and last wave: one store per lane, no tests)
            *reinterpret_cast<Quad *>(po) = pend;
        } else {
            if (pflags & 1) {
                po[0] = pend.v[0];
                po[1] = pend.v[1];
            }
            if (pflags & 2) {
                po[2] = pend.v[2];
                po[3] = pend.v[3];
            }
        }
        pflags = 0;
    };
    for (;;) {
        flush();
        const int next = gid + gstride;
        const bool has_next = next < ngroups;
        if (has_next) {
            PH_RW_STEP(ns_, next);
            if (ns_interior)
                request(ns_line, ns_base_e);  // flies under this step's tap loop
        }
        PH_RS_STAMP(1);

        // ---- the tap loop: every lane, also one whose outputs lie outside the call (its window is staged like any
        // other; only the stores know)
        const double *P0 = bufs + (size_t)cur * 2 * t.plane;
        {
            // plane cell of X[0] = x[base + H + kPairPad + nrel + DMIN + 1]; X[pos] sits pos cells below
            const int idx0 = H + kPairPad + cs_odd + nrel + DMIN + 1;
            const unsigned lo0 = (unsigned)(uintptr_t)(lds_ptr)(P0 + idx0 - (NPOS - 1));
            const unsigned lo1 = lo0 + 8u * (unsigned)t.plane;
            double accA0 = 0.0, accA1 = 0.0, accB0 = 0.0, accB1 = 0.0;
            double v0[2][G4], v1[2][G4];
            lds_read_pos<0, NPOS>(v0[0], lo0, std::make_integer_sequence<int, G4>{});
            lds_read_pos<0, NPOS>(v1[0], lo1, std::make_integer_sequence<int, G4>{});
            for_each_const(std::make_integer_sequence<int, NG>{}, [&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if constexpr (g + 1 < NG) {
                    lds_read_pos<(g + 1) * G4, NPOS>(v0[(g + 1) & 1], lo0, std::make_integer_sequence<int, G4>{});
                    lds_read_pos<(g + 1) * G4, NPOS>(v1[(g + 1) & 1], lo1, std::make_integer_sequence<int, G4>{});
                    lds_wait_cnt<2 * G4>();
                } else {
                    lds_wait_cnt<0>();
                }
                lds_pin(v0[g & 1]);
                lds_pin(v1[g & 1]);
#pragma unroll
                for (int i = 0; i < G4; ++i) {
                    const int pos = g * G4 + i;
                    const double x0 = v0[g & 1][i], x1 = v1[g & 1][i];
                    if (pos <= TT) {  // output B
                        const double b0 = __builtin_fma(hB[pos < TT + 1 ? pos : 0], x0, accB0);
                        const double b1 = __builtin_fma(hB[pos < TT + 1 ? pos : 0], x1, accB1);
                        if (pos == 0) {
                            accB0 = e1 ? b0 : accB0;
                            accB1 = e1 ? b1 : accB1;
                        } else if (pos == TT) {
                            accB0 = e1 ? accB0 : b0;
                            accB1 = e1 ? accB1 : b1;
                        } else {
                            accB0 = b0;
                            accB1 = b1;
                        }
                    }
                    if (pos >= DMIN + 1 && pos <= DMIN + TT) {  // output A
                        accA0 = __builtin_fma(hA[pos - DMIN - 1 < TT && pos >= DMIN + 1 ? pos - DMIN - 1 : 0], x0, accA0);
                        accA1 = __builtin_fma(hA[pos - DMIN - 1 < TT && pos >= DMIN + 1 ? pos - DMIN - 1 : 0], x1, accA1);
                    }
                }
                // (as in the pair kernel: this group's fma stay ahead of the reads of the group after next)
                asm volatile("" : "+v"(accA0), "+v"(accA1), "+v"(accB0), "+v"(accB1));
                __builtin_amdgcn_sched_barrier(0);
            });
            const int64_t mA = cs_m0 + 2 * lane;  // the lane's output A, relative to the call's first
            po = reinterpret_cast<TOut *>(a.out) + ((int64_t)cs_line * a.out_cap + mA) * 2;
            pend.v[0] = (TOut)accA0;
            pend.v[1] = (TOut)accA1;
            pend.v[2] = (TOut)accB0;
            pend.v[3] = (TOut)accB1;
            pflags = cs_full ? 4 : ((mA >= 0 && mA < a.out_frames ? 1 : 0) | (mA + 1 >= 0 && mA + 1 < a.out_frames ? 2 : 0));
        }
        order();  // (this step's reads of the planes are done -- lgkmcnt(0) above; the next step's window goes to the OTHER planes)
        PH_RS_STAMP(2);
        if (!has_next)
            break;
        if (ns_interior)
            deposit(bufs + (size_t)(cur ^ 1) * 2 * t.plane);
        else
            stage_slow(ns_line, ns_base_e, bufs + (size_t)(cur ^ 1) * 2 * t.plane);
        order();
        PH_RS_STAMP(3);
        cur ^= 1;
        cs_line = ns_line;
        cs_odd = ns_odd;
        cs_m0 = ns_m0;
        cs_base_e = ns_base_e;
        cs_interior = ns_interior;
        cs_full = ns_full;
        gid = next;
    }
    flush();
#undef PH_RW_STEP
#ifdef PH_RS_PROF
    if (t.prof && lane == 0) {
        unsigned long long *dst = t.prof + (size_t)blockIdx.x * 5;
        for (int i = 0; i < 5; ++i)
            dst[i] = rsprof[i];
    }
#endif
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
        host_proto_.assign(proto, proto + n);
        // the row form's table (resampler_rows.hip): row i = the taps of output i of a period, j ascending
        if (cfg.channels >= 2 && cfg.channels <= 128 && cfg.channels % 2 == 0 &&
            (T == 8 || T == 12 || T == 16 || T == 24)) {
            std::vector<double> rt(n);
            for (int i = 0; i < up; ++i) {
                const int ph = (int)(((int64_t)i * down) % up);
                for (int j = 0; j < T; ++j)
                    rt[(size_t)i * T + j] = proto[(size_t)ph + (size_t)j * up];
            }
            PH_TRY(row_taps_.alloc(sizeof(double) * n));
            PH_HIP(hipMemcpy(row_taps_.p, rt.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        }
        hist_bytes_ = sizeof(double) * (size_t)cfg.lines * (size_t)(T - 1) * (size_t)cfg.channels;
        PH_TRY(hist_[0].alloc(hist_bytes_));
        PH_TRY(hist_[1].alloc(hist_bytes_));
        int n_cus = 0;
        if (hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, cfg.device) == hipSuccess && n_cus > 0)
            cus_ = n_cus;
        else
            (void)hipGetLastError();
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
                              !PH_ENV_AB("PIPE_HIP_RESAMPLE_LDS_TAPS");
        const int q = reg_taps ? up_ * (kThreads / up_) : kThreads;
        // (taps in registers: as many waves as hold the q computing lanes, e.g. 3 for 160)
        const int threads = reg_taps ? (q + 63) / 64 * 64 : kThreads;
        // outputs per tile: kOutTile, halved until the workgroup's LDS -- planes, and the table unless
        // the taps sit in registers -- fits 64 KB (wide Lines: 8 channels at 160 x 24 stay on this
        // kernel: 74 us where the gather kernel takes 213) -- never below one output per computing lane
        // float32 streams with the taps in registers keep the window as float32 channel pairs
        // (from four channels on: with two the float64 planes are faster, 51.5 against 59.7 us)
        const bool f32win = reg_taps && in_dtype == PIPE_HIP_F32 && cfg.channels % 2 == 0 && cfg.channels >= 4 && T_ % 4 == 0 &&
                            !PH_ENV_AB("PIPE_HIP_RESAMPLE_F64_PLANES");
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
        // 2-channel streams with the taps in registers: two adjacent outputs per lane share their
        // window reads (resample_pair_kernel)
        if (total > 0 && n_out > 0 && launch_rows(a, in_dtype, out_dtype, s))
            return finish_call(d_in, in_dtype, in_frames, n_out, a, s);
        if (total > 0 && n_out > 0 && launch_wave(a, in_dtype, out_dtype, reg_taps, s))
            return finish_call(d_in, in_dtype, in_frames, n_out, a, s);
        if (total > 0 && n_out > 0 && launch_pair(a, in_dtype, out_dtype, reg_taps, s))
            return finish_call(d_in, in_dtype, in_frames, n_out, a, s);
        const bool big_lds = lds > 64 * 1024;
        const bool tiled = lds <= (reg_taps ? (size_t)64 * 1024 : kBigLds) && n_out > 0 && !PH_ENV_AB("PIPE_HIP_RESAMPLE_GATHER");
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
        return finish_call(d_in, in_dtype, in_frames, n_out, a, s);
    }

private:
    // the history for the next call, the stream's counters
    int finish_call(const void *d_in, int in_dtype, int64_t in_frames, int64_t n_out, const ResampleArgs &a, hipStream_t s)
    {
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

    // the lanes' taps of the pair kernel, [2 T + 1][ql]: rows 0 .. T-1 output A's taps, rows T .. 2T
    // output B's shifted by the lane's e (resample_pair_kernel); made once per (ql, dmin)
    bool ensure_pair_taps(int ql, int dmin)
    {
        if (pair_taps_.p && pair_ql_ == ql)
            return true;
        const size_t n = (size_t)(2 * T_ + 1) * (size_t)ql;
        std::vector<double> h(n, 0.0);
        for (int l = 0; l < ql; ++l) {
            const int64_t ttA = (int64_t)2 * l * down_, ttB = ttA + down_;
            const int nA = (int)(ttA / up_), pA = (int)(ttA % up_), nB = (int)(ttB / up_), pB = (int)(ttB % up_);
            const int e = (nB - nA) - dmin;
            for (int j = 0; j < T_; ++j)
                h[(size_t)j * ql + l] = host_proto_[(size_t)pA + (size_t)j * up_];
            for (int i = 0; i <= T_; ++i) {
                const int j = i - 1 + e;
                h[(size_t)(T_ + i) * ql + l] = j >= 0 && j < T_ ? host_proto_[(size_t)pB + (size_t)j * up_] : 0.0;
            }
        }
        if (pair_taps_.alloc(sizeof(double) * n) != PIPE_HIP_OK)
            return false;
        if (hipMemcpy(pair_taps_.p, h.data(), sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            pair_taps_.release();
            return false;
        }
        pair_ql_ = ql;
        return true;
    }

    // the wave kernel's taps, [G][2 T + 1][64]: slot j, lane l is pair q = 64 j + l of the group
    bool ensure_wave_taps(int G, int dmin)
    {
        if (wave_taps_.p && wave_G_ == G)
            return true;
        const size_t rows = (size_t)(2 * T_ + 1);
        std::vector<double> h((size_t)G * rows * 64u, 0.0);
        for (int j = 0; j < G; ++j)
            for (int l = 0; l < 64; ++l) {
                const int64_t qq = (int64_t)64 * j + l;
                const int64_t ttA = 2 * qq * down_, ttB = ttA + down_;
                const int nA = (int)(ttA / up_), pA = (int)(ttA % up_), nB = (int)(ttB / up_), pB = (int)(ttB % up_);
                const int e = (nB - nA) - dmin;
                double *row = h.data() + (size_t)j * rows * 64u + (size_t)l;
                for (int k = 0; k < T_; ++k)
                    row[(size_t)k * 64u] = host_proto_[(size_t)pA + (size_t)k * up_];
                for (int i = 0; i <= T_; ++i) {
                    const int k = i - 1 + e;
                    row[(size_t)(T_ + i) * 64u] = k >= 0 && k < T_ ? host_proto_[(size_t)pB + (size_t)k * up_] : 0.0;
                }
            }
        if (wave_taps_.alloc(sizeof(double) * h.size()) != PIPE_HIP_OK)
            return false;
        if (hipMemcpy(wave_taps_.p, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            wave_taps_.release();
            return false;
        }
        wave_G_ = G;
        return true;
    }

    // true when the row kernel took the call (resampler_rows.hip): an even channel count, T one of the
    // specialised sizes, a stream long enough that workgroups of 64 / C rows (B periods of the phase pattern each) fill the chip
    bool launch_rows(const ResampleArgs &a, int in_dtype, int out_dtype, hipStream_t s)
    {
        if (!row_taps_.p || knobs.resample_rows_min_blocks < 0 || in_dtype != PIPE_HIP_F32 || out_dtype != PIPE_HIP_F32)
            return false;
        // Two channels: measured level with the wave kernel (0.29 / 0.37 of HBM streaming against 0.30 / 0.39 at 1 and 4
        // Lines of 1024 buffers; 4 channels 0.34 against 0.21, 8 channels 0.31 - 0.37 against 0.16 - 0.23): stereo
        // streams the wave kernel takes keep it unless the threshold is set explicitly; the ones it does not take (a phase
        // pattern that more than 160 waves side by side would cover, e.g. up = 161: the workgroup-tiled pair kernel's,
        // 0.245 at the bench shape) come here
        if (cfg.channels == 2 && !knobs.resample_rows_stereo && wave_takes())
            return false;
        const int es = (int)dtype_size(in_dtype);
        if (reinterpret_cast<uintptr_t>(a.in) % es != 0 || reinterpret_cast<uintptr_t>(a.out) % es != 0)
            return false;
        const int C = cfg.channels;
        const int rpb = 64 / (C / 2);
        const int big = up_ > down_ ? up_ : down_;
        int B = big >= 144 ? 1 : 144 / big;
        // A row's window refill reads the T - 1 frames ahead of the row out of the row directly ABOVE it (the kernel stages one
        // row above a block, no more): a row must hold them.  Upsamplers by a large factor (up = 8, down = 1: 18 frames a row
        // by the rule above; 160 / 3: 3 frames) take more periods a row.
        if ((int64_t)B * down_ < T_ - 1)
            B = (T_ - 1 + down_ - 1) / down_;
        if ((int64_t)B * up_ >= 32768)
            return false;
        rows::Args t{};
        t.C = C;
        t.rpb = rpb;
        t.pair_rcp = (unsigned)((65536 + C / 2 - 1) / (C / 2));
        t.row_out = B * up_;
        t.row_in = B * down_;
        // LDS rows: the row's samples, room for a last 16-byte piece that overshoots, rows 16 bytes x an odd number apart
        // (16-byte pieces in and out; a frame's 8-byte reads of 32 rows then fall two to a bank, once per frame: nothing)
        auto stride_for = [&](int n) {
            const int unit = C > 4 ? (C % 4 ? 2 * C : C) : 4;  // (rows a multiple of 16 bytes apart; C channels: the rows of a 32-lane group max(4, C) dwords x distinct numbers apart)
            int st = (n + 4 + unit - 1) / unit * unit;
            while ((st / unit) % 2 == 0)
                st += unit;
            return st;
        };
        if (t.row_in < T_ - 1)
            return false;
        t.in_stride = stride_for(t.row_in * C);
        t.out_off = (int)(((int64_t)(rpb + 1) * t.in_stride * es + 15) / 16 * 16);  // (two buffers of a block's input ...
        int64_t lds = (int64_t)2 * t.out_off;
        if (lds > 160 * 1024) {  // ... or one, and a second barrier a block, when the rows are that long)
            lds = t.out_off;
            t.out_off = 0;
        }
        if (lds > 160 * 1024)
            return false;
        t.lds_bytes = (int)lds;
        const int pei = 16 / es, npieces = (t.row_in * C + pei - 1) / pei;
        t.piece_magic = (unsigned)(0x100000000ull / (unsigned)npieces) + 1u;
        const int64_t first_row = a.out_total / t.row_out;
        const int64_t last_row = (a.out_total + a.out_frames - 1) / t.row_out;
        const int64_t rows_per_line = last_row - first_row + 1;
        const int64_t blocks_per_line = (rows_per_line + rpb - 1) / rpb;
        // whole workgroups of rows, and enough of them: the wave kernel keeps the short calls
        // (6 channels and more: from the first block -- one Line x 8 ch x 4 .. 32 buffers 13.3 - 13.7 us against the
        // tiled kernel's 17.7 - 18.6, 6 ch x 16 buffers 13.1 against 16.7, profiles/r06_dispatch_audit.txt; 4 channels: the
        // tiled kernel keeps the short calls, 11.8 against 13.0; PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS set: that count for
        // every channel count)
        const int64_t min_blocks = !knobs.resample_rows_stereo && C >= 6 ? 1 : knobs.resample_rows_min_blocks;
        if (rows_per_line * 4 < blocks_per_line * rpb * 3 || (int64_t)cfg.lines * blocks_per_line < min_blocks)
            return false;
        // 32-bit sample arithmetic in the kernel
        if ((a.in_frames + (int64_t)(rpb + 2) * t.row_in + T_) * C * es >= 0x7FFFFFFF || (a.out_frames + (int64_t)(rpb + 2) * t.row_out) * C >= 0x7FFFFFFF)
            return false;
        // waves of a workgroup = segments of a row: twelve (three a SIMD: the kernel's registers), 8 outputs each at least
        int segs = t.row_out / 8;
        segs = segs > 12 ? 12 : segs < 1 ? 1 : segs;
        if (const char *e = PH_ENV_AB("PIPE_HIP_RESAMPLE_ROWS_SEGS")) {
            segs = std::atoi(e);
            segs = segs > 12 ? 12 : segs < 1 ? 1 : segs;  // (the kernel's launch bound is 12 waves; seg_b has 17 entries)
        }
        const int seg_out = (t.row_out + segs - 1) / segs;
        t.segs = (t.row_out + seg_out - 1) / seg_out;
        if (t.row_out >= 32768)
            return false;
        {
            // Segments of equal length.  (The hardware serves a SIMD's oldest wave first: with equal segments the first wave
            // of every SIMD finishes its tap loops after 16 k ticks, the third after 26.7 k.  Longer segments for the
            // older waves -- weights by groups of four waves, the A/B switch below -- were measured: 1.6 / 1 / 0.6 gains
            // 2 - 5 % on one Line and loses 2 % on eight, stronger weights lose everywhere: not taken.)
            double w[16], sum = 0.0;
            const char *e = PH_ENV_AB("PIPE_HIP_RESAMPLE_ROWS_SEG_WEIGHTS");
            double g3[4] = {1.0, 1.0, 1.0, 1.0};
            if (e)
                std::sscanf(e, "%lf,%lf,%lf,%lf", &g3[0], &g3[1], &g3[2], &g3[3]);
            for (double &g : g3)
                if (!(g > 0.0))
                    g = 1.0;
            for (int k = 0; k < t.segs; ++k) {
                w[k] = g3[k / 4 < 4 ? k / 4 : 3];
                sum += w[k];
            }
            double acc = 0.0;
            t.seg_b[0] = 0;
            for (int k = 0; k < t.segs; ++k) {
                acc += w[k];
                t.seg_b[k + 1] = k + 1 == t.segs ? t.row_out : (int)(acc / sum * t.row_out + 0.5);
                if (t.seg_b[k + 1] < t.seg_b[k])
                    t.seg_b[k + 1] = t.seg_b[k];
            }
        }
        // a thread holds at most 8 pieces of a block's input in registers
        if ((int64_t)(rpb + 1) * npieces > (int64_t)8 * 64 * t.segs || (int64_t)(rpb + 1) * npieces >= 65536)
            return false;
        const int per_cu = (int)((160 * 1024) / lds);
        t.max_groups = cus_ * (per_cu < 1 ? 1 : per_cu);
        t.in = a.in;
        t.out = a.out;
        t.hist = a.hist;
        t.rtaps = static_cast<const double *>(row_taps_.p);
        t.in_frames = a.in_frames;
        t.out_frames = a.out_frames;
        t.out_cap = a.out_cap;
        t.up = up_;
        t.down = down_;
        t.lines = cfg.lines;
        t.T = T_;
        t.blocks_per_line = (int)blocks_per_line;
        t.fb0 = (int)(first_row * t.row_in - a.in_total);
        t.ob0 = (int)(first_row * t.row_out - a.out_total);
        hipEvent_t ev_a = nullptr, ev_b = nullptr;
        if (timer.pair(&ev_a, &ev_b) != PIPE_HIP_OK)
            return false;
        if (!rows::launch(t, in_dtype == PIPE_HIP_F64, out_dtype == PIPE_HIP_F64, s, ev_a, ev_b)) {
            (void)hipGetLastError();
            timer.unpair(ev_a);  // (never recorded: the next kernel takes a pair of its own)
            return false;
        }
        last_kernel = "resample_rows_kernel<f32,f32>";
        return true;
    }

    // the shapes launch_wave accepts (its first tests, without a call at hand)
    bool wave_takes() const
    {
        if (cfg.channels != 2 || !(T_ == 8 || T_ == 12 || T_ == 16 || T_ == 24 || T_ == 32) || up_ > kThreads || down_ / up_ > 1)
            return false;
        int g = up_, b = 128;
        while (b) {
            const int r = g % b;
            g = b;
            b = r;
        }
        return up_ / g <= 160;
    }

    // true when the wave kernel took the call (2 channels, taps in registers, down / up < 2, a group of at most
    // kMaxWaveGroup waves covering whole periods of the phase pattern)
    bool launch_wave(const ResampleArgs &a, int in_dtype, int out_dtype, bool reg_taps, hipStream_t s)
    {
        constexpr int kMaxWaveGroup = 160;
        if (cfg.channels != 2 || !reg_taps || T_ % 4 != 0 || PH_ENV_AB("PIPE_HIP_RESAMPLE_NO_WAVE") || PH_ENV_AB("PIPE_HIP_RESAMPLE_NO_PAIR"))
            return false;
        const int dmin = down_ / up_;
        if (dmin > 1)
            return false;
        int g = up_, b = 128;
        while (b) {
            const int r = g % b;
            g = b;
            b = r;
        }
        const int G = up_ / g;  // 128 G outputs are a multiple of `up`
        if (G > kMaxWaveGroup)
            return false;
        if (reinterpret_cast<uintptr_t>(a.in) % dtype_size(in_dtype) != 0)
            return false;
        WaveArgs t{};
        t.r = a;
        t.G = G;
        t.adv = (int)((int64_t)128 * G * down_ / up_);
        // frames a wave stages: from lane 0's oldest (minus the pad) to lane 63's newest, rounded to pieces
        const int span = (int)(((int64_t)126 * down_) / up_) + 1 + dmin + 1 + (T_ - 1) + kPairPad + 1 + 1;
        t.pieces = (span + 1) / 2 + 1;
        if (t.pieces > kPairVecs * 64)
            return false;
        t.nvec = (t.pieces + 63) / 64;
        t.plane = 128 * t.nvec + 16;  // every lane stages nvec pieces, no masks; stride == 16 (mod 32): the channels' planes on distinct banks
        if (!ensure_wave_taps(G, dmin))
            return false;
        t.ptaps = static_cast<const double *>(wave_taps_.p);
        t.lead = (int)(a.out_total % up_);
        t.nb = (a.out_total - t.lead) / up_ * down_ - a.in_total;
        t.groups_per_line = (int)((t.lead + a.out_frames + 128 * (int64_t)G - 1) / (128 * (int64_t)G));
        t.small_in = a.in_frames * 2 * (int64_t)dtype_size(in_dtype) < 0x7FFFFFFF ? 1 : 0;
        const int64_t ngroups = (int64_t)t.groups_per_line * cfg.lines;
        // twelve waves a CU (three a SIMD: the kernel's registers), whole groups of them
        int64_t waves = ngroups * G;
        const char *wpc = PH_ENV_AB("PIPE_HIP_RESAMPLE_WAVES_PER_CU");  // A/B: resident waves per CU
        int64_t cap = (int64_t)(wpc ? std::atoi(wpc) : 12) * cus_ / G * G;
        {
            // ... and of eight where that costs less than a tenth of them (the XCD-major order of the kernel needs it)
            int64_t l = G;
            while (l % 8 != 0)
                l += G;
            const int64_t cap8 = cap / l * l;
            if (cap8 * 10 >= cap * 9)
                cap = cap8;
        }
        if (waves > cap && cap >= G)
            waves = cap;
        const size_t lds = sizeof(double) * 4 * (size_t)t.plane;
        hipEvent_t ev_a = nullptr, ev_b = nullptr;
        if (timer.pair(&ev_a, &ev_b) != PIPE_HIP_OK)
            return false;
        const dim3 grid((unsigned)waves);
#ifdef PH_RS_PROF
        static DevBuf wprof;
        if (!wprof.p && wprof.alloc(sizeof(unsigned long long) * 5 * 65536) != PIPE_HIP_OK)
            return false;
        t.prof = static_cast<unsigned long long *>(wprof.p);
#endif
#define PH_RW(TI, TO, TTV, DM) hipExtLaunchKernelGGL((resample_wave_kernel<TI, TO, TTV, DM>), grid, dim3(64), lds, s, ev_a, ev_b, 0, t)
#define PH_RW_T(TI, TO, DM)                 \
    switch (T_) {                           \
    case 8: PH_RW(TI, TO, 8, DM); break;    \
    case 12: PH_RW(TI, TO, 12, DM); break;  \
    case 16: PH_RW(TI, TO, 16, DM); break;  \
    case 24: PH_RW(TI, TO, 24, DM); break;  \
    default: PH_RW(TI, TO, 32, DM); break;  \
    }
#define PH_RW_D(TI, TO)      \
    if (dmin == 0) {         \
        PH_RW_T(TI, TO, 0)   \
    } else {                 \
        PH_RW_T(TI, TO, 1)   \
    }
        if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32) {
            PH_RW_D(float, float)
            last_kernel = "resample_wave_kernel<f32,f32>";
        } else if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F64) {
            PH_RW_D(double, double)
            last_kernel = "resample_wave_kernel<f64,f64>";
        } else if (in_dtype == PIPE_HIP_F32) {
            PH_RW_D(float, double)
            last_kernel = "resample_wave_kernel<f32,f64>";
        } else {
            PH_RW_D(double, float)
            last_kernel = "resample_wave_kernel<f64,f32>";
        }
#undef PH_RW_D
#undef PH_RW_T
#undef PH_RW
#ifdef PH_RS_PROF
        {
            static int launches = 0;
            if (++launches == 15 && waves <= 65536) {
                (void)hipStreamSynchronize(s);
                std::vector<unsigned long long> h((size_t)waves * 5);
                (void)hipMemcpy(h.data(), t.prof, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
                static const char *names[5] = {"taps + first window", "request next window", "tap loop + stores", "deposit next window", "-"};
                double sum[5] = {}, tot = 0, mx = 0;
                for (int64_t w = 0; w < waves; ++w) {
                    double wt = 0;
                    for (int i = 0; i < 5; ++i) {
                        sum[i] += (double)h[(size_t)w * 5 + i];
                        wt += (double)h[(size_t)w * 5 + i];
                    }
                    mx = wt > mx ? wt : mx;
                }
                for (double v : sum)
                    tot += v;
                std::fprintf(stderr, "[resampler wave prof] s_memtime ticks per wave, %lld waves, %lld groups of %d, %.1f steps a wave\n",
                             (long long)waves, (long long)ngroups, G, (double)ngroups * G / (double)waves);
                for (int i = 0; i < 4; ++i)
                    std::fprintf(stderr, "[resampler wave prof]   %-22s %9.1f  %5.1f %%\n", names[i], sum[i] / (double)waves, 100.0 * sum[i] / tot);
                std::fprintf(stderr, "[resampler wave prof]   %-22s %9.1f (slowest wave %9.1f)\n", "total", tot / (double)waves, mx);
            }
        }
#endif
        return hipGetLastError() == hipSuccess;
    }

    // true when the pair kernel took the call
    bool launch_pair(const ResampleArgs &a, int in_dtype, int out_dtype, bool reg_taps, hipStream_t s)
    {
        if (cfg.channels != 2 || !reg_taps || T_ % 4 != 0 || PH_ENV_AB("PIPE_HIP_RESAMPLE_NO_PAIR"))
            return false;
        const int dmin = down_ / up_;
        if (dmin > 1)
            return false;
        // computing lanes: 2 ql outputs are whole periods of the phase pattern
        int ql = 0;
        for (int q = kThreads; q >= 32; --q)
            if ((2 * q) % up_ == 0) {
                ql = q;
                break;
            }
        if (ql == 0)
            return false;
        const char *qle = PH_ENV_AB("PIPE_HIP_RESAMPLE_QL");  // A/B: the largest count of computing lanes
        const int qcap = qle ? std::atoi(qle) : 192;
        if (ql > qcap && (2 * qcap) % up_ == 0)
            ql = qcap;  // (three waves: four workgroups share a CU)
        else if (ql > qcap) {
            for (int q = qcap; q >= 32; --q)
                if ((2 * q) % up_ == 0) {
                    ql = q;
                    break;
                }
        }
        const int threads = (ql + 63) / 64 * 64;
        const size_t es_in = dtype_size(in_dtype);
        if (reinterpret_cast<uintptr_t>(a.in) % es_in != 0)
            return false;
        if (!ensure_pair_taps(ql, dmin))
            return false;
        PairArgs t{};
        t.r = a;
        t.ql = ql;
        t.ptaps = static_cast<const double *>(pair_taps_.p);
        t.adv = (int)((int64_t)2 * ql * down_ / up_);
        const char *ote = PH_ENV_AB("PIPE_HIP_RESAMPLE_OUT_TILE");  // A/B: outputs per tile
        const int out_tile = ote ? std::atoi(ote) : kOutTile;
        t.steps = out_tile / (2 * ql) > 0 ? out_tile / (2 * ql) : 1;
        for (;; --t.steps) {
            t.win = t.steps * t.adv + (T_ - 1) + kPairPad + dmin + 3;
            t.win += t.win & 1;
            if (t.win / 2 <= kPairVecs * threads || t.steps == 1)
                break;
        }
        if (t.win / 2 > kPairVecs * threads)
            return false;
        t.tile_out = 2 * ql * t.steps;
        t.plane = t.win;
        t.plane += (16 - t.plane % 32 + 32) % 32;  // plane stride == 16 (mod 32): the channels' planes on distinct banks
        // tiles start at a multiple of `up` outputs from Start (phase 0): `lead` virtual outputs sit
        // ahead of the call's first one and are not stored
        t.lead = (int)(a.out_total % up_);
        t.nb = (a.out_total - t.lead) / up_ * down_ - a.in_total;
        t.tiles_per_line = (int)((t.lead + a.out_frames + t.tile_out - 1) / t.tile_out);
        t.vec_ok = 1;
        const size_t lds = sizeof(double) * 4 * (size_t)t.plane;
        if (lds > 64 * 1024)
            return false;
        const int64_t ntiles = (int64_t)t.tiles_per_line * cfg.lines;
        const int64_t slots = 4 * 256;
        const int64_t per = (ntiles + slots - 1) / slots;
        const dim3 grid((unsigned)((ntiles + per - 1) / per));
        hipEvent_t ev_a = nullptr, ev_b = nullptr;
        if (timer.pair(&ev_a, &ev_b) != PIPE_HIP_OK)
            return false;
#ifdef PH_RS_PROF
        static DevBuf prof;
        if (!prof.p && prof.alloc(sizeof(unsigned long long) * 5 * 4 * 4096) != PIPE_HIP_OK)
            return false;
        t.prof = static_cast<unsigned long long *>(prof.p);
#endif
#define PH_RP(TI, TO, TTV, DM) hipExtLaunchKernelGGL((resample_pair_kernel<TI, TO, TTV, DM>), grid, dim3(threads), lds, s, ev_a, ev_b, 0, t)
#define PH_RP_T(TI, TO, DM)                 \
    switch (T_) {                           \
    case 8: PH_RP(TI, TO, 8, DM); break;    \
    case 12: PH_RP(TI, TO, 12, DM); break;  \
    case 16: PH_RP(TI, TO, 16, DM); break;  \
    case 24: PH_RP(TI, TO, 24, DM); break;  \
    default: PH_RP(TI, TO, 32, DM); break;  \
    }
#define PH_RP_D(TI, TO)      \
    if (dmin == 0) {         \
        PH_RP_T(TI, TO, 0)   \
    } else {                 \
        PH_RP_T(TI, TO, 1)   \
    }
        if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32) {
            PH_RP_D(float, float)
            last_kernel = "resample_pair_kernel<f32,f32>";
        } else if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F64) {
            PH_RP_D(double, double)
            last_kernel = "resample_pair_kernel<f64,f64>";
        } else if (in_dtype == PIPE_HIP_F32) {
            PH_RP_D(float, double)
            last_kernel = "resample_pair_kernel<f32,f64>";
        } else {
            PH_RP_D(double, float)
            last_kernel = "resample_pair_kernel<f64,f32>";
        }
#undef PH_RP_D
#undef PH_RP_T
#undef PH_RP
#ifdef PH_RS_PROF
        {
            static int launches = 0;
            if (++launches == 30) {
                (void)hipStreamSynchronize(s);
                const size_t nw = (size_t)grid.x * (threads / 64);
                std::vector<unsigned long long> h(nw * 5);
                (void)hipMemcpy(h.data(), t.prof, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
                static const char *names[5] = {"taps + first window", "request next window", "tap loops + stores", "deposit next window", "barrier"};
                double sum[5] = {}, tot = 0;
                for (size_t w = 0; w < nw; ++w)
                    for (int i = 0; i < 5; ++i)
                        sum[i] += (double)h[w * 5 + i];
                for (double v : sum)
                    tot += v;
                std::fprintf(stderr, "[resampler prof] s_memtime ticks per wave, %zu waves, %lld tiles\n", nw, (long long)ntiles);
                for (int i = 0; i < 5; ++i)
                    std::fprintf(stderr, "[resampler prof]   %-22s %9.1f  %5.1f %%\n", names[i], sum[i] / (double)nw, 100.0 * sum[i] / tot);
                std::fprintf(stderr, "[resampler prof]   %-22s %9.1f\n", "total", tot / (double)nw);
            }
        }
#endif
        return hipGetLastError() == hipSuccess;
    }

    int T_ = 1, up_ = 1, down_ = 1;
    std::vector<double> host_proto_;
    DevBuf pair_taps_;
    int pair_ql_ = 0;
    DevBuf wave_taps_;
    int wave_G_ = 0;
    DevBuf row_taps_;
    int cus_ = 256;
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
