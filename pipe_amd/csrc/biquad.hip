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
//
// Time-segmented form (float32 results only, never float64 buffers, never when the handle
// asks for bit-exactness): 4096 series are 64 waves -- 6 % of the chip.  The recurrence is
// linear in its state, so the call is cut into T segments per series and run as three launches
// that fill the machine:
//   1. every (series, segment) runs the recurrence from a ZERO state and keeps only its end
//      state z_k                                        (parallel over series x segments)
//   2. per series, the true incoming state of every segment follows from
//      s_{k+1} = z_k + M s_k, M = the zero-input state transition over one segment
//      (2S x 2S, computed on the host from the coefficients)       (T short serial steps)
//   3. every (series, segment) runs the SAME ordered fma recurrence as the exact kernel,
//      started from s_k, and stores its outputs          (parallel over series x segments)
// Pass 3 is the exact arithmetic given its start state; the start states carry the
// O(1e-16) relative rounding difference of step 2's reassociation, which a stable filter
// damps.  The float32 result therefore equals the oracle's except where the float64 value
// sits within ~1e-15 of a rounding boundary (then: the neighbouring float32, 1 ulp) -- the same
// contract as the FIR's overlap-save form ("the one tolerance").
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <cstring>
#include <memory>
#include <vector>
#include <atomic>

#include <hip/hip_ext.h>

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
    // time-segmented form
    double *seg;      // [T][nseries][S][2]: end states of pass 1, start states after pass 2
    int seglen, T;    // frames per segment (the last one may be shorter), segments per series
    int blocks_per_seg;
    int sstride, soff;  // tile kernel, one pass: doubles between two series' states, and this cascade's first one (a cascade run as two halves)
    int spb;          // few series (< a workgroup's lanes): segments per workgroup, lanes = (segment, series); 0 = one segment
    double *state_out;  // tile kernel, one pass: where a series' new state goes (a.state itself, or the other half of a double buffer)
};

// zero-input state transition over one segment, row-major (2S x 2S); a kernel argument, so the
// segmented form is offered up to kMaxSegSections sections
constexpr int kMaxSegSections = 4;
struct BiquadTransition {
    double m[2 * kMaxSegSections][2 * kMaxSegSections];
};

enum { kWhole = 0, kSegZeroState = 1, kSegFinal = 2, kSegSingle = 3 };

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

// NS = compile-time section count (1, 2) or 0 = runtime a.S; GAIN = a gain is folded in;
// MODE: the whole call per series (exact form), or one segment of it (passes 1 and 3 above)
template <typename TIn, typename TOut, int NS, bool GAIN, int MODE>
__global__ void __launch_bounds__(kThreads)
biquad_kernel(const TIn *__restrict__ in_base, TOut *__restrict__ out_base, const BiquadArgs a,
              const BiquadCoeffs q)
{
    // a workgroup never straddles two segments, so the frame count is wave-uniform
    int seg = MODE == kWhole ? 0 : (int)(blockIdx.x / (unsigned)a.blocks_per_seg);
    const int sblock = MODE == kWhole ? (int)blockIdx.x : (int)(blockIdx.x % (unsigned)a.blocks_per_seg);
    int sid = sblock * kThreads + threadIdx.x;
    bool live = sid < a.nseries;
    int64_t nframes = a.frames;
    if (MODE != kWhole) {
        if (a.spb > 0) {
            // Few series (one long stereo stream has two): a workgroup's lanes are (segment, series) pairs, spb
            // full-length segments per workgroup, so that every lane has a chain to walk (one segment per
            // workgroup left all but `nseries` of its lanes idle: 1 Line x 2 ch x 8.4 M frames ran at 22
            // Gsamples/s).  The series' LAST segment, which may be shorter, has the launch's last workgroup
            // to itself: the frame count stays uniform over a workgroup.
            const bool tail = blockIdx.x + 1 == gridDim.x;
            const int sl = (int)threadIdx.x / a.nseries;
            sid = (int)threadIdx.x - sl * a.nseries;
            seg = tail ? a.T - 1 : (int)blockIdx.x * a.spb + sl;
            live = tail ? (int)threadIdx.x < a.nseries : (sl < a.spb && seg < a.T - 1);
            nframes = tail ? a.frames - (int64_t)(a.T - 1) * a.seglen : (int64_t)a.seglen;
        } else {
            const int64_t f0s = (int64_t)seg * a.seglen;
            nframes = a.frames - f0s < a.seglen ? a.frames - f0s : (int64_t)a.seglen;
        }
    }
    const int sidc = live ? sid : 0;
    const int line = sidc / a.C;
    const int c = sidc - line * a.C;
    if (!live)
        seg = 0;
    double *__restrict__ st = MODE == kWhole ? a.state + (int64_t)sidc * a.S * 2
                                             : a.seg + ((int64_t)seg * a.nseries + sidc) * a.S * 2;
    const int64_t f_first = (int64_t)seg * a.seglen;
    // whole call through 32-bit offsets (the launcher guarantees the buffers are < 4 GiB)
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TIn *>(in_base), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out_base, 0, (int)a.out_bytes, 0x00020000);
    // dead lanes point past the end: their loads return 0 and their stores are dropped
    const int64_t e0 = ((int64_t)line * a.frames + f_first) * a.C + c;  // first element of this lane
    const unsigned vin = live ? (unsigned)(e0 * sizeof(TIn)) : 0xFFFFFFFFu;
    const unsigned vout = live ? (unsigned)(e0 * sizeof(TOut)) : 0xFFFFFFFFu;
    const unsigned sin_step = (unsigned)(a.C * sizeof(TIn));    // bytes per frame
    const unsigned sout_step = (unsigned)(a.C * sizeof(TOut));

    double s1[kMaxSections], s2[kMaxSections];
#pragma unroll
    for (int s = 0; s < kMaxSections; ++s) {
        s1[s] = 0.0;
        s2[s] = 0.0;
        if (MODE != kSegZeroState && live && s < a.S) {
            s1[s] = st[2 * s];
            s2[s] = st[2 * s + 1];
        }
    }
    // Runtime section count (3..8): the 40 coefficients live in VGPRs.  As kernel arguments they are
    // 80 SGPRs, more than a wave has: the compiler spilled them into VGPR lanes and fetched every
    // operand with v_readlane (a 3-section cascade took 7x the time of a 2-section one).
    double cv[NS == 0 ? kMaxSections : 1][5];
    if constexpr (NS == 0) {
#pragma unroll
        for (int s = 0; s < kMaxSections; ++s) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                cv[s][k] = q.c[s][k];
                asm volatile("" : "+v"(cv[s][k]));
            }
        }
    }
    auto step = [&](double x) -> double {
        double y;
        if constexpr (NS > 0) {
            y = biquad_step<NS>(x, s1, s2, q);
        } else {
            // nested, not a flat row of guards: the sections in use are 0..S-1, so a step takes one
            // taken branch (out of the nest) instead of one per unused section
            y = x;
            auto section = [&](auto self, auto idx) -> void {
                constexpr int sI = decltype(idx)::value;
                if constexpr (sI < kMaxSections) {
                    if (sI < a.S) {
                        const double v = __builtin_fma(cv[sI][0], y, s1[sI]);
                        const double t = __builtin_fma(cv[sI][1], y, s2[sI]);
                        s1[sI] = __builtin_fma(-cv[sI][3], v, t);
                        const double u = cv[sI][2] * y;
                        s2[sI] = __builtin_fma(-cv[sI][4], v, u);
                        y = v;
                        self(self, std::integral_constant<int, sI + 1>{});
                    }
                }
            };
            section(section, std::integral_constant<int, 0>{});
        }
        if constexpr (GAIN && MODE != kSegZeroState)
            y = y * a.gain;
        return y;
    };

    // two chunks in flight: chunk k+1 is loading while the dependent chain walks chunk k
    // (series are scarce -- 64 waves for 4096 series -- so no other wave hides the latency)
    const int64_t nchunks = nframes / kChunk;
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
        if constexpr (MODE == kSegZeroState) {
            // only the end state matters; keep the chain alive without a store
            asm volatile("" ::"v"(y[kChunk - 1]));
        } else {
            const unsigned so = (unsigned)f0 * sout_step;
#pragma unroll
            for (int u = 0; u < kChunk; ++u)
                buf_store(rout, vout, so + (unsigned)u * sout_step, y[u]);
        }
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
    for (int64_t f = nchunks * kChunk; f < nframes; ++f) {
        const double y = step((double)buf_load<TIn>(rin, vin, (unsigned)f * sin_step));
        if constexpr (MODE != kSegZeroState)
            buf_store(rout, vout, (unsigned)f * sout_step, (TOut)y);
    }

    if (MODE != kSegFinal && live) {
#pragma unroll
        for (int s = 0; s < kMaxSections; ++s) {
            if (s < a.S) {
                st[2 * s] = s1[s];
                st[2 * s + 1] = s2[s];
            }
        }
    }
}

// ---- time-segmented form through LDS tiles (up to 8 channels) ---------------------------------------
// The kernels above give every lane ONE series and let it walk its frames in place: with many channels per
// frame neighbouring lanes share cache lines (8 channels, 512 Lines: 400 Gsamples/s), with one or two each lane
// drags its own line along (74 / 166 Gsamples/s; one long stereo stream: 22).  Here a workgroup takes a TILE of
// (256 / C) SEG consecutive frames of all C channels of a Line (C <= 8): staged coalesced into LDS (16 bytes
// a lane, all of the tile's loads in flight at once), lane (c, g) walks segment g -- SEG frames -- of channel c out
// of LDS (segments SEG + 1 apart: the lanes on different banks), the 256 / C segments of a channel are chained by
// a scan inside the workgroup (affine maps with one constant matrix: y_g = A y_{g-1} + z_g, Hillis-Steele with
// A^(2^k) from a table).  The tiles of a series are chained
//   * in ONE pass (MODE kSegSingle, what ships): by look-back between the tiles of the launch, see BiquadLookArgs;
//   * or (PIPE_HIP_BIQUAD_TWO_PASS, the A/B form) by the same three passes as above: pass 1 (kSegZeroState) writes a
//     tile's zero-start end state, a scan kernel turns those into the tiles' start states, pass 2 (kSegFinal) folds
//     the tile's start state into segment 0, scans, walks every segment from its true start state.
// The tile is stored coalesced.  Same arithmetic contract as the lane-walk form (start states through powers of
// the transition matrix).  16.7 M samples a call, one pass: 528 Gsamples/s over 2048 stereo Lines of one buffer,
// 362 over ONE Line of 2048 buffers (scripts/biquad_shapes_probe.py; two passes: 350 / 308).
constexpr int kTileThreads = 256;
// frames per segment: SEG = 32, or 16 where Lines are short against a tile of 32s (4096-frame mono Lines would
// leave half of every 8192-frame tile empty)
constexpr int kTilePowers = 8;                 // A^(2^k), k = 0 .. 7 (256 segments of one channel)
constexpr int kTileMaxSections = 2;
struct BiquadTilePowers {
    double m[kTilePowers][2 * kTileMaxSections][2 * kTileMaxSections];
};

// ---- one pass over the input (MODE kSegSingle): the tiles of a series chained by look-back -----------
// A tile publishes its zero-start end state z_k as soon as its scan has it, then looks back: over the tiles
// before it, adding P z_j and multiplying P by M^tile, until one of them has published its TRUE end state (or the
// series' carried state is reached); that sum is the tile's start state, M^tile s + z_k its own true end state
// (published for the tiles after it).  Every 64-bit word of a published state carries the launch's epoch in its
// upper half (no flags, no fences, nothing to clear between launches).  Tiles take their index from their Line's
// counter in the order they start, so a tile only ever waits for tiles that already run: no residency condition.
// Segment g's start state is then its zero-start scan value + A^g s, A^g from a table.
struct BiquadLookArgs {
    unsigned long long *aggr, *incl;  // [T][nseries][2 S] doubles as two tagged words each
    unsigned *ticket, *ticket_next;   // [nl][8] tiles started per (Line, class): this launch's counters, the next launch's
    int classes;                      // counters per Line in use: 1, or 8 / nl (few Lines: one counter per XCD -- see the kernel)
    const double *tab;                // [256][2 S][2 S]: A^g
    int *err;                         // host-visible: a look-back that gave up
    unsigned epoch;
    int nl;
    double *state_bak;                // [nseries][2 S]: the carried state as tile 0 read it (what a failed launch is taken back to)
    unsigned long long spin_ticks;    // a wait for a predecessor's record gives up after this many s_memtime ticks
    int withhold;                     // debug (PIPE_HIP_PARAM_DEBUG): tiles of this index publish nothing (-1: none)
};
// (bounded by TIME: a preempted predecessor may be away for milliseconds; every 256th poll reads the clock)
__device__ __forceinline__ bool look_expired(unsigned &spins, unsigned long long &t0, unsigned long long limit)
{
    if ((++spins & 255u) != 0u)
        return false;
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    if (t0 == 0ull) {
        t0 = now;
        return false;
    }
    return now - t0 > limit;
}
__device__ __forceinline__ void look_publish(unsigned long long *p, const double *v, int n, unsigned epoch)
{
    for (int i = 0; i < n; ++i) {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, v[i]);
        __hip_atomic_store(p + 2 * i, ((unsigned long long)epoch << 32) | (b & 0xffffffffull), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 2 * i + 1, ((unsigned long long)epoch << 32) | (b >> 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int N>
__device__ __forceinline__ bool look_read(const unsigned long long *p, double *v, unsigned epoch)
{
    unsigned long long w[2 * N];
#pragma unroll
    for (int i = 0; i < 2 * N; ++i)
        w[i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 2 * N; ++i)
        ok = ok && (unsigned)(w[i] >> 32) == epoch;
#pragma unroll
    for (int i = 0; i < N; ++i)
        v[i] = __builtin_bit_cast(double, (w[2 * i] & 0xffffffffull) | (w[2 * i + 1] << 32));
    return ok;
}

template <typename T>
struct TileVec;
template <>
struct TileVec<float> {
    struct __attribute__((packed, aligned(4))) type {
        float v[4];
    };
};
template <>
struct TileVec<double> {
    struct __attribute__((aligned(8))) type {
        double v[2];
    };
};
// what a tile is staged as: float32 when the call reads and writes float32 (half the LDS: three workgroups on a CU)
template <typename TIn, typename TOut>
using TileStage = typename std::conditional<sizeof(TIn) == 4 && sizeof(TOut) == 4, float, double>::type;

template <typename TIn, typename TOut, int NS, bool GAIN, int MODE, int SEG>
__global__ void __launch_bounds__(kTileThreads)
biquad_tile_kernel(const TIn *__restrict__ in_base, TOut *__restrict__ out_base, const BiquadArgs a, const BiquadCoeffs q,
                   const BiquadTilePowers pw, int tiles_per_line, int C, int spc, int cmagic, const BiquadLookArgs lk,
                   const BiquadTransition mt)
{
    constexpr int N = 2 * NS;
    constexpr int kTileSeg = SEG, kTileElems = kTileThreads * SEG, kSegLog = SEG == 32 ? 5 : 4;
    using TS = TileStage<TIn, TOut>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *yb = reinterpret_cast<double *>(smem_raw);                 // [256][N]: the scan's exchange
    TS *xs = reinterpret_cast<TS *>(yb + kTileThreads * N);            // [256 segments][kTileSeg + 1]
    const int tid = (int)threadIdx.x;
    int line, tile;
    if constexpr (MODE == kSegSingle) {  // in starting order, tile-major: the tile before mine started before me
        if (tiles_per_line == 1) {  // (nobody to wait for: no need for an order)
            tile = 0;
            line = (int)blockIdx.x;
        } else {
            // a counter per Line (one counter for the launch: thousands of device-scope atomics on one address,
            // 17 of 47 us); this launch's counters count up from zero, the tile that draws 0 clears the Line's
            // counter of the NEXT launch (the two sets alternate).
            // FEW Lines (1, 2, 4: ONE stereo Line of 2048 buffers is 2048 draws from one address again, round 5:
            // 0.35 of HBM against 0.44 for 512 Lines): R = 8 / nl counters per Line.  Workgroup b is the k-th of its
            // Line (k = b / nl) and belongs to class j = k % R, the tiles t = j (mod R) of the Line; it draws its
            // class's next number q and takes tile q R + j.  Class (Line l, j) is exactly the workgroups with
            // b % 8 == j nl + l -- ONE XCD's (block b runs on XCD b % 8) -- and holds as many workgroups as tiles.
            // No residency condition, as before: the lowest unfinished tile of the launch has drawn its number (the
            // numbers of a class are drawn in starting order, the lower ones are finished, and an XCD whose
            // workgroups have finished starts its next one, which draws it) and all its predecessors are finished.
            line = (int)(blockIdx.x % (unsigned)lk.nl);
            const int R = lk.classes, cls = R > 1 ? (int)((blockIdx.x / (unsigned)lk.nl) % (unsigned)R) : 0;
            if (tid == 0) {
                const int t = (int)__hip_atomic_fetch_add(lk.ticket + line * 8 + cls, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t == 0)
                    lk.ticket_next[line * 8 + cls] = 0u;
                *reinterpret_cast<int *>(yb) = t * R + cls;
            }
            __syncthreads();
            tile = *reinterpret_cast<volatile int *>(yb);
            __syncthreads();
        }
    } else {
        line = (int)blockIdx.x / tiles_per_line;
        tile = (int)blockIdx.x - line * tiles_per_line;
    }
    // C channels (1 .. 8); spc = 256 / C segments per channel in a tile (lanes past C spc idle when C is not a power
    // of two); e / C for e < 9362 as (e cmagic) >> 16, cmagic = ceil(65536 / C)
    const int tfr = spc * kTileSeg;                                    // frames per tile
    const int clog = C == 1 ? 0 : C == 2 ? 1 : C == 4 ? 2 : C == 8 ? 3 : -1;
    const int nelt = tfr * C;                                          // elements per tile (<= kTileElems)
    const int64_t f0 = (int64_t)tile * tfr;
    const int nreal = (int)(a.frames - f0 < tfr ? a.frames - f0 : tfr);  // frames of this tile inside the Line
    const TIn *__restrict__ in = in_base + ((int64_t)line * a.frames + f0) * C;
    // element e of the tile = frame e / C, channel e % C -> segment (c, frame / 32), position frame % 32
    auto cell = [&](int e) {
        const int ft = (int)(((unsigned)e * (unsigned)cmagic) >> 16), cc = e - ft * C;
        return (cc * spc + (ft >> kSegLog)) * (kTileSeg + 1) + (ft & (kTileSeg - 1));
    };

    // ---- stage (coalesced, 16 bytes a lane; every load of the tile in flight before the first LDS write)
    const int nel = nreal * C;  // elements of the tile inside the Line
    {
        using V = typename TileVec<TIn>::type;
        constexpr int VW = 16 / (int)sizeof(TIn), NCH = kTileElems / (kTileThreads * VW);
        if (nel % VW == 0) {
            const V *__restrict__ vin = reinterpret_cast<const V *>(in);
            V v[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = tid + i * kTileThreads;
                if (ch * VW < nel)
                    v[i] = vin[ch];
                else
                    v[i] = V{};
            }
            if (clog >= 0) {
                // a power of two channels: chunk i of a lane sits i (256 VW / C / SEG) rows below chunk 0, element u of
                // a chunk in channel u % C, frame u / C of it (no carries: VW divides SEG) -- one add per cell
                const int b0 = cell(tid * VW);
                const int di = ((kTileThreads * VW) >> clog >> kSegLog) * (kTileSeg + 1);
#pragma unroll
                for (int i = 0; i < NCH; ++i)
#pragma unroll
                    for (int u = 0; u < VW; ++u)
                        xs[b0 + i * di + (u & (C - 1)) * spc * (kTileSeg + 1) + (u >> clog)] = (TS)v[i].v[u];
            } else {
#pragma unroll
                for (int i = 0; i < NCH; ++i)
#pragma unroll
                    for (int u = 0; u < VW; ++u) {
                        const int e = (tid + i * kTileThreads) * VW + u;
                        if (e < nelt)
                            xs[cell(e)] = (TS)v[i].v[u];
                    }
            }
        } else {
            for (int e = tid; e < nelt; e += kTileThreads)
                xs[cell(e)] = e < nel ? (TS)in[e] : (TS)0;
        }
    }
    __syncthreads();

    const bool active = tid < C * spc;
    const int c = active ? tid / spc : 0, g = active ? tid - c * spc : 0;  // my channel, my segment of it
    const int nfr = !active || nreal - g * kTileSeg < 0 ? 0 : (nreal - g * kTileSeg < kTileSeg ? nreal - g * kTileSeg : kTileSeg);
    const int64_t series = (int64_t)line * C + c;
    TS *__restrict__ col = xs + tid * (kTileSeg + 1);
    double s1[kMaxSections], s2[kMaxSections];
    auto walk = [&](bool write) {
        if (nfr == kTileSeg) {  // (a whole segment: no test per frame)
#pragma unroll
            for (int i = 0; i < kTileSeg; ++i) {
                double y = biquad_step<NS>((double)col[i], s1, s2, q);
                if (write) {
                    if constexpr (GAIN)
                        y = y * a.gain;
                    col[i] = (TS)y;  // (float32 staging only with float32 results: this is the result's rounding)
                }
            }
        } else {
#pragma unroll 4
            for (int i = 0; i < nfr; ++i) {
                double y = biquad_step<NS>((double)col[i], s1, s2, q);
                if (write) {
                    if constexpr (GAIN)
                        y = y * a.gain;
                    col[i] = (TS)y;
                }
            }
        }
    };
    // ---- zero-start end state of my segment
#pragma unroll
    for (int k = 0; k < NS; ++k)
        s1[k] = s2[k] = 0.0;
    walk(false);
    double y[N];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        y[2 * k] = s1[k];
        y[2 * k + 1] = s2[k];
    }
    double *__restrict__ tstate = a.seg + ((int64_t)tile * a.nseries + series) * N;
    if (MODE == kSegFinal && g == 0) {  // the tile's true start state rides into segment 0: y_0 = A t0 + z_0
        double t0[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
            t0[i] = tstate[i];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j)
                y[i] = __builtin_fma(pw.m[0][i][j], t0[j], y[i]);
    }
    // ---- inclusive scan over the channel's segments: y_g += A^(2^k) y_(g - 2^k)
#pragma unroll
    for (int k = 0; k < kTilePowers; ++k) {
        const int d = 1 << k;
        if (d >= spc)
            break;
#pragma unroll
        for (int i = 0; i < N; ++i)
            yb[tid * N + i] = y[i];
        __syncthreads();
        if (g >= d) {
            double o[N];
#pragma unroll
            for (int i = 0; i < N; ++i)
                o[i] = yb[(tid - d) * N + i];
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j)
                    y[i] = __builtin_fma(pw.m[k][i][j], o[j], y[i]);
        }
        __syncthreads();
    }
    // my segment's start state: the inclusive value of the segment before (segment 0: the tile's start state)
#pragma unroll
    for (int i = 0; i < N; ++i)
        yb[tid * N + i] = y[i];
    __syncthreads();
    double st0[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
        st0[i] = g > 0 ? yb[(tid - 1) * N + i] : (MODE == kSegFinal ? tstate[i] : 0.0);
    if constexpr (MODE == kSegSingle) {
        double *pf = reinterpret_cast<double *>(xs + kTileThreads * (kTileSeg + 1));  // [8 channels][N]: the tiles' start states
        if (active && g == 0) {
            const int64_t slot = ((int64_t)tile * a.nseries + series) * (2 * N);
            const bool full = nreal == tfr;
            double zk[N], acc[N], P[N][N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                zk[i] = yb[(tid + spc - 1) * N + i];
                acc[i] = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j)
                    P[i][j] = i == j ? 1.0 : 0.0;
            }
            const bool held = lk.withhold >= 0 && tile == lk.withhold;  // (debug: this tile's records never show up)
            if (full && tile + 1 < tiles_per_line && !held)
                look_publish(lk.aggr + slot, zk, N, lk.epoch);
            unsigned spins = 0;
            unsigned long long spin_t0 = 0;
            for (int k = tile - 1;;) {
                double v[N];
                bool last = false, got = false;
                if (k < 0) {  // (tile 0 only: see below)
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        v[i] = a.state[series * a.sstride + a.soff + i];
                        lk.state_bak[series * N + i] = v[i];  // (what a launch that fails is taken back to)
                    }
                    last = got = true;
                } else {
                    // tile 0 is asked for its TRUE end state only: it alone reads the series' carried state, and the
                    // Line's last tile overwrites that state once tile 0 has published (no copy after the launch)
                    // (both records requested together: one round trip, not one after the other -- with ONE Line the
                    // predecessor is the workgroup that started just ahead of this one, its aggregate is published about
                    // when it is asked for, and every extra round trip is on the path of all of the Line's tiles)
                    const int64_t sk = ((int64_t)k * a.nseries + series) * (2 * N);
                    double va[N];
                    const bool gi = look_read<N>(lk.incl + sk, v, lk.epoch);
                    const bool ga = look_read<N>(lk.aggr + sk, va, lk.epoch);
                    if (gi) {
                        last = got = true;
                    } else if (k > 0 && ga) {
#pragma unroll
                        for (int i = 0; i < N; ++i)
                            v[i] = va[i];
                        got = true;
                    }
                }
                if (got) {
#pragma unroll
                    for (int i = 0; i < N; ++i)
#pragma unroll
                        for (int j = 0; j < N; ++j)
                            acc[i] = __builtin_fma(P[i][j], v[j], acc[i]);
                    if (last)
                        break;
                    double Q[N][N];
#pragma unroll
                    for (int i = 0; i < N; ++i)
#pragma unroll
                        for (int j = 0; j < N; ++j) {
                            double q2 = 0.0;
#pragma unroll
                            for (int l = 0; l < N; ++l)
                                q2 = __builtin_fma(P[i][l], mt.m[l][j], q2);
                            Q[i][j] = q2;
                        }
                    double pmax = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i)
#pragma unroll
                        for (int j = 0; j < N; ++j) {
                            P[i][j] = Q[i][j];
                            pmax = __builtin_fmax(pmax, __builtin_fabs(Q[i][j]));
                        }
                    // what lies further back reaches this tile times less than 2^-70: below any rounding of the
                    // state, so a long stream's tiles do not queue up behind each other (with one Line hundreds
                    // of tiles run at once, none of them with its true end state yet)
                    if (pmax < 0x1p-70)
                        break;
                    --k;
                    spins = 0;
                } else {
                    __builtin_amdgcn_s_sleep(1);
                    if (look_expired(spins, spin_t0, lk.spin_ticks)) {  // seconds: something is wrong; give up loudly
                        *lk.err = 1;
                        break;
                    }
                }
            }
            if (full && tile + 1 < tiles_per_line && !held) {
                double e[N];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    e[i] = zk[i];
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        e[i] = __builtin_fma(mt.m[i][j], acc[j], e[i]);
                }
                look_publish(lk.incl + slot, e, N, lk.epoch);
            }
#pragma unroll
            for (int i = 0; i < N; ++i)
                pf[c * N + i] = acc[i];
        }
        __syncthreads();
        // segment g starts from its zero-start scan value + A^g (the tile's start state)
        const double *__restrict__ tg = lk.tab + (int64_t)g * (N * N);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j)
                st0[i] = __builtin_fma(tg[i * N + j], pf[c * N + j], st0[i]);
    }
    if constexpr (MODE == kSegZeroState) {
        // the tile's zero-start end state: after the tile's last real frame.  A full last segment: its inclusive
        // value; a partial one: walked again from its start state (its own map is not A)
        const int gl = (nreal - 1) >> kSegLog;  // owner of the last real frame (nreal >= 1: the grid covers real tiles only)
        if (active && g == gl) {
            if (nfr == kTileSeg) {
#pragma unroll
                for (int i = 0; i < N; ++i)
                    tstate[i] = y[i];
            } else {
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    s1[k] = st0[2 * k];
                    s2[k] = st0[2 * k + 1];
                }
                walk(false);
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    tstate[2 * k] = s1[k];
                    tstate[2 * k + 1] = s2[k];
                }
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            s1[k] = st0[2 * k];
            s2[k] = st0[2 * k + 1];
        }
        walk(true);
        if constexpr (MODE == kSegSingle) {
            // the series' state after the call: with the lane that walked the Line's last frame
            if (active && tile + 1 == tiles_per_line && g == ((nreal - 1) >> kSegLog)) {
                bool tile0_read = true;
                if (tiles_per_line > 1) {  // not before tile 0 has read the state this overwrites
                    double v[N];
                    unsigned spins = 0;
                    unsigned long long spin_t0 = 0;
                    while (!look_read<N>(lk.incl + series * (2 * N), v, lk.epoch)) {
                        __builtin_amdgcn_s_sleep(2);
                        if (look_expired(spins, spin_t0, lk.spin_ticks)) {
                            *lk.err = 2;
                            tile0_read = false;  // (the state stays what it was: the launch is taken back anyway)
                            break;
                        }
                    }
                }
                if (tile0_read) {
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
                        a.state_out[series * a.sstride + a.soff + 2 * k] = s1[k];
                        a.state_out[series * a.sstride + a.soff + 2 * k + 1] = s2[k];
                    }
                }
            }
        }
        __syncthreads();
        TOut *__restrict__ out = out_base + ((int64_t)line * a.frames + f0) * C;
        using V = typename TileVec<TOut>::type;
        constexpr int VW = 16 / (int)sizeof(TOut), NCH = kTileElems / (kTileThreads * VW);
        if (nel % VW == 0) {
            V *__restrict__ vout = reinterpret_cast<V *>(out);
            const int b0 = cell(tid * VW);
            const int di = clog >= 0 ? ((kTileThreads * VW) >> clog >> kSegLog) * (kTileSeg + 1) : 0;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = tid + i * kTileThreads;
                V v;
                if (clog >= 0) {
#pragma unroll
                    for (int u = 0; u < VW; ++u)
                        v.v[u] = (TOut)xs[b0 + i * di + (u & (C - 1)) * spc * (kTileSeg + 1) + (u >> clog)];
                } else {
#pragma unroll
                    for (int u = 0; u < VW; ++u)
                        v.v[u] = (TOut)xs[ch * VW + u < nelt ? cell(ch * VW + u) : 0];
                }
                if (ch * VW < nel)
                    vout[ch] = v;
            }
        } else {
            for (int e = tid; e < nel; e += kTileThreads)
                out[e] = (TOut)xs[cell(e)];
        }
    }
}

// LDS-staged form of the exact recurrence.  One workgroup = one Line (x one group of up to 64
// channels).  The one-lane-per-series kernel above pays a memory round trip per chunk of
// frames, in series with its dependent fma chain -- over PCIe (zero-copy ProcessFunc form) that
// is ~2 us per 32 frames.  Here all 256 lanes first copy a block of frames into LDS (coalesced,
// everything in flight at once, widened to float64), the channels' lanes then run the SAME
// ordered recurrence out of LDS (reads issued one chunk ahead, ~100 cycles instead of a memory
// round trip), results go back in place, and all lanes store them coalesced.  Bit-exact like the
// register form; the state lives in registers across blocks.
constexpr int kLdsThreads = 256;
constexpr int kLdsChunk = 32;

struct BiquadLdsArgs {
    double *state;
    int64_t frames;
    int C, S, cgroups;  // channels, sections, channel groups of <= 64 per Line
    int fb;             // frames per LDS block
    int plane;          // doubles between two channels' planes in LDS (even, plane / 2 odd)
    double gain;
};

typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename TIn, typename TOut, int NS, bool GAIN>
__global__ void __launch_bounds__(kLdsThreads)
biquad_lds_kernel(const TIn *__restrict__ in_base, TOut *__restrict__ out_base, const BiquadLdsArgs a,
                  const BiquadCoeffs q)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *xs = reinterpret_cast<double *>(smem_raw);  // [cg][plane]: one plane of frames per channel
    const int line = blockIdx.x / a.cgroups;
    const int c0 = (blockIdx.x - line * a.cgroups) * 64;
    const int cg = a.C - c0 < 64 ? a.C - c0 : 64;
    const int tid = threadIdx.x;
    const bool lane_live = tid < cg;
    double *__restrict__ st = a.state + ((int64_t)line * a.C + c0 + (lane_live ? tid : 0)) * a.S * 2;

    double s1[kMaxSections], s2[kMaxSections];
#pragma unroll
    for (int s = 0; s < kMaxSections; ++s) {
        s1[s] = 0.0;
        s2[s] = 0.0;
        if (lane_live && s < a.S) {
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

    const TIn *__restrict__ in = in_base + (int64_t)line * a.frames * a.C;
    TOut *__restrict__ out = out_base + (int64_t)line * a.frames * a.C;
    const bool whole = cg == a.C;  // the block is one contiguous run of elements
    for (int64_t f0 = 0; f0 < a.frames; f0 += a.fb) {
        const int nb = a.frames - f0 < a.fb ? (int)(a.frames - f0) : a.fb;
        const int nel = nb * cg;
        // ---- stage: 8 loads in flight per lane ------------------------------------
        for (int e0 = tid; e0 < nel; e0 += 8 * kLdsThreads) {
            TIn v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int e = e0 + u * kLdsThreads;
                e = e < nel ? e : nel - 1;
                const int64_t g = whole ? f0 * a.C + e : (f0 + e / cg) * a.C + c0 + e % cg;
                v[u] = in[g];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * kLdsThreads;
                if (e < nel)
                    xs[(e % cg) * a.plane + e / cg] = (double)v[u];
            }
        }
        __syncthreads();
        // ---- the recurrence, out of LDS, in place -----------------------------------
        // The lane of a series is alone on its SIMD: every instruction it issues costs it >= 4.4
        // cycles whatever it is (an LDS access of 16 bytes per lane ~20), so the loop holds nothing
        // but the five float64 operations of a step and HALF an LDS instruction each way -- the
        // series' frames are contiguous in its plane, two of them per 16-byte access, at immediate
        // offsets from one chunk pointer.  Measured (scripts/micro/dep_fma_clock.hip): the five
        // operations alone 21-24 cycles per step, with the two half accesses 40-43.
        if (lane_live) {
            double *__restrict__ col = xs + tid * a.plane;
            const int nch = nb / kLdsChunk;
            f64x2 xa[kLdsChunk / 2], xb[kLdsChunk / 2];
            if (nch > 0) {
#pragma unroll
                for (int u = 0; u < kLdsChunk / 2; ++u)
                    xa[u] = *reinterpret_cast<const f64x2 *>(col + 2 * u);
            }
            // chunk k is computed while chunk k+1 is being read (requested first; past the last
            // chunk the same chunk again, so that the request stays in this basic block); the two
            // register sets swap roles instead of being copied
            auto run_chunk = [&](f64x2 (&x)[kLdsChunk / 2], f64x2 (&nx)[kLdsChunk / 2], int k) {
                double *cur = col + k * kLdsChunk;
                const double *nxt = k + 1 < nch ? cur + kLdsChunk : cur;
#pragma unroll
                for (int u = 0; u < kLdsChunk / 2; ++u)
                    nx[u] = *reinterpret_cast<const f64x2 *>(nxt + 2 * u);
                __builtin_amdgcn_sched_barrier(0);  // the requests stay ahead of the chain
#pragma unroll
                for (int u = 0; u < kLdsChunk / 2; ++u) {
                    f64x2 y;
                    y.x = step(x[u].x);
                    y.y = step(x[u].y);
                    *reinterpret_cast<f64x2 *>(cur + 2 * u) = y;
                }
            };
            int k = 0;
            for (; k + 2 <= nch; k += 2) {
                run_chunk(xa, xb, k);
                run_chunk(xb, xa, k + 1);
            }
            if (k < nch)
                run_chunk(xa, xb, k);
            for (int n = nch * kLdsChunk; n < nb; ++n)
                col[n] = step(col[n]);
        }
        __syncthreads();
        // ---- store ------------------------------------------------------------------
        for (int e = tid; e < nel; e += kLdsThreads) {
            const int64_t g = whole ? f0 * a.C + e : (f0 + e / cg) * a.C + c0 + e % cg;
            out[g] = (TOut)xs[(e % cg) * a.plane + e / cg];
        }
        __syncthreads();
    }
    if (lane_live) {
#pragma unroll
        for (int s = 0; s < kMaxSections; ++s) {
            if (s < a.S) {
                st[2 * s] = s1[s];
                st[2 * s + 1] = s2[s];
            }
        }
    }
}

// Section-pipelined LDS form (2..8 sections, few Lines).  In the kernel above ONE lane runs all S
// sections of a frame, 5 S float64 operations per step on a wave that issues one instruction per
// >= 4.4 cycles.  Here a channel's sections sit in neighbouring lanes (8 lanes per channel) and every
// lane runs ONE section on the same plane in place: section s works two chunks behind section s - 1,
// reads what that lane wrote there, and overwrites it with its own output.  All communication is LDS
// reads and writes of one wave (they retire in order: no barrier), the loop is one section's five
// operations plus half an LDS access each way for any S, and the arithmetic of every section is the
// oracle's, in its order: bit-exact.  Chunks are kLdsChunk frames at a stride of kLdsChunk + 2 doubles, so
// that the lanes of a channel (two chunks = 544 bytes apart) read different banks.
constexpr int kSpLanes = 8;                  // lanes per channel: one per section
constexpr int kSpChannels = 64 / kSpLanes;   // channels per workgroup
constexpr int kSpStride = kLdsChunk + 2;     // doubles from chunk to chunk in a plane

__device__ __forceinline__ int sp_index(int f) { return f + 2 * (f / kLdsChunk); }

template <typename TIn, typename TOut, bool GAIN>
__global__ void __launch_bounds__(kLdsThreads)
biquad_lds_sp_kernel(const TIn *__restrict__ in_base, TOut *__restrict__ out_base, const BiquadLdsArgs a,
                     const BiquadCoeffs q)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *xs = reinterpret_cast<double *>(smem_raw);  // [cg][plane], chunks of kLdsChunk frames, kSpStride doubles apart
    const int line = blockIdx.x / a.cgroups;
    const int c0 = (blockIdx.x - line * a.cgroups) * kSpChannels;
    const int cg = a.C - c0 < kSpChannels ? a.C - c0 : kSpChannels;
    const int tid = threadIdx.x;
    const int ch = tid / kSpLanes, sec = tid % kSpLanes;
    const bool lane_live = tid < 64 && ch < cg && sec < a.S;
    double *st = a.state + (((int64_t)line * a.C + c0 + (lane_live ? ch : 0)) * a.S + (lane_live ? sec : 0)) * 2;

    double s1 = 0.0, s2 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, a1 = 0.0, a2 = 0.0;
    if (lane_live) {
        s1 = st[0];
        s2 = st[1];
    }
#pragma unroll
    for (int j = 0; j < kMaxSections; ++j) {  // the lane's own section (uniform reads, per-lane select)
        if (sec == j) {
            b0 = q.c[j][0];
            b1 = q.c[j][1];
            b2 = q.c[j][2];
            a1 = q.c[j][3];
            a2 = q.c[j][4];
        }
    }
    const double g = GAIN && sec == a.S - 1 ? a.gain : 1.0;  // y * 1.0 is y, bit for bit
    auto step = [&](double x) -> double {
        const double y = __builtin_fma(b0, x, s1);
        const double t = __builtin_fma(b1, x, s2);
        s1 = __builtin_fma(-a1, y, t);
        const double u = b2 * x;
        s2 = __builtin_fma(-a2, y, u);
        if constexpr (GAIN)
            return y * g;
        return y;
    };

    const TIn *__restrict__ in = in_base + (int64_t)line * a.frames * a.C;
    TOut *__restrict__ out = out_base + (int64_t)line * a.frames * a.C;
    const bool whole = cg == a.C;
    for (int64_t f0 = 0; f0 < a.frames; f0 += a.fb) {
        const int nb = a.frames - f0 < a.fb ? (int)(a.frames - f0) : a.fb;
        const int nel = nb * cg;
        for (int e0 = tid; e0 < nel; e0 += 8 * kLdsThreads) {
            TIn v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int e = e0 + u * kLdsThreads;
                e = e < nel ? e : nel - 1;
                const int64_t gi = whole ? f0 * a.C + e : (f0 + e / cg) * a.C + c0 + e % cg;
                v[u] = in[gi];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * kLdsThreads;
                if (e < nel)
                    xs[(e % cg) * a.plane + sp_index(e / cg)] = (double)v[u];
            }
        }
        __syncthreads();
        if (tid < 64) {  // one wave: its LDS operations retire in order
            double *col = xs + (lane_live ? ch : 0) * a.plane;
            const int nch = nb / kLdsChunk;
            const int trips = nch + 2 * (a.S - 1);
            f64x2 xa[kLdsChunk / 2], xb[kLdsChunk / 2];
#pragma unroll
            for (int u = 0; u < kLdsChunk / 2; ++u)
                xa[u] = xb[u] = f64x2{0.0, 0.0};
            if (lane_live && sec == 0 && nch > 0) {
#pragma unroll
                for (int u = 0; u < kLdsChunk / 2; ++u)
                    xa[u] = *reinterpret_cast<const f64x2 *>(col + 2 * u);
            }
            // trip k: the lane of section s works on chunk k - 2 s (if it has one) and asks for
            // chunk k - 2 s + 1, which section s - 1 finished in trip k - 1
            auto run_trip = [&](f64x2 (&x)[kLdsChunk / 2], f64x2 (&nx)[kLdsChunk / 2], int k) {
                const int kk = k - 2 * sec;
                if (lane_live && kk >= -1 && kk < nch - 1) {
                    const double *nxt = col + (kk + 1) * kSpStride;
#pragma unroll
                    for (int u = 0; u < kLdsChunk / 2; ++u)
                        nx[u] = *reinterpret_cast<const f64x2 *>(nxt + 2 * u);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (lane_live && kk >= 0 && kk < nch) {
                    double *cur = col + kk * kSpStride;
#pragma unroll
                    for (int u = 0; u < kLdsChunk / 2; ++u) {
                        f64x2 y;
                        y.x = step(x[u].x);
                        y.y = step(x[u].y);
                        *reinterpret_cast<f64x2 *>(cur + 2 * u) = y;
                    }
                }
                // What this lane stored is read by the NEXT section's lane in the next trip: a
                // cross-lane dependency the compiler cannot see (per lane the addresses differ).
                // The wavefront-scope fence makes the order part of the program (it emits no
                // instruction: one wave's LDS operations retire in order); the scheduling barrier
                // keeps the requests ahead of the chain.
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_sched_barrier(0);
            };
            int k = 0;
            for (; k + 2 <= trips; k += 2) {
                run_trip(xa, xb, k);
                run_trip(xb, xa, k + 1);
            }
            if (k < trips)
                run_trip(xa, xb, k);
            // the frames past the last whole chunk, section after section
            for (int j = 0; j < a.S; ++j) {
                if (lane_live && sec == j) {
                    for (int n = nch * kLdsChunk; n < nb; ++n)
                        col[sp_index(n)] = step(col[sp_index(n)]);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // section j + 1 reads what section j stored
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        for (int e = tid; e < nel; e += kLdsThreads) {
            const int64_t gi = whole ? f0 * a.C + e : (f0 + e / cg) * a.C + c0 + e % cg;
            out[gi] = (TOut)xs[(e % cg) * a.plane + sp_index(e / cg)];
        }
        __syncthreads();
    }
    if (lane_live) {
        st[0] = s1;
        st[1] = s2;
    }
}

// pass 2: per series, turn the zero-state end states z_k into the true start states s_k
// (in place) and leave the state after the last segment in the persistent state array.
// N = 2S states.  The z_k of kScanChunk segments are fetched together (independent loads), so
// the serial part is arithmetic only.
constexpr int kScanChunk = 16;
template <int N>
__global__ void __launch_bounds__(kThreads)
biquad_scan_kernel(const BiquadArgs a, const BiquadTransition mfull, const BiquadTransition mlast)
{
    const int sid = blockIdx.x * kThreads + threadIdx.x;
    if (sid >= a.nseries)
        return;
    double s[N];
    double *__restrict__ st = a.state + (int64_t)sid * N;
#pragma unroll
    for (int i = 0; i < N; ++i)
        s[i] = st[i];
    const int64_t stride = (int64_t)a.nseries * N;  // doubles between segments
    double *__restrict__ zbase = a.seg + (int64_t)sid * N;
    for (int k0 = 0; k0 < a.T; k0 += kScanChunk) {
        double z[kScanChunk][N];
#pragma unroll
        for (int u = 0; u < kScanChunk; ++u) {
            const int k = k0 + u < a.T ? k0 + u : a.T - 1;
#pragma unroll
            for (int i = 0; i < N; ++i)
                z[u][i] = zbase[k * stride + i];
        }
#pragma unroll
        for (int u = 0; u < kScanChunk; ++u) {
            const int k = k0 + u;
            if (k < a.T) {
                const BiquadTransition &m = k == a.T - 1 ? mlast : mfull;
                double nx[N];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    double acc = z[u][i];
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        acc = __builtin_fma(m.m[i][j], s[j], acc);
                    nx[i] = acc;
                }
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    zbase[k * stride + i] = s[i];
                    s[i] = nx[i];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
        st[i] = s[i];
}

// The same pass for MANY tiles of few series (one long stream: thousands of tiles, the lane above walks them one
// after the other at a memory round trip per chunk): one wave per series, lane l owns the R consecutive tiles
// l R .. l R + R - 1.  Every lane folds its own tiles from zero state, the 64 partial results are chained by a scan
// over the lanes (every lane's map is z + M^R s, so the scan needs only (M^R)^(2^j), j = 0 .. 5 -- a table from the
// host), and every lane walks its tiles again from its true start state.  (Lanes past the last tile, and the short
// range of the last used lane, never feed another lane.)
struct BiquadScanPowers {
    double m[6][2 * kTileMaxSections][2 * kTileMaxSections];
};
template <int N>
__global__ void __launch_bounds__(64)
biquad_scan_wave_kernel(const BiquadArgs a, const BiquadTransition mfull, const BiquadTransition mlast,
                        const BiquadScanPowers sp, int R)
{
    const int sid = (int)blockIdx.x, lane = (int)threadIdx.x;
    double *__restrict__ st = a.state + (int64_t)sid * N;
    const int64_t stride = (int64_t)a.nseries * N;
    double *__restrict__ zbase = a.seg + (int64_t)sid * N;
    const int k0 = lane * R < a.T ? lane * R : a.T, k1 = k0 + R < a.T ? k0 + R : a.T;
    constexpr int kU = 8;
    double y[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
        y[i] = 0.0;
    for (int kk = k0; kk < k1; kk += kU) {
        double z[kU][N];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int k = kk + u < k1 ? kk + u : k1 - 1;
#pragma unroll
            for (int i = 0; i < N; ++i)
                z[u][i] = zbase[k * stride + i];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (kk + u < k1) {
                double nx[N];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    double acc = z[u][i];
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        acc = __builtin_fma(mfull.m[i][j], y[j], acc);
                    nx[i] = acc;
                }
#pragma unroll
                for (int i = 0; i < N; ++i)
                    y[i] = nx[i];
            }
    }
    double s0[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
        s0[i] = st[i];
    if (lane == 0) {  // the series' carried state rides in with lane 0: Y_0 = y_0 + M^R s0
        double nx[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double acc = y[i];
#pragma unroll
            for (int j = 0; j < N; ++j)
                acc = __builtin_fma(sp.m[0][i][j], s0[j], acc);
            nx[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < N; ++i)
            y[i] = nx[i];
    }
#pragma unroll
    for (int jp = 0; jp < 6; ++jp) {
        const int d = 1 << jp;
        double o[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
            o[i] = __shfl_up(y[i], d, 64);
        if (lane >= d) {
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j)
                    y[i] = __builtin_fma(sp.m[jp][i][j], o[j], y[i]);
        }
    }
    double s[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double prev = __shfl_up(y[i], 1, 64);
        s[i] = lane == 0 ? s0[i] : prev;
    }
    for (int kk = k0; kk < k1; kk += kU) {
        double z[kU][N];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int k = kk + u < k1 ? kk + u : k1 - 1;
#pragma unroll
            for (int i = 0; i < N; ++i)
                z[u][i] = zbase[k * stride + i];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int k = kk + u;
            if (k < k1) {
                const BiquadTransition &m = k == a.T - 1 ? mlast : mfull;
                double nx[N];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    double acc = z[u][i];
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        acc = __builtin_fma(m.m[i][j], s[j], acc);
                    nx[i] = acc;
                }
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    zbase[k * stride + i] = s[i];
                    s[i] = nx[i];
                }
            }
        }
    }
    if (k1 == a.T && k0 < k1) {
#pragma unroll
        for (int i = 0; i < N; ++i)
            st[i] = s[i];
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
    int start_lines(int first, int count, hipStream_t s) override
    {
        const size_t per = sizeof(double) * (size_t)cfg.channels * (size_t)S_ * 2u;
        if (count > 0)
            PH_HIP(hipMemsetAsync(static_cast<char *>(state_.p) + per * (size_t)first, 0, per * (size_t)count, s));
        return PIPE_HIP_OK;
    }
    int set_param(int32_t param, const double *values, int32_t count) override
    {
        if (param == PIPE_HIP_PARAM_EXACT && count == 1 && values) {
            exact_ = values[0] != 0.0;
            return PIPE_HIP_OK;
        }
        if (param == PIPE_HIP_PARAM_RELAXED_F64 && count == 1 && values) {  // float64 results may take the relaxed forms too
            relaxed_f64_ = values[0] != 0.0;
            half_[0].reset();
            half_[1].reset();
            return PIPE_HIP_OK;
        }
        if (param == PIPE_HIP_PARAM_DEBUG && count == 2 && values) {  // the next tile launch fails on demand
            debug_withhold_ = (int)values[0];
            debug_limit_us_ = values[1];
            for (auto &h : half_)
                if (h) {
                    h->debug_withhold_ = debug_withhold_;
                    h->debug_limit_us_ = debug_limit_us_;
                }
            return PIPE_HIP_OK;
        }
        if (param != PIPE_HIP_PARAM_COEFFS || count != 5 * S_ || !values)
            return PIPE_HIP_EINVAL;
        std::memcpy(q_.c, values, sizeof(double) * 5u * (size_t)S_);  // kernel argument
        mfull_len_ = mlast_len_ = sp_len_ = tab_seg_ = -1;
        kappa_ = -1.0;
        half_[0].reset();
        half_[1].reset();
        pw_seg_ = -1;
        return PIPE_HIP_OK;
    }
    // PIPE_HIP_PARAM_RESIDENT: a queued launch can be taken back when it is the one-pass tile form over all Lines
    // (states out of place); the ordered forms update their states in place and cannot.  Whether a call takes that
    // form depends on its frame count: armable_for() is asked per call.
    bool armable() const override { return S_ <= kTileMaxSections && cfg.channels <= 8; }
    bool armable_for(int64_t frames, int out_dtype) override
    {
        if (exact_ || env_exact_ || (out_dtype != PIPE_HIP_F32 && !relaxed_f64_) || !relaxed_ok() || windowed() || ext_state_)
            return false;
        const int64_t nseries = (int64_t)cfg.lines * cfg.channels;
        const bool long_few = !seg_min_from_env_ && frames >= kTileLatencyFrames;
        return S_ <= kTileMaxSections && cfg.channels <= 8 && (frames * nseries >= seg_min_samples_ || long_few) &&
               frames >= tile_min_frames() && !PH_ENV_AB("PIPE_HIP_BIQUAD_NO_TILE") && !PH_ENV_AB("PIPE_HIP_BIQUAD_TWO_PASS") &&
               !(cfg.channels >= kTileWalkChannels && cfg.lines >= tile_walk_lines_);
    }
    void rollback_launch() override
    {
        if (last_oop_) {
            std::swap(state_.p, state_alt_.p);
            last_oop_ = false;
            last_tile_.valid = false;
        }
    }
    // The one-pass tile launch of a synchronous entry: wait for it; if a look-back gave up, the carried state goes
    // back to what tile 0 of every series read (the kernel keeps that copy) and the call runs again through the
    // ordered recurrence, which waits for nobody.
    int take_back(hipStream_t s)
    {
        const TileCall &c = last_tile_;
        if (c.oop) {  // (the old states sit untouched in the other half)
            std::swap(state_.p, state_alt_.p);
            return PIPE_HIP_OK;
        }
        const size_t w = sizeof(double) * 2u * (size_t)S_;
        PH_HIP(hipMemcpy2DAsync(c.state + c.soff, sizeof(double) * (size_t)c.sstride, state_bak_.p, w, w, (size_t)c.nseries,
                                hipMemcpyDeviceToDevice, s));
        return PIPE_HIP_OK;
    }
    // the look-back flag of this handle's last tile launch (read and cleared); the launch has been waited for
    bool read_flag()
    {
        if (!err_.p || err_checked_)
            return false;
        volatile int *e = static_cast<volatile int *>(err_.p);
        err_checked_ = true;
        if (*e == 0)
            return false;
        *e = 0;
        return true;
    }
    // (the queued-ahead path: launch k's flag while launch k + 1 is queued -- the flag only, err_checked_ describes k + 1)
    bool take_failure_flag() override
    {
        bool f = false;
        for (auto &h : half_)
            f = (h && h->take_failure_flag()) || f;
        if (err_.p) {
            volatile int *e = static_cast<volatile int *>(err_.p);
            if (*e != 0) {
                *e = 0;
                f = true;
            }
        }
        return f;
    }
    int settle(hipStream_t s, bool *reran) override
    {
        if (reran)
            *reran = false;
        bool failed = false, halves = false;
        for (int h = 0; h < 2; ++h)
            if (half_[h] && half_[h]->last_tile_.valid) {  // (3 - 4 sections: the halves' launches)
                halves = true;
                failed = half_[h]->read_flag() || failed;
            }
        if (halves) {
            for (int h = 0; h < 2; ++h)
                if (half_[h] && half_[h]->last_tile_.valid) {
                    if (failed)
                        PH_TRY(half_[h]->take_back(s));
                    half_[h]->last_tile_.valid = false;
                }
            if (!failed)
                return PIPE_HIP_OK;
            PH_TRY(rerun_ordered(split_call_, s));
            PH_HIP(hipStreamSynchronize(s));
            if (reran)
                *reran = true;
            return PIPE_HIP_OK;
        }
        if (!last_tile_.valid)
            return poll_error();
        const TileCall c = last_tile_;
        if (!read_flag()) {
            last_tile_.valid = false;
            return PIPE_HIP_OK;
        }
        PH_TRY(take_back(s));
        last_tile_.valid = false;
        PH_TRY(rerun_ordered(c, s));
        PH_HIP(hipStreamSynchronize(s));
        if (reran)
            *reran = true;
        return PIPE_HIP_OK;
    }
    // Precondition as for the fused chain's: the stream of the last launch has been synchronised.
    int poll_error() override
    {
        for (auto &h : half_)
            if (h && h->poll_error() != PIPE_HIP_OK)
                return PIPE_HIP_EHIP;
        if (read_flag()) {
            // (an asynchronous call whose buffers are no longer ours: it cannot be run again from here, but the
            // carried state can be what it was before it -- the caller may submit the batch again)
            if (last_tile_.valid) {
                (void)take_back(stream);
                (void)hipStreamSynchronize(stream);
                last_tile_.valid = false;
            }
            return PIPE_HIP_EHIP;
        }
        return PIPE_HIP_OK;
    }
    struct TileCall {
        const void *d_in;
        void *d_out;
        int in_dtype, out_dtype;
        int64_t frames;
        double *state;
        int sstride, soff, nseries;
        bool valid;
        bool has_gain = false;  // a chain's gain folded into this launch's store (set around run() only)
        double gain = 1.0;
        bool oop = false;       // the launch wrote the other half of the state double buffer (and the halves traded places)
    };
    TileCall last_tile_{nullptr, nullptr, 0, 0, 0, nullptr, 0, 0, 0, false}, split_call_{nullptr, nullptr, 0, 0, 0, nullptr, 0, 0, 0, false};
    int rerun_ordered(const TileCall &c, hipStream_t s)
    {
        static bool said = false;
        if (!said) {
            said = true;
            std::fprintf(stderr, "pipe_hip: a tile biquad launch gave up waiting for a predecessor tile; the call was run again "
                                 "through the ordered recurrence (further occurrences are not reported)\n");
        }
        const bool hg = has_gain_;
        const double gv = gain_;
        has_gain_ = c.has_gain;
        gain_ = c.gain;
        ordered_once_ = true;
        const int rc = run(c.d_in, c.in_dtype, c.d_out, c.out_dtype, c.frames, s);
        ordered_once_ = false;
        has_gain_ = hg;
        gain_ = gv;
        return rc;
    }
    bool ordered_once_ = false;
    bool relaxed_f64_ = false;  // PIPE_HIP_PARAM_RELAXED_F64: float64 buffers may take the time-segmented forms
    int debug_withhold_ = -1;
    double debug_limit_us_ = 0.0;
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
        // (a window of Lines: the state slice of exactly those Lines)
        const int nl = active_lines();
        a.state = static_cast<double *>(state_.p) + (size_t)win_first * (size_t)cfg.channels * (size_t)S_ * 2u;
        a.frames = frames;
        a.C = cfg.channels;
        a.S = S_;
        a.nseries = nl * cfg.channels;
        a.sstride = 2 * S_;
        a.soff = 0;
        if (ext_state_) {  // (half of a longer cascade: the states live in the whole cascade's array)
            a.state = ext_state_ + (size_t)win_first * (size_t)cfg.channels * (size_t)ext_stride_;
            a.sstride = ext_stride_;
            a.soff = ext_off_;
        }
        a.gain = gain_;
        a.in_bytes = (int64_t)dtype_size(in_dtype) * frames * cfg.channels * nl;
        a.out_bytes = (int64_t)dtype_size(out_dtype) * frames * cfg.channels * nl;
        if (a.in_bytes >= ((int64_t)1 << 32) - 4096 || a.out_bytes >= ((int64_t)1 << 32) - 4096)
            return PIPE_HIP_EINVAL;  // 32-bit buffer offsets: split the call (never reached by buffer_size*max_batch in practice)
        const unsigned sblocks = (unsigned)((a.nseries + kThreads - 1) / kThreads);
        // time-segmented form: float32 results (or float64 intermediates of a float32 chain)
        // only, and only when the series alone cannot fill the machine
        // (the relaxed forms carry states through powers of the transition matrix: stable sections only)
        const bool relaxed = !exact_ && !env_exact_ && !ordered_once_ && (out_dtype == PIPE_HIP_F32 || relaxed_f64_out || relaxed_f64_) && relaxed_ok();
        last_tile_.valid = false;
        last_oop_ = false;
        // (small calls are launch-bound either way and stay bit-exact, like the FIR's)
        // (... or LONG: see long_few below -- more than 8 channels cannot take the tile form, and one 4096 x 16 float32
        // buffer a call took the ordered form's 141 us where the lane walk over segments takes 14)
        bool segmented = relaxed && S_ <= kMaxSegSections && frames >= 4 * kChunk && a.nseries < 65536 &&
                         (frames * a.nseries >= seg_min_samples_ ||
                          (cfg.channels > 8 && !seg_min_from_env_ && frames >= kTileLatencyFrames));
        if (segmented) {
            // few series: lanes are (segment, series) pairs, more and shorter segments (the serial scan over them
            // bounds their number)
            a.spb = a.nseries < kThreads && !PH_ENV_AB("PIPE_HIP_BIQUAD_SEG_ONE_PER_BLOCK") ? kThreads / a.nseries : 0;
            const int tmax = a.spb > 0 ? 4096 : 1024;
            int T = (int)(frames / 64);
            T = T < 2 ? 2 : (T > tmax ? tmax : T);
            int64_t seglen = (frames + T - 1) / T;
            seglen = (seglen + kChunk - 1) / kChunk * kChunk;
            T = (int)((frames + seglen - 1) / seglen);
            segmented = T >= 2;
            a.seglen = (int)seglen;
            a.T = T;
            a.blocks_per_seg = (int)sblocks;
        }
        // up to 8 channels and one or two sections: the LDS-tiled form (coalesced; the lane walks are not)
        const int tc = cfg.channels, tspc = tc <= 8 ? kTileThreads / tc : 0;
        // A call that is too small for the thresholds above but LONG -- one pipe buffer of a Line or two per
        // ProcessFunc call -- is where the ordered recurrence hurts most: it is one wave's issue, 22 ns a frame
        // whatever the chip (4096 x 2: 90 us, a host core does it in 10).  The tile form takes a float32 buffer of
        // kTileLatencyFrames or more frames in one short launch (4096 x 2: 12.8 us); PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES
        // set in the environment is the only rule when it is there.  (Until round 6 only for up to 64 series: 65 - 255
        // series of 4096 frames -- 16 Lines x 8 ch, one buffer each -- fell between this rule and the sample count above
        // and took the ordered form, 86 - 118 us where the tile form takes 9 - 12, profiles/r06_biquad_dispatch_gap.txt.)
        const bool long_few = !seg_min_from_env_ && frames >= kTileLatencyFrames;
        a.state_out = a.state;
        const bool tiled = relaxed && S_ <= kTileMaxSections && tc <= 8 && (frames * a.nseries >= seg_min_samples_ || long_few) &&
                           frames >= tile_min_frames() && !PH_ENV_AB("PIPE_HIP_BIQUAD_NO_TILE") &&
                           !(cfg.channels >= kTileWalkChannels && nl >= tile_walk_lines_ && segmented);
        // 3 or 4 sections: the tile kernel holds two, so two tile passes over the halves of the cascade with a float64
        // stream between them (24 bytes a sample instead of 8) -- where the lane walk crawls (few Lines or channels)
        // (a LONG buffer of few series here too, since round 6: one 4096 x 2 float32 buffer through four sections took the
        // ordered form's 100 us where the two tile passes take 24, profiles/r06_biquad_dispatch_gap.txt)
        if (relaxed && S_ > kTileMaxSections && S_ <= 2 * kTileMaxSections && tc <= 8 && (frames * a.nseries >= seg_min_samples_ || long_few) &&
            frames >= tile_min_frames() && split_wanted(nl))
            return run_split(d_in, in_dtype, d_out, out_dtype, frames, a, s);
        if (ext_state_ && !tiled)
            return PIPE_HIP_EINVAL;  // (only the one-pass tile kernel knows the stride: run_split checked)
        PH_TRY(timer.begin(s));
        if (tiled) {
            // segments of 32 frames, or of 16 where that fills the tiles better by a quarter of the call
            const int64_t t32 = tspc * 32, t16 = tspc * 16;
            const int64_t waste32 = (frames + t32 - 1) / t32 * t32 - frames, waste16 = (frames + t16 - 1) / t16 * t16 - frames;
            const char *seg_env = PH_ENV_AB("PIPE_HIP_BIQUAD_TILE_SEG");  // A/B: 16 or 32 whatever the shape
            const int seg = seg_env ? (std::atoi(seg_env) == 16 ? 16 : 32)
                                    : (waste32 - waste16 > frames / 4 && !PH_ENV_AB("PIPE_HIP_BIQUAD_TILE_SEG32") ? 16 : 32);
            const int tfr = (int)(seg == 32 ? t32 : t16);
            a.T = (int)((frames + tfr - 1) / tfr);
            a.seglen = tfr;
            if (mfull_len_ != tfr) {
                transition(tfr, &mfull_);
                mfull_len_ = tfr;
            }
            if (pw_seg_ != seg) {
                for (int k = 0; k < kTilePowers; ++k) {
                    BiquadTransition m;
                    transition(seg << k, &m);
                    for (int i = 0; i < 2 * kTileMaxSections; ++i)
                        for (int j = 0; j < 2 * kTileMaxSections; ++j)
                            pw_.m[k][i][j] = m.m[i][j];
                }
                pw_seg_ = seg;
            }
            const int last_len = (int)(frames - (int64_t)(a.T - 1) * tfr);
            if (mlast_len_ != last_len) {
                if (last_len == tfr)
                    mlast_ = mfull_;
                else
                    transition(last_len, &mlast_);
                mlast_len_ = last_len;
            }
            const dim3 tgrid((unsigned)a.T * (unsigned)nl);
            const size_t lds_rest = sizeof(double) * ((size_t)kTileThreads * 2u * (size_t)S_ + 8u * 2u * (size_t)S_);
            const int cmagic = (65536 + tc - 1) / tc;
            const bool single = !PH_ENV_AB("PIPE_HIP_BIQUAD_TWO_PASS");
            BiquadLookArgs lk{};
            if (ext_state_ && !single)
                return PIPE_HIP_EINVAL;
            // One pass over ALL of the handle's Lines: the new states go to the other half of a double buffer and the
            // halves trade places -- a launch that is taken back (a look-back that gave up, queued work of the resident
            // path that is dropped) then left the old states untouched.  (A window of Lines, or a half of a longer
            // cascade working on its parent's array, writes in place and keeps the copy tile 0 makes.)
            const bool oop = single && !windowed() && !ext_state_;
            if (oop) {
                if (!state_alt_.p)
                    PH_TRY(state_alt_.alloc(state_bytes_));
                a.state_out = static_cast<double *>(state_alt_.p);
            }
            if (single) {
                PH_TRY(prepare_look(&lk, a, seg, nl, tgrid.x, s));
            } else {
                const size_t need = sizeof(double) * (size_t)a.T * (size_t)a.nseries * (size_t)S_ * 2u;
                if (seg_.bytes < need)
                    PH_TRY(seg_.alloc(need));
                a.seg = static_cast<double *>(seg_.p);
            }
#define PH_BT4(TI, TO, NSV, G, SEGV)                                                                                       \
    do {                                                                                                              \
        const size_t lds = lds_rest + sizeof(TileStage<TI, TO>) * (size_t)kTileThreads * (SEGV + 1);                 \
        if (single) {                                                                                                 \
            auto k0 = biquad_tile_kernel<TI, TO, NSV, G, kSegSingle, SEGV>;                                           \
            PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)lds));                                                                    \
            hipLaunchKernelGGL(k0, tgrid, dim3(kTileThreads), lds, s, static_cast<const TI *>(d_in),                  \
                               static_cast<TO *>(d_out), a, q_, pw_, a.T, tc, tspc, cmagic, lk, mfull_);              \
            break;                                                                                                    \
        }                                                                                                             \
        auto k1 = biquad_tile_kernel<TI, TO, NSV, G, kSegZeroState, SEGV>;                                            \
        auto k2 = biquad_tile_kernel<TI, TO, NSV, G, kSegFinal, SEGV>;                                                \
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k1), hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                   (int)lds));                                                                        \
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                   (int)lds));                                                                        \
        hipLaunchKernelGGL(k1, tgrid, dim3(kTileThreads), lds, s, static_cast<const TI *>(d_in),                      \
                           static_cast<TO *>(d_out), a, q_, pw_, a.T, tc, tspc, cmagic, lk, mfull_);                  \
        if (a.T > kWaveScanMinTiles && !PH_ENV_AB("PIPE_HIP_BIQUAD_NO_WAVE_SCAN"))                                \
            launch_scan_wave(s, a);                                                                                   \
        else                                                                                                          \
            launch_scan(sblocks, s, a, mlast_);                                                                       \
        hipLaunchKernelGGL(k2, tgrid, dim3(kTileThreads), lds, s, static_cast<const TI *>(d_in),                      \
                           static_cast<TO *>(d_out), a, q_, pw_, a.T, tc, tspc, cmagic, lk, mfull_);                  \
    } while (0)
#define PH_BT3(TI, TO, NSV, G)                 \
    do {                                       \
        if (seg == 32)                         \
            PH_BT4(TI, TO, NSV, G, 32);        \
        else                                   \
            PH_BT4(TI, TO, NSV, G, 16);        \
    } while (0)
#define PH_BT(TI, TO, NAME)                    \
    do {                                       \
        if (has_gain_) {                       \
            if (S_ == 1)                       \
                PH_BT3(TI, TO, 1, true);       \
            else                               \
                PH_BT3(TI, TO, 2, true);       \
        } else {                               \
            if (S_ == 1)                       \
                PH_BT3(TI, TO, 1, false);      \
            else                               \
                PH_BT3(TI, TO, 2, false);      \
        }                                      \
        last_kernel = NAME;                    \
    } while (0)
            if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32)
                PH_BT(float, float, "biquad_tile_kernel<f32,f32,segmented>");
            else if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F64)
                PH_BT(double, double, "biquad_tile_kernel<f64,f64,segmented>");
            else if (in_dtype == PIPE_HIP_F32)
                PH_BT(float, double, "biquad_tile_kernel<f32,f64,segmented>");
            else
                PH_BT(double, float, "biquad_tile_kernel<f64,f32,segmented>");
#undef PH_BT
#undef PH_BT3
#undef PH_BT4
            if (single)  // (what settle() takes back and runs again if the look-back gives up)
                last_tile_ = TileCall{d_in, d_out, in_dtype, out_dtype, frames, a.state, a.sstride, a.soff, (int)a.nseries, true, has_gain_, gain_, oop};
            if (oop)
                std::swap(state_.p, state_alt_.p);
            last_oop_ = oop;
        } else if (segmented) {
            const size_t need = sizeof(double) * (size_t)a.T * (size_t)a.nseries * (size_t)S_ * 2u;
            if (seg_.bytes < need)
                PH_TRY(seg_.alloc(need));
            a.seg = static_cast<double *>(seg_.p);
            if (mfull_len_ != a.seglen) {
                transition(a.seglen, &mfull_);
                mfull_len_ = a.seglen;
            }
            BiquadTransition mlast;
            const int64_t last_len = frames - (int64_t)(a.T - 1) * a.seglen;
            if (last_len == a.seglen)
                mlast = mfull_;
            else
                transition((int)last_len, &mlast);
            const dim3 sgrid(a.spb > 0 ? (unsigned)((a.T - 1 + a.spb - 1) / a.spb + 1) : sblocks * (unsigned)a.T);
#define PH_BQ3(TI, TO, NSV, G)                                                                         \
    do {                                                                                               \
        hipLaunchKernelGGL((biquad_kernel<TI, TO, NSV, G, kSegZeroState>), sgrid, dim3(kThreads), 0, s, \
                           static_cast<const TI *>(d_in), static_cast<TO *>(d_out), a, q_);            \
        launch_scan(sblocks, s, a, mlast);                                                             \
        hipLaunchKernelGGL((biquad_kernel<TI, TO, NSV, G, kSegFinal>), sgrid, dim3(kThreads), 0, s,     \
                           static_cast<const TI *>(d_in), static_cast<TO *>(d_out), a, q_);            \
    } while (0)
#define PH_BQ2(TI, TO, G)              \
    do {                               \
        if (S_ == 1)                   \
            PH_BQ3(TI, TO, 1, G);      \
        else if (S_ == 2)              \
            PH_BQ3(TI, TO, 2, G);      \
        else                           \
            PH_BQ3(TI, TO, 0, G);      \
    } while (0)
#define PH_BQ(TI, TO, NAME)            \
    do {                               \
        if (has_gain_)                 \
            PH_BQ2(TI, TO, true);      \
        else                           \
            PH_BQ2(TI, TO, false);     \
        last_kernel = NAME;            \
    } while (0)
            if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32)
                PH_BQ(float, float, "biquad_kernel<f32,f32,segmented>");
            else if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F64)
                PH_BQ(double, double, "biquad_kernel<f64,f64,segmented>");
            else if (in_dtype == PIPE_HIP_F32)
                PH_BQ(float, double, "biquad_kernel<f32,f64,segmented>");
            else
                PH_BQ(double, float, "biquad_kernel<f64,f32,segmented>");
#undef PH_BQ
#undef PH_BQ2
#undef PH_BQ3
        } else if (use_sp_form()) {
            // several sections: one lane per section, two chunks apart on the same LDS plane
            BiquadLdsArgs la{};
            la.state = a.state;
            la.frames = frames;
            la.C = cfg.channels;
            la.S = S_;
            la.cgroups = (cfg.channels + kSpChannels - 1) / kSpChannels;
            la.gain = gain_;
            const int cg = cfg.channels < kSpChannels ? cfg.channels : kSpChannels;
            int64_t chunks = (60 * 1024) / (int64_t)(sizeof(double) * kSpStride * cg);
            const int64_t need = (frames + kLdsChunk - 1) / kLdsChunk;
            if (chunks > need)
                chunks = need;
            la.fb = (int)(chunks * kLdsChunk);
            // planes an odd number of 16-byte units apart (see the S = 1 form)
            la.plane = (int)(chunks * kSpStride) + (chunks % 2 == 0 ? 2 : 0);
            const size_t lds = sizeof(double) * (size_t)la.plane * (size_t)cg;
            const dim3 grid((unsigned)(nl * la.cgroups));
            // a ProcessFunc-form buffer: this launch is the call's last operation and signals its completion
            hipEvent_t done = completion;
            completion = nullptr;
#define PH_BQ(TI, TO, NAME)                                                                             \
    do {                                                                                               \
        if (has_gain_)                                                                                 \
            hipExtLaunchKernelGGL((biquad_lds_sp_kernel<TI, TO, true>), grid, dim3(kLdsThreads), lds, s, nullptr,  \
                                  done, 0, static_cast<const TI *>(d_in), static_cast<TO *>(d_out), la, q_);       \
        else                                                                                           \
            hipExtLaunchKernelGGL((biquad_lds_sp_kernel<TI, TO, false>), grid, dim3(kLdsThreads), lds, s, nullptr, \
                                  done, 0, static_cast<const TI *>(d_in), static_cast<TO *>(d_out), la, q_);       \
        last_kernel = NAME;                                                                            \
    } while (0)
            if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32)
                PH_BQ(float, float, "biquad_lds_sp_kernel<f32,f32>");
            else if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F64)
                PH_BQ(double, double, "biquad_lds_sp_kernel<f64,f64>");
            else if (in_dtype == PIPE_HIP_F32)
                PH_BQ(float, double, "biquad_lds_sp_kernel<f32,f64>");
            else
                PH_BQ(double, float, "biquad_lds_sp_kernel<f64,f32>");
#undef PH_BQ
        } else if (use_lds_form()) {
            BiquadLdsArgs la{};
            la.state = a.state;
            la.frames = frames;
            la.C = cfg.channels;
            la.S = S_;
            la.cgroups = (cfg.channels + 63) / 64;
            la.gain = gain_;
            const int cg = cfg.channels < 64 ? cfg.channels : 64;
            // as many frames per block as 60 KB of LDS hold (two workgroups per CU), a multiple
            // of the chunk, never more than the call
            int64_t fb = (60 * 1024) / (int64_t)(sizeof(double) * cg);
            fb = fb / kLdsChunk * kLdsChunk;
            if (fb > frames)
                fb = (frames + kLdsChunk - 1) / kLdsChunk * kLdsChunk;
            la.fb = (int)fb;
            // planes 16-byte aligned and an odd number of 16-byte units apart: the channels'
            // lanes read different banks
            la.plane = (int)fb + ((fb / 2) % 2 == 0 ? 2 : 0);
            const size_t lds = sizeof(double) * (size_t)la.plane * (size_t)cg;
            const dim3 grid((unsigned)(nl * la.cgroups));
            hipEvent_t done = completion;  // (see the section-pipelined form above)
            completion = nullptr;
#define PH_BQ2(TI, TO, G)                                                                              \
    do {                                                                                               \
        if (S_ == 1)                                                                                   \
            hipExtLaunchKernelGGL((biquad_lds_kernel<TI, TO, 1, G>), grid, dim3(kLdsThreads), lds, s, nullptr, done, 0, \
                                  static_cast<const TI *>(d_in), static_cast<TO *>(d_out), la, q_);    \
        else if (S_ == 2)                                                                              \
            hipExtLaunchKernelGGL((biquad_lds_kernel<TI, TO, 2, G>), grid, dim3(kLdsThreads), lds, s, nullptr, done, 0, \
                                  static_cast<const TI *>(d_in), static_cast<TO *>(d_out), la, q_);    \
        else                                                                                           \
            hipExtLaunchKernelGGL((biquad_lds_kernel<TI, TO, 0, G>), grid, dim3(kLdsThreads), lds, s, nullptr, done, 0, \
                                  static_cast<const TI *>(d_in), static_cast<TO *>(d_out), la, q_);    \
    } while (0)
#define PH_BQ(TI, TO, NAME)            \
    do {                               \
        if (has_gain_)                 \
            PH_BQ2(TI, TO, true);      \
        else                           \
            PH_BQ2(TI, TO, false);     \
        last_kernel = NAME;            \
    } while (0)
            if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32)
                PH_BQ(float, float, "biquad_lds_kernel<f32,f32>");
            else if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F64)
                PH_BQ(double, double, "biquad_lds_kernel<f64,f64>");
            else if (in_dtype == PIPE_HIP_F32)
                PH_BQ(float, double, "biquad_lds_kernel<f32,f64>");
            else
                PH_BQ(double, float, "biquad_lds_kernel<f64,f32>");
#undef PH_BQ
#undef PH_BQ2
        } else {
            const dim3 grid(sblocks);
#define PH_BQ2(TI, TO, G)                                                                              \
    do {                                                                                               \
        if (S_ == 1)                                                                                   \
            hipLaunchKernelGGL((biquad_kernel<TI, TO, 1, G, kWhole>), grid, dim3(kThreads), 0, s,      \
                               static_cast<const TI *>(d_in), static_cast<TO *>(d_out), a, q_);        \
        else if (S_ == 2)                                                                              \
            hipLaunchKernelGGL((biquad_kernel<TI, TO, 2, G, kWhole>), grid, dim3(kThreads), 0, s,      \
                               static_cast<const TI *>(d_in), static_cast<TO *>(d_out), a, q_);        \
        else                                                                                           \
            hipLaunchKernelGGL((biquad_kernel<TI, TO, 0, G, kWhole>), grid, dim3(kThreads), 0, s,      \
                               static_cast<const TI *>(d_in), static_cast<TO *>(d_out), a, q_);        \
    } while (0)
#define PH_BQ(TI, TO, NAME)            \
    do {                               \
        if (has_gain_)                 \
            PH_BQ2(TI, TO, true);      \
        else                           \
            PH_BQ2(TI, TO, false);     \
        last_kernel = NAME;            \
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
        }
        PH_HIP(hipGetLastError());
        PH_TRY(timer.end(s));
        return PIPE_HIP_OK;
    }

    bool fuse_view_biquad(BiquadFuseView *v) override
    {
        if (windowed())
            return false;
        v->state = static_cast<double *>(state_.p);
        v->coeffs = &q_.c[0][0];
        v->sections = S_;
        v->relaxed = !exact_ && !env_exact_ && relaxed_ok();
        v->relaxed_f64 = relaxed_f64_;
        return true;
    }

    // every section's poles strictly inside the unit circle (the stability triangle of 1 + a1 z^-1 +
    // a2 z^-2): |a2| < 1 and |a1| < 1 + a2.  Anything else -- an integrator, a deliberately unstable
    // test section, NaN coefficients -- keeps the ordered recurrence, which is what the oracle does.
    bool stable() const
    {
        for (int s = 0; s < S_; ++s) {
            const double a1 = q_.c[s][3], a2 = q_.c[s][4];
            if (!(std::fabs(a2) < 1.0 && std::fabs(a1) < 1.0 + a2))
                return false;
        }
        return true;
    }

    // How far the relaxed forms' float64 values sit from the oracle's.  They rebuild a segment's start state as
    // M^k s + (the segment walked from zero); the oracle walks on from s.  Equal in exact arithmetic; in float64 the
    // two differ by the recurrence's own rounding noise, which a resonant section amplifies: an error in the state
    // comes back k frames later times the entries of M^k, up to ~1 / sin(w0) for poles at angle w0 (21 for the
    // 300 Hz, Q = 4 test section at 48 kHz: measured 3e-14 of full scale on a 614 K-frame stream; the soak test's worst 190 kappa eps, with kappa 88 -- the noise is a random sum over the resonance's decay time; the oracle's own
    // distance from the exact result is of the same size).  kappa = the largest entry of any power of the
    // one-frame transition matrix.  The bound include/pipe_hip.h states scales with it (one float32 ulp measured
    // at max(|y|, 2^-19 kappa x full scale)); cascades with kappa above kMaxKappa -- poles within ~1e-3 rad of
    // z = 1 -- keep the ordered recurrence.
    static constexpr double kMaxKappa = 1024.0;
    double kappa()
    {
        if (kappa_ >= 0.0)
            return kappa_;
        BiquadTransition m1;
        transition(1, &m1);
        const int n = 2 * S_;
        double p[2 * kMaxSegSections][2 * kMaxSegSections], t[2 * kMaxSegSections][2 * kMaxSegSections];
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                p[i][j] = m1.m[i][j];
        double worst = 0.0;
        for (int k = 1; k <= 65536; ++k) {
            double mx = 0.0;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    mx = std::fmax(mx, std::fabs(p[i][j]));
            worst = std::fmax(worst, mx);
            if (!(worst <= kMaxKappa) || (k > 64 && mx < 1e-3 * worst))
                break;  // (past the limit, or decayed past its peak)
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double acc = 0.0;
                    for (int l = 0; l < n; ++l)
                        acc += p[i][l] * m1.m[l][j];
                    t[i][j] = acc;
                }
            std::memcpy(p, t, sizeof p);
        }
        kappa_ = worst;
        return kappa_;
    }
    bool relaxed_ok() { return stable() && S_ <= kMaxSegSections && kappa() <= kMaxKappa; }  // (no relaxed form takes more sections)

    // The LDS-staged exact kernel runs one workgroup per Line: it wins when Lines are few enough
    // that the register form cannot fill its waves anyway (always the case for the per-buffer
    // ProcessFunc form); with thousands of series the register form's 64 busy lanes per wave do.
    static bool no_sp()  // A/B knob
    {
        static const bool v = PH_ENV_AB("PIPE_HIP_BIQUAD_NO_SP") != nullptr;
        return v;
    }
    // The section-pipelined form: always where the LDS-staged form applies; with three or more
    // sections also for more Lines, as long as its workgroups (one live wave each, two per CU) make
    // at most two rounds -- 512 Lines x 8 ch x 4 sections: 163 us against 497 us for the
    // one-lane-per-series form, whose runtime section count costs a branch nest per step.
    bool use_sp_form() const
    {
        if (S_ < 2 || no_sp())
            return false;
        if (use_lds_form())
            return true;
        static const char *env = PH_ENV_AB("PIPE_HIP_BIQUAD_LDS");
        if (env)
            return false;
        const int64_t wgs = (int64_t)active_lines() * ((cfg.channels + kSpChannels - 1) / kSpChannels);
        // (two sections: the one-lane-per-series form has a compile-time variant, 172 us against 150 for one round)
        return S_ >= 3 ? wgs <= 4 * (int64_t)cus() : wgs <= 2 * (int64_t)cus();
    }
    int cus() const
    {
        if (cus_ == 0) {
            hipDeviceProp_t prop;
            cus_ = hipGetDeviceProperties(&prop, cfg.device) == hipSuccess && prop.multiProcessorCount > 0
                       ? prop.multiProcessorCount
                       : 256;
        }
        return cus_;
    }
    mutable int cus_ = 0;
    bool use_lds_form() const
    {
        static const char *env = PH_ENV_AB("PIPE_HIP_BIQUAD_LDS");
        if (env)
            return std::atoi(env) != 0;
        return active_lines() * ((cfg.channels + 63) / 64) <= 256 && active_lines() * cfg.channels <= 2048;
    }

    void launch_scan(unsigned sblocks, hipStream_t s, const BiquadArgs &a, const BiquadTransition &mlast)
    {
        switch (S_) {
        case 1: hipLaunchKernelGGL(biquad_scan_kernel<2>, dim3(sblocks), dim3(kThreads), 0, s, a, mfull_, mlast); break;
        case 2: hipLaunchKernelGGL(biquad_scan_kernel<4>, dim3(sblocks), dim3(kThreads), 0, s, a, mfull_, mlast); break;
        case 3: hipLaunchKernelGGL(biquad_scan_kernel<6>, dim3(sblocks), dim3(kThreads), 0, s, a, mfull_, mlast); break;
        default: hipLaunchKernelGGL(biquad_scan_kernel<8>, dim3(sblocks), dim3(kThreads), 0, s, a, mfull_, mlast); break;
        }
    }

    // (3 - 4 sections through the tile kernel) the halves of the cascade as two handles of their own: their states are
    // slices of this handle's state array, copied in before and out after the two passes, so every other form (and
    // the next call, whatever form it takes) finds the state where it always is
    bool split_wanted(int nl) const
    {
        if (PH_ENV_AB("PIPE_HIP_BIQUAD_NO_SPLIT"))
            return false;
        const char *e = PH_ENV_AB("PIPE_HIP_BIQUAD_SPLIT_MAX_SERIES");
        // (one channel: the lane walk drags a cache line per lane whatever the number of Lines -- 4096 x 1 ch, 3 sections: 68)
        return (int64_t)nl * cfg.channels <= (e ? std::atoll(e) : kSplitMaxSeries) || (cfg.channels == 1 && !e);
    }
    int run_split(const void *d_in, int in_dtype, void *d_out, int out_dtype, int64_t frames, const BiquadArgs &a, hipStream_t s)
    {
        const int sa = kTileMaxSections, sb = S_ - kTileMaxSections, n = 2 * S_;
        if (!half_[0]) {
            for (int h = 0; h < 2; ++h) {
                auto c = std::make_unique<Biquad>();
                PH_TRY(c->init_common(&cfg));
                PH_TRY(c->init(&q_.c[h ? sa : 0][0], h ? sb : sa));
                c->seg_min_samples_ = 1;  // (this handle decided)
                half_[h] = std::move(c);
            }
        }
        const size_t need = sizeof(double) * (size_t)frames * (size_t)a.nseries;
        if (mid_.bytes < need)
            PH_TRY(mid_.alloc(need));
        PH_TRY(timer.begin(s));
        const size_t row = sizeof(double) * (size_t)n, wa = sizeof(double) * 2u * (size_t)sa, wb = sizeof(double) * 2u * (size_t)sb;
        double *sta = static_cast<double *>(half_[0]->state_.p) + (size_t)win_first * cfg.channels * 2u * sa;
        double *stb = static_cast<double *>(half_[1]->state_.p) + (size_t)win_first * cfg.channels * 2u * sb;
        // the one-pass tile kernel reads and writes a series' state at any stride: the halves then work on this
        // handle's array itself; any other form of a half (an A/B switch, a half that must stay exact) works on a copy
        const bool direct = !PH_ENV_AB("PIPE_HIP_BIQUAD_NO_TILE") && !PH_ENV_AB("PIPE_HIP_BIQUAD_TWO_PASS") &&
                            !PH_ENV_AB("PIPE_HIP_BIQUAD_TILE_WALK_LINES") && !PH_ENV_AB("PIPE_HIP_BIQUAD_SPLIT_COPIES") &&
                            half_[0]->relaxed_ok() && half_[1]->relaxed_ok();
        for (int h = 0; h < 2; ++h) {
            half_[h]->ext_state_ = direct ? static_cast<double *>(state_.p) : nullptr;
            half_[h]->ext_stride_ = n;
            half_[h]->ext_off_ = h ? 2 * sa : 0;
        }
        if (!direct) {
            PH_HIP(hipMemcpy2DAsync(sta, wa, a.state, row, wa, (size_t)a.nseries, hipMemcpyDeviceToDevice, s));
            PH_HIP(hipMemcpy2DAsync(stb, wb, a.state + 2 * sa, row, wb, (size_t)a.nseries, hipMemcpyDeviceToDevice, s));
        }
        for (int h = 0; h < 2; ++h)
            half_[h]->set_window(win_first, win_count);
        half_[0]->relaxed_f64_out = true;
        half_[1]->relaxed_f64_out = relaxed_f64_out || relaxed_f64_;
        half_[1]->set_post_gain(has_gain_, gain_);
        split_call_ = TileCall{d_in, d_out, in_dtype, out_dtype, frames, a.state, a.sstride, a.soff, (int)a.nseries, true, has_gain_, gain_};
        if (debug_withhold_ >= 0) {  // (halves made after the parameter was set)
            for (int h = 0; h < 2; ++h) {
                half_[h]->debug_withhold_ = debug_withhold_;
                half_[h]->debug_limit_us_ = debug_limit_us_;
            }
            debug_withhold_ = -1;
        }
        PH_TRY(half_[0]->run(d_in, in_dtype, mid_.p, PIPE_HIP_F64, frames, s));
        PH_TRY(half_[1]->run(mid_.p, PIPE_HIP_F64, d_out, out_dtype, frames, s));
        if (!direct) {
            PH_HIP(hipMemcpy2DAsync(a.state, row, sta, wa, wa, (size_t)a.nseries, hipMemcpyDeviceToDevice, s));
            PH_HIP(hipMemcpy2DAsync(a.state + 2 * sa, row, stb, wb, wb, (size_t)a.nseries, hipMemcpyDeviceToDevice, s));
        }
        PH_TRY(timer.end(s));
        const bool tiles = std::strstr(half_[0]->last_kernel, "biquad_tile_kernel") && std::strstr(half_[1]->last_kernel, "biquad_tile_kernel");
        last_kernel = tiles ? "biquad_tile_kernel<segmented, two halves of the cascade>" : "biquad_kernel<segmented, two halves of the cascade>";
        return PIPE_HIP_OK;
    }

    // (tile form, one pass) scratch for the look-back, the table A^g, the launch's epoch
    int prepare_look(BiquadLookArgs *lk, const BiquadArgs &a, int seg, int nl, unsigned grid, hipStream_t s)
    {
        const int n = 2 * S_;
        const size_t words = (size_t)a.T * (size_t)a.nseries * 2u * (size_t)n;  // per array
        const size_t need = 2 * words * sizeof(unsigned long long);
        if (look_.bytes < need) {
            PH_TRY(look_.alloc(need + need / 2));
            PH_HIP(hipMemsetAsync(look_.p, 0, look_.bytes, s));  // (tags of a previous owner of the memory)
        }
        if (!err_.p) {
            PH_TRY(err_.alloc(sizeof(int)));
            *static_cast<volatile int *>(err_.p) = 0;
            void *alias = nullptr;
            PH_HIP(hipHostGetDevicePointer(&alias, err_.p, 0));
            err_dev_ = static_cast<int *>(alias);
        }
        // two sets of per-Line counters, used in turn; a launch over other Lines than the last one starts from
        // cleared sets (the tile that clears a Line's next counter only runs in launches that cover the Line)
        const size_t tick_bytes = sizeof(unsigned) * 2u * 8u * (size_t)cfg.lines;
        // (1, 2 or 4 Lines of many tiles: a counter per XCD and Line -- biquad_tile_kernel)
        const int classes = (nl == 1 || nl == 2 || nl == 4) && a.T >= 64 ? 8 / nl : 1;
        if (!ticket_.p || tick_first_ != win_first || tick_nl_ != nl || tick_classes_ != classes) {
            if (!ticket_.p)
                PH_TRY(ticket_.alloc(tick_bytes));
            PH_HIP(hipMemsetAsync(ticket_.p, 0, tick_bytes, s));
            tick_first_ = win_first;
            tick_nl_ = nl;
            tick_classes_ = classes;
        }
        if (a.T > 1)  // (a launch of one-tile Lines draws no numbers and clears nothing)
            tick_set_ ^= 1;
        if (tab_seg_ != seg) {
            BiquadTransition ma;
            transition(seg, &ma);
            tab_host_.assign((size_t)kTileThreads * n * n, 0.0);
            long double p[2 * kTileMaxSections][2 * kTileMaxSections], t[2 * kTileMaxSections][2 * kTileMaxSections];
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    p[i][j] = i == j ? 1.0L : 0.0L;
            for (int g = 0; g < kTileThreads; ++g) {
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j)
                        tab_host_[((size_t)g * n + i) * n + j] = (double)p[i][j];
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j) {
                        long double acc = 0.0L;
                        for (int l = 0; l < n; ++l)
                            acc += (long double)ma.m[i][l] * p[l][j];
                        t[i][j] = acc;
                    }
                std::memcpy(p, t, sizeof p);
            }
            if (!tab_.p)
                PH_TRY(tab_.alloc(sizeof(double) * (size_t)kTileThreads * 16u));
            PH_HIP(hipMemcpyAsync(tab_.p, tab_host_.data(), sizeof(double) * tab_host_.size(), hipMemcpyHostToDevice, s));
            tab_seg_ = seg;
        }
        static std::atomic<unsigned> epochs{0};
        unsigned e = ++epochs;
        if (e == 0)
            e = ++epochs;
        lk->aggr = static_cast<unsigned long long *>(look_.p);
        lk->incl = lk->aggr + words;
        lk->ticket = static_cast<unsigned *>(ticket_.p) + ((size_t)tick_set_ * cfg.lines + win_first) * 8u;
        lk->ticket_next = static_cast<unsigned *>(ticket_.p) + ((size_t)(tick_set_ ^ 1) * cfg.lines + win_first) * 8u;
        lk->classes = classes;
        lk->tab = static_cast<const double *>(tab_.p);
        lk->err = err_dev_;
        lk->epoch = e;
        lk->nl = nl;
        const size_t bak = sizeof(double) * (size_t)cfg.lines * (size_t)cfg.channels * (size_t)n;
        if (state_bak_.bytes < bak)
            PH_TRY(state_bak_.alloc(bak));
        lk->state_bak = static_cast<double *>(state_bak_.p);
        // ~4 s of the shader clock: a predecessor that a context switch took away is back long before that
        lk->spin_ticks = 1ull << 33;
        lk->withhold = -1;
        if (debug_withhold_ >= 0) {
            lk->withhold = debug_withhold_;
            lk->spin_ticks = (unsigned long long)(debug_limit_us_ * 2000.0);  // (~2 ticks a nanosecond)
            debug_withhold_ = -1;
        }
        err_checked_ = false;
        return PIPE_HIP_OK;
    }

    // (tile form, many tiles) one wave per series; the table (M^R)^(2^j) by squaring in long double
    static constexpr int kWaveScanMinTiles = 32;
    // (A/B: PIPE_HIP_BIQUAD_TILE_WALK_LINES=n sends n or more Lines of 6+ channels to the lane walk, the rule before the
    // one-pass tile kernel: 512 x 8 ch 402 against 500, 4096 x 8 ch 355 against 625)
    static constexpr int kTileWalkLines = 1 << 30, kTileWalkChannels = 6;
    void launch_scan_wave(hipStream_t s, const BiquadArgs &a)
    {
        const int R = (a.T + 63) / 64, n = 2 * S_;
        if (sp_len_ != mfull_len_ || sp_R_ != R) {
            long double b[4][4], r[4][4], t[4][4];
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    b[i][j] = mfull_.m[i][j];
                    r[i][j] = i == j ? 1.0L : 0.0L;
                }
            auto mul = [&](long double(*x)[4], long double(*y)[4]) {  // x = x * y
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j) {
                        long double acc = 0.0L;
                        for (int k = 0; k < n; ++k)
                            acc += x[i][k] * y[k][j];
                        t[i][j] = acc;
                    }
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j)
                        x[i][j] = t[i][j];
            };
            for (int e = R; e > 0; e >>= 1) {
                if (e & 1)
                    mul(r, b);
                mul(b, b);
            }
            std::memset(&sp_, 0, sizeof sp_);
            for (int jp = 0; jp < 6; ++jp) {
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j)
                        sp_.m[jp][i][j] = (double)r[i][j];
                mul(r, r);
            }
            sp_len_ = mfull_len_;
            sp_R_ = R;
        }
        if (S_ == 1)
            hipLaunchKernelGGL(biquad_scan_wave_kernel<2>, dim3((unsigned)a.nseries), dim3(64), 0, s, a, mfull_, mlast_, sp_, R);
        else
            hipLaunchKernelGGL(biquad_scan_wave_kernel<4>, dim3((unsigned)a.nseries), dim3(64), 0, s, a, mfull_, mlast_, sp_, R);
    }

    // zero-input transition of the cascade over `len` frames: column j = the state after len
    // frames of silence started from unit state j (order s1_0, s2_0, s1_1, s2_1, ...)
    void transition(int len, BiquadTransition *m) const
    {
        const int n = 2 * S_;
        std::memset(m, 0, sizeof *m);
        for (int j = 0; j < n; ++j) {
            long double st[2 * kMaxSections] = {0};
            st[j] = 1.0L;
            for (int f = 0; f < len; ++f) {
                long double x = 0.0L;
                for (int sct = 0; sct < S_; ++sct) {
                    const long double b0 = q_.c[sct][0], b1 = q_.c[sct][1], b2 = q_.c[sct][2];
                    const long double a1 = q_.c[sct][3], a2 = q_.c[sct][4];
                    const long double y = b0 * x + st[2 * sct];
                    st[2 * sct] = -a1 * y + (b1 * x + st[2 * sct + 1]);
                    st[2 * sct + 1] = -a2 * y + b2 * x;
                    x = y;
                }
            }
            for (int i = 0; i < n; ++i)
                m->m[i][j] = (double)st[i];
        }
    }

private:
    int S_ = 1;
    bool has_gain_ = false;
    double gain_ = 1.0;
    BiquadCoeffs q_{};
    DevBuf state_, seg_, state_alt_;
    bool last_oop_ = false;
    size_t state_bytes_ = 0;
    bool exact_ = false;
    const bool env_exact_ = std::getenv("PIPE_HIP_BIQUAD_EXACT") != nullptr;
    static constexpr int64_t kTileLatencyFrames = 1024;
    const bool seg_min_from_env_ = std::getenv("PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES") != nullptr;
    int64_t seg_min_samples_ = std::getenv("PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES")
                                         ? std::atoll(std::getenv("PIPE_HIP_BIQUAD_SEG_MIN_SAMPLES"))
                                         : (int64_t)1 << 20;
    // Shortest Line (frames a call) for the LDS-tile form; shorter Lines leave the tiles mostly empty and keep the lane walk.
    // By channels since round 6 (scripts/dev/tile_min_frames_probe.py, profiles/r06_dispatch_audit.txt: 128 .. 512 frames x
    // 1 .. 8 channels over 2^21 and 2^23 samples): 8 channels from 256 frames (8.5 against 11.4 us), 2 and 4 channels from 320
    // (2 ch x 384 frames x 2730 Lines: 14.4 against 24.3 us), the others from 384 -- it was 512 for all (mono Lines of 256
    // frames: the tile form 30.0 us, the lane walk 22.8).  PIPE_HIP_BIQUAD_TILE_MIN_FRAMES set: that count for all.
    const int64_t tile_min_frames_env_ = std::getenv("PIPE_HIP_BIQUAD_TILE_MIN_FRAMES")
                                             ? std::atoll(std::getenv("PIPE_HIP_BIQUAD_TILE_MIN_FRAMES"))
                                             : -1;
    int64_t tile_min_frames() const
    {
        if (tile_min_frames_env_ >= 0)
            return tile_min_frames_env_;
        return cfg.channels == 8 ? 256 : (cfg.channels == 2 || cfg.channels == 4 ? 320 : 384);
    }
    static int walk_lines_knob()
    {
        const char *e = PH_ENV_AB("PIPE_HIP_BIQUAD_TILE_WALK_LINES");
        return e ? std::atoi(e) : kTileWalkLines;
    }
    const int tile_walk_lines_ = walk_lines_knob();
    BiquadTransition mfull_{}, mlast_{};
    int mfull_len_ = -1, mlast_len_ = -1;  // (the lane-walk form computes its own last-segment matrix per call)
    BiquadTilePowers pw_{};
    int pw_seg_ = -1;
    double kappa_ = -1.0;
    std::unique_ptr<Biquad> half_[2];
    double *ext_state_ = nullptr;
    int ext_stride_ = 0, ext_off_ = 0;
    DevBuf mid_;
    // (3 sections, 16.7 M samples: 1 Line x 2 ch 113 Gsamples/s against the lane walk's 12, 64 x 2 ch 120 against 48,
    // 2048 x 2 ch even; 512 x 8 ch 127 against 218)
    static constexpr int64_t kSplitMaxSeries = 2048;
    DevBuf look_, ticket_, tab_, state_bak_;
    PinnedBuf err_;
    int *err_dev_ = nullptr;
    bool err_checked_ = true;
    int tick_set_ = 0, tick_first_ = -1, tick_nl_ = -1, tick_classes_ = -1;
    std::vector<double> tab_host_;
    int tab_seg_ = -1;
    BiquadScanPowers sp_{};
    int sp_len_ = -1, sp_R_ = -1;
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
