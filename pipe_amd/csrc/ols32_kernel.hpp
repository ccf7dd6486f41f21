// The kernel of the 32 x 32 overlap-save form (see fir_ols32.hip), with an optional epilogue that
// runs the FIR's output tile through a biquad cascade and a gain before it is stored
// (chain_fused.hip).  Included by both files; each instantiates what it launches.
#pragma once

#include <type_traits>
#include <utility>

#include <hip/hip_runtime.h>

#include "common.hpp"
#include "fir_hist.hpp"
#include "ols32_core.hpp"

namespace pipehip {
namespace ols {

constexpr int kWaves32 = 8;             // waves per workgroup = per CU
#ifndef PH_FUSE_ABLATE
#define PH_FUSE_ABLATE 0
#endif

// -DPH_FUSE_PROF: per-phase s_memtime sums of every wave of the fused kernel (a debug build of
// the library, scripts/build_prof_lib.sh; never the shipped one)
#ifdef PH_FUSE_PROF
constexpr int kFuseProfPhases = 13;
#define PH_FSTAMP(i)                                                          \
    do {                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                    \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();         \
        fprof_acc[(i)] += now_ - fprof_last;                                  \
        fprof_last = now_;                                                    \
        __builtin_amdgcn_sched_barrier(0);                                    \
    } while (0)
#define PH_FPROF_PARAMS , unsigned long long (&fprof_acc)[kFuseProfPhases], unsigned long long &fprof_last
#define PH_FPROF_ARGS , fprof_acc, fprof_last
// ... and, with -DPH_FUSE_TIMELINE on top, a timeline: absolute s_memtime of five events of each
// wave's first kTlUnits units (before the round gate, unit start, transform done, epilogue done, stores
// issued), behind the phase sums.  (Its stores sit in the waves' vmcnt order: the phase sums of such a
// build are distorted, read only the timeline from it.)
constexpr int kTlEvents = 5, kTlUnits = 4;
constexpr size_t kTlOffset = (size_t)kFuseProfPhases * 8 * 4096;
#endif
#if defined(PH_FUSE_PROF) && defined(PH_FUSE_TIMELINE)
#define PH_FTIME(ev)                                                                                         \
    do {                                                                                                     \
        if (fa.prof && tl_unit < kTlUnits && lane == 0)                                                      \
            fa.prof[kTlOffset + (((size_t)blockIdx.x * 8 + wave) * kTlUnits + tl_unit) * kTlEvents + (ev)] = \
                __builtin_amdgcn_s_memtime();                                                                \
    } while (0)
#else
#define PH_FTIME(ev) \
    do {             \
    } while (0)
#endif
#ifndef PH_FUSE_PROF
#define PH_FSTAMP(i) \
    do {             \
    } while (0)
#define PH_FPROF_PARAMS
#define PH_FPROF_ARGS
#endif
constexpr unsigned kOut32 = 0x80000000u;  // a buffer offset beyond any num_records: loads 0, stores dropped

struct Args32 {
    int64_t frames;       // frames per Line in this call
    int64_t line_stride;  // elements between Lines
    int C, N, H;          // channels, taps, history frames (N - 1)
    int HP;               // window frames ahead of a tile's first output: H, or H rounded up to 32
                          // (fused chain: a tile's first output then opens a 32-position segment)
    int L;                // valid outputs per tile = 1024 - HP
    int pairs, lines, tiles_per_line;
    int ipl, upl;         // items per Line (tiles x pairs); units (item pairs) per Line
    int local;            // fused chain: every Line's tiles run in ONE workgroup, records in LDS (below)
    int stagger;          // fused chain: the second wave of every SIMD starts this many s_sleep(127) late
    int64_t nunits;
    int d_slot, d_line;   // the wave stride of the launch as (slot, Line) digits
    int64_t mono_shift;   // MONO kernels (one channel): the frames between the two tiles a half-wave's complex sequence carries
    int odd;              // C is odd: the last "pair" is ONE channel (its imaginary part: whatever follows it in memory,
                          // finite and unused -- real taps keep the parts apart; only its real part is stored; the
                          // fused chain's epilogue runs the phantom channel like any other and drops it)
    void *hist_new;       // float64 elements (S = 0), the stream's float32 (fused chain, S > 0)
};

// ---- the biquad + gain epilogue of the fused chain (chain_fused.hip) ---------------------------
// Arguments that are the same for every tile of a launch.
template <int S>
struct FuseConst {
    double c[S][5];  // {b0, b1, b2, a1, a2} per section
    double gain;     // 1.0 when the chain has no gain stage (x * 1.0 is x, bit for bit)
    int D;           // (M^L)^j is below 2^-60 from j = D on (2^30: never within a look-back window)
};
// Zero-input state transitions of the S-section DF2T cascade (state order s1_0, s2_0, s1_1, s2_1,
// ...; M = one frame), computed on the host in long double; 2S x 2S matrices, row-major, in ONE
// device array at these matrix offsets:
constexpr int kMatAk = 0;    // [17]  (M^32)^j, j = 0..16 : the scan over a tile's 32 segments
constexpr int kMatML = 17;   //       M^L                 : one whole tile
constexpr int kMatT32 = 18;  //       (M^L)^32            : one look-back window
constexpr int kMatTj = 19;   // [33]  (M^L)^j, j = 0..32  : a predecessor at distance j
constexpr int kMatPk = 52;   // [32]  M^(32 (k - k0)) for k >= k0, else 1: a segment's offset in the tile
constexpr int kMatWw = 84;   // [33]  (M^L)^(32 w), w = 0..32: w whole look-back windows
constexpr int kMatGz = 117;  // [16]  32 x {g0, g1}: g_i = M^(31 - i) c, the weight of a segment's sample i in its
                             //       zero-start end state (c: what one input sample adds to the state); S = 1
constexpr int kMatCount = 133;
constexpr int kMaxWindows = 32;  // a look-back that needs more (> 1024 tiles to the nearest P) gives up
struct FuseArgs {
    int k0;                      // HP / 32: the lane (segment) of a tile's first output; HP % 32 == 0
    unsigned epoch;              // tag of this launch's records (never 0)
    unsigned long long *rec;     // [series][tile][A | P][2 NV] 8-byte {tag, half a double} granules
    unsigned long long *own;     // [series][2 slots][2 NV] granules: the cascade's state between launches
                                 // (own_read / own_write below)
    double *seg_state;           // [series][2 channels][2S]: start state of the segment that holds the
                                 // Line's last frame (for chain_tail_kernel)
    const double *mats;          // the matrices above
    int *err;                    // set when a bounded spin gives up
    unsigned long long spin_ticks;  // ... after this many s_memtime ticks without the record it waits for (seconds)
    int withhold;                // debug (PIPE_HIP_PARAM_DEBUG): tiles with this index publish nothing (-1: none)
    unsigned long long *prof;    // PH_FUSE_PROF builds: [waves][kFuseProfPhases]
};

template <int S, typename F, int... Is>
__device__ __forceinline__ void for_each_section_impl(F &&f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int S, typename F>
__device__ __forceinline__ void for_each_section(F &&f)
{
    for_each_section_impl<S>(static_cast<F &&>(f), std::make_integer_sequence<int, S>{});
}

template <int S>
__device__ __forceinline__ double biquad_step(double x, double (&st)[2 * S], const FuseConst<S> &fc)
{
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const double y = __builtin_fma(fc.c[s][0], x, st[2 * s]);
        const double t = __builtin_fma(fc.c[s][1], x, st[2 * s + 1]);
        st[2 * s] = __builtin_fma(-fc.c[s][3], y, t);
        const double u = fc.c[s][2] * x;
        st[2 * s + 1] = __builtin_fma(-fc.c[s][4], y, u);
        x = y;
    }
    return x;
}

// out = z + m * v  (2S x 2S)
template <int S>
__device__ __forceinline__ void affine(double (&out)[2 * S], const double (&z)[2 * S], const double (&m)[2 * S][2 * S],
                                       const double (&v)[2 * S])
{
#pragma unroll
    for (int i = 0; i < 2 * S; ++i) {
        double acc = z[i];
#pragma unroll
        for (int j = 0; j < 2 * S; ++j)
            acc = __builtin_fma(m[i][j], v[j], acc);
        out[i] = acc;
    }
}
template <int S>
__device__ __forceinline__ void load_mat(double (&m)[2 * S][2 * S], const double *__restrict__ mats, int index)
{
    const double *src = mats + (size_t)index * (2 * S) * (2 * S);
#pragma unroll
    for (int i = 0; i < 2 * S; ++i)
#pragma unroll
        for (int j = 0; j < 2 * S; ++j)
            m[i][j] = src[i * 2 * S + j];
}

// cross-lane moves of a double inside 16-lane rows, on the VALU (no LDS round trip):
// CTRL 0x110 + d: row_shr:d (lanes whose source falls outside the row get 0);
// CTRL 0x142 with ROWS 0xA: lane 15 of rows 0 / 2 to every lane of rows 1 / 3 (the others get 0)
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_f64(double x)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROWS, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROWS, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ unsigned long long granule_load(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(unsigned long long *p, unsigned tag, unsigned v)
{
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A spin on a predecessor's record is bounded by TIME (a preempted predecessor may be away for milliseconds; a
// count of polls says nothing): every 256th poll reads the clock, the first reading starts it.
__device__ __forceinline__ bool spin_expired(unsigned &spins, unsigned long long &t0, unsigned long long limit)
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

// ---- the cascade's state between launches -------------------------------------------------------
// The state after a Line's last frame is written by the wave that runs the Line's LAST tile and read
// by the waves of its FIRST tiles -- of the same launch, in no particular order.  Two slots per
// series, every granule tagged with the epoch of the launch that wrote it: the writer takes the
// slot with the OLDER tag, the reader the NEWER one that is not this launch's.  Neither ever waits.
//   slot layout: [2 NV] granules = NV doubles (channel 0's 2S states, channel 1's), low word first
template <int NV>
__device__ __forceinline__ int own_newer_slot(const unsigned long long *o, unsigned epoch)
{
    const unsigned a0 = epoch - (unsigned)(granule_load(o) >> 32);
    const unsigned a1 = epoch - (unsigned)(granule_load(o + 2 * NV) >> 32);
    // age 0: being written by this launch
    return a0 == 0u ? 1 : (a1 == 0u ? 0 : (a1 < a0 ? 1 : 0));
}
template <int NV>
__device__ __forceinline__ void own_read(const unsigned long long *o, unsigned epoch, double (&pay)[NV])
{
    const unsigned long long *r = o + own_newer_slot<NV>(o, epoch) * (2 * NV);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const unsigned long long g0 = granule_load(r + 2 * j), g1 = granule_load(r + 2 * j + 1);
        pay[j] = __builtin_bit_cast(double, (g1 << 32) | (g0 & 0xFFFFFFFFull));
    }
}
// (channel threads of chain_tail_kernel write half a slot each: the slot choice depends only on
// tags of earlier launches, or on "already this launch's")
template <int NV>
__device__ __forceinline__ unsigned long long *own_write_slot(unsigned long long *o, unsigned epoch)
{
    return o + (1 - own_newer_slot<NV>(o, epoch)) * (2 * NV);
}
__device__ __forceinline__ void own_store(unsigned long long *dst, unsigned epoch, double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    granule_store(dst, epoch, (unsigned)b);
    granule_store(dst + 1, epoch, (unsigned)(b >> 32));
}

// ---- block-local look-back --------------------------------------------------------------------
// A hand-off between workgroups costs 2-3 us under load (the poll waits in the consumer CU's memory
// queue).  When there are at least as many Lines as workgroups, a workgroup takes whole Lines, so a
// tile's predecessors run in the same workgroup -- in the same round of 16 items or the one before --
// and the aggregates pass through a ring of records in LDS instead: kLocalRing records of
// {NV doubles, tag}, slot = item index % kLocalRing, tag = item index / kLocalRing + 1 (the ring is
// zeroed at kernel start).  A wave starts round r only when every wave has finished round r - 2
// (four counters in LDS, one per round mod 4): waves drift apart by at most one round, so the record
// of item g -- read by items up to g + D * pairs <= g + 32, i.e. in rounds <= round(g) + 2 -- is
// overwritten (by item g + 64, in round(g) + 4) only after its last reader's round has ended.  A
// plain barrier per round would do, and costs a fifth of the kernel in waves waiting for the slowest.
constexpr int kLocalRing = 64;
constexpr int kLocalReach = 32;  // the host checks D * pairs against this
// Two sections: one ring per section, 48 records each, reach 16.  A record of round r is then read in
// rounds <= r + 1 and overwritten by item g + 48 in round r + 3, whose wave starts only when every wave is
// through round r + 1: the same argument with one round less of reach.
constexpr int kLocalRing2 = 48;
constexpr int kLocalReach2 = 16;
template <int NV>
struct LocalRec {
    double v[NV];
    unsigned long long tag;
};

// One tile (FIR output in lo/hi, natural layout, re/im = the pair's two channels) through the
// cascade, in place.  Per half-wave = per item:
//   1. transpose to SEGMENT layout through the item's plane: lane k owns window positions
//      [32 k, 32 k + 32); positions below HP = 32 k0 -- no output -- become zeros;
//   2. every lane runs its segment from a ZERO state (the exact recurrence) and keeps the end state;
//   3. a scan over the 32 lanes (s_{k+1} = z_k + M^32 s_k) gives every segment's start state for
//      a tile that starts from zero, and the tile's own map (M^L, Z);
//   4. decoupled look-back over the predecessor tiles of the series gives the tile's true start
//      state: tiles publish their zero-start aggregate (A) as soon as they have it and their true
//      end state (P) when they know it; a tile combines the A's back to the nearest P;
//   5. every lane re-runs its segment with the SAME ordered fma recurrence as the exact kernel,
//      from its true start state, applies the gain, and the tile goes back to natural layout.
// Float64 values differ from the ordered recurrence only through the start states (O(1e-16)
// relative, reassociation of steps 3-4), exactly like the time-segmented biquad.
template <int S, bool GENERAL, bool LOCAL>
__device__ __forceinline__ void fused_epilogue(cd (&lo)[16], cd (&hi)[16], double *pa, double *pb, const Args32 &a,
                                               const FuseArgs &fa, const FuseConst<S> &fc, int line, int tile, int pair,
                                               bool valid, int l5_in, int half, LocalRec<4 * S> *ring,
                                               int gi PH_FPROF_PARAMS)
{
    static_assert(!(GENERAL && LOCAL), "the block-local records serve the forgetful form");
    // The epilogue's per-lane table addresses are invariant over the unit loop; hoisted out of it
    // they would sit in ~40 registers through the transform, which has none to give.  An opaque
    // copy of the lane index keeps them inside.
    int l5 = l5_in;
    asm volatile("" : "+v"(l5));
    constexpr int N2 = 2 * S;
    constexpr int NV = 2 * N2;  // doubles per record: two channels x 2S states
    const int64_t len64 = a.frames - (int64_t)tile * a.L;
    const int len = (int)(len64 < a.L ? len64 : a.L);  // output frames of this tile (<= 0: none)
    const bool last_tile = tile == a.tiles_per_line - 1;
    valid = valid && len > 0;
    const int64_t series = (int64_t)line * a.pairs + pair;
    unsigned long long *recs = fa.rec + (series * a.tiles_per_line) * (2 * 2 * NV);  // this series' records

    // requested now, wanted later: the first scan matrix
    double m0[N2][N2];
    load_mat<S>(m0, fa.mats, kMatAk + 1);

    // ---- 1. segment layout, 2. zero-state pass ----------------------------------------------
    // Channel 0's segment stays in registers; channel 1's stays in the plane (it is the last one
    // written there) and is read again in step 5: 64 registers less across the look-back.
    double xr[32];
    double zr[N2], zi[N2];
#pragma unroll
    for (int i = 0; i < N2; ++i)
        zr[i] = zi[i] = 0.0;
#pragma unroll
    for (int r = 0; r < 32; ++r)
        PH_COL(r) = PH_NAT(r).re;
    wave_fence();
#pragma unroll
    for (int c = 0; c < 32; ++c)
        xr[c] = PH_ROW(c);
    wave_fence();
#pragma unroll
    for (int r = 0; r < 32; ++r)
        PH_COL(r) = PH_NAT(r).im;
    wave_fence();
    {
        double xz[32];
#pragma unroll
        for (int c = 0; c < 32; ++c)
            xz[c] = PH_ROW(c);
        if constexpr (S == 1) {
            // the zero-start end state is linear in the segment's samples: z = sum_i g_i x_i, two fma per
            // sample and channel with no chain between them (the recurrence takes five, each waiting for
            // the one before); the weights are the same for every lane: scalar loads
            typedef const __attribute__((address_space(4))) double *const_f64;
            const const_f64 gz = (const_f64)(fa.mats + kMatGz * 4);
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const double g0 = gz[2 * c], g1 = gz[2 * c + 1];
                zr[0] = __builtin_fma(g0, xr[c], zr[0]);
                zr[1] = __builtin_fma(g1, xr[c], zr[1]);
                zi[0] = __builtin_fma(g0, xz[c], zi[0]);
                zi[1] = __builtin_fma(g1, xz[c], zi[1]);
            }
        } else {
            // the two channels' chains interleaved: each step is two dependent fma per section
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                (void)biquad_step<S>(xr[c], zr, fc);
                (void)biquad_step<S>(xz[c], zi, fc);
            }
        }
    }
    // positions below HP = 32 k0 carry no output (the transform's circular wrap): the lanes that
    // own them contribute nothing to the scan, and what they compute later is never stored
    if (l5 < fa.k0) {
#pragma unroll
        for (int i = 0; i < N2; ++i)
            zr[i] = zi[i] = 0.0;
    }
    PH_FSTAMP(1);  // to segments + zero-state pass

    // ---- 3. scan over the half-wave's 32 segments: inside 16-lane rows on the VALU, then row 0's
    //         total into row 1.  Each step's matrix is requested a step ahead (the first one before
    //         the zero-state pass): one memory round trip per step would cost more than the scan.
    {
        double mn[N2][N2], tr[N2], ti[N2];
#define PH_SCAN_STEP(D_, NEXT_)                                  \
    load_mat<S>(mn, fa.mats, (NEXT_));                           \
    _Pragma("unroll") for (int j = 0; j < N2; ++j)               \
    {                                                            \
        tr[j] = dpp_f64<0x110 + (D_), 0xF>(zr[j]);               \
        ti[j] = dpp_f64<0x110 + (D_), 0xF>(zi[j]);               \
    }                                                            \
    affine<S>(zr, zr, m0, tr);                                   \
    affine<S>(zi, zi, m0, ti);                                   \
    _Pragma("unroll") for (int i = 0; i < N2; ++i)               \
        _Pragma("unroll") for (int j = 0; j < N2; ++j) m0[i][j] = mn[i][j];
        PH_SCAN_STEP(1, kMatAk + 2)
        PH_SCAN_STEP(2, kMatAk + 4)
        PH_SCAN_STEP(4, kMatAk + 8)
        PH_SCAN_STEP(8, kMatAk + (l5 & 15) + 1)  // next: lanes 16..31 need (M^32)^(l5 - 15)
#undef PH_SCAN_STEP
#pragma unroll
        for (int j = 0; j < N2; ++j) {
            tr[j] = dpp_f64<0x142, 0xA>(zr[j]);
            ti[j] = dpp_f64<0x142, 0xA>(zi[j]);
        }
        affine<S>(zr, zr, m0, tr);
        affine<S>(zi, zi, m0, ti);
    }
    // zr / zi: end state of segment l5 for a zero tile start.  Exclusive form and the tile's aggregate:
    double er[N2], ei[N2], Zr[N2], Zi[N2];
#pragma unroll
    for (int j = 0; j < N2; ++j) {
        const double sr1 = dpp_f64<0x111, 0xF>(zr[j]), si1 = dpp_f64<0x111, 0xF>(zi[j]);
        const double br = dpp_f64<0x142, 0xA>(zr[j]), bi = dpp_f64<0x142, 0xA>(zi[j]);
        er[j] = l5 == 16 ? br : sr1;
        ei[j] = l5 == 16 ? bi : si1;
        Zr[j] = __shfl(zr[j], 31, 32);
        Zi[j] = __shfl(zi[j], 31, 32);
    }
    PH_FSTAMP(2);  // scan

    // ---- 4. the tile's true start state ---------------------------------------------------------
    const bool held = fa.withhold >= 0 && tile == fa.withhold;  // (debug: this tile's records never show up)
    auto publish = [&](int kind, const double (&vr)[N2], const double (&vi)[N2]) {
        if (l5 == 31 && valid && !last_tile && !held) {
            unsigned long long *dst = recs + ((int64_t)tile * 2 + kind) * (2 * NV);
#pragma unroll
            for (int j = 0; j < N2; ++j) {
                const unsigned long long br = __builtin_bit_cast(unsigned long long, vr[j]);
                const unsigned long long bi = __builtin_bit_cast(unsigned long long, vi[j]);
                granule_store(dst + 2 * j, fa.epoch, (unsigned)br);
                granule_store(dst + 2 * j + 1, fa.epoch, (unsigned)(br >> 32));
                granule_store(dst + 2 * (N2 + j), fa.epoch, (unsigned)bi);
                granule_store(dst + 2 * (N2 + j) + 1, fa.epoch, (unsigned)(bi >> 32));
            }
        }
    };
    if constexpr (LOCAL) {
        if (l5 == 31 && valid && !last_tile && !held) {
            LocalRec<NV> *r = ring + (gi & (kLocalRing - 1));
#pragma unroll
            for (int j = 0; j < N2; ++j) {
                r->v[j] = Zr[j];
                r->v[N2 + j] = Zi[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_store(&r->tag, (unsigned long long)(gi / kLocalRing + 1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        publish(0, Zr, Zi);  // A: the aggregate, before anything is waited for
    }
    PH_FSTAMP(3);  // publish A

    // for the forgetful form: this lane's power of M^L and -- if this lane stands for "tile -1" --
    // the stage's own state; requested here (not at the top: 16 registers held through the scan
    // were 16 too many), they arrive while the predecessors are polled
    double tj0[N2][N2], own[NV];
    if constexpr (!GENERAL) {
        load_mat<S>(tj0, fa.mats, kMatTj + l5);
        const bool mine = valid && tile - 1 - l5 == -1 && l5 < fc.D;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            own[j] = 0.0;
        if (mine)
            own_read<NV>(fa.own + series * (2 * 2 * NV), fa.epoch, own);
    }

    PH_FSTAMP(10);  // look-back: own state / matrix requests
    double sr[N2], si[N2];  // start state of the tile
#pragma unroll
    for (int j = 0; j < N2; ++j)
        sr[j] = si[j] = 0.0;
    if constexpr (!GENERAL) {
        // The filter forgets within one look-back window ((M^L)^j below 2^-60 from j = D <= 32 on):
        // the start state is the sum of the D nearest predecessors' zero-start aggregates,
        //     s = sum_{j < D} (M^L)^j A_{t-1-j}      (the stage's own state stands in for tile -1),
        // in this fixed order -- no P records, no dependence on how far other tiles have come.
        const int u = tile - 1 - l5;  // lane j looks at predecessor t - 1 - j
        const bool need = valid && l5 < fc.D && u >= -1;
        double wr[N2], wi[N2];
#pragma unroll
        for (int j = 0; j < N2; ++j)
            wr[j] = wi[j] = 0.0;
        if (need && u == -1) {
#pragma unroll
            for (int j = 0; j < N2; ++j) {
                wr[j] = own[j];
                wi[j] = own[N2 + j];
            }
        }
        bool ready = !(need && u >= 0);
        unsigned spins = 0;
        unsigned long long spin_t0 = 0;
        if constexpr (LOCAL) {
            const int gp = gi - (l5 + 1) * a.pairs;  // predecessor t - 1 - l5 of this pair
            const LocalRec<NV> *r = ring + (gp & (kLocalRing - 1));
            const unsigned long long want = (unsigned long long)(gp / kLocalRing + 1);
#if PH_FUSE_ABLATE == 5  // (5: the predecessors' records are taken as they are, nobody waits)
            ready = true;
#endif
            while (!__all(ready)) {
                if (!ready) {
                    if (__hip_atomic_load(&r->tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == want) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                        for (int j = 0; j < N2; ++j) {
                            wr[j] = r->v[j];
                            wi[j] = r->v[N2 + j];
                        }
                        ready = true;
                    } else {
                        __builtin_amdgcn_s_sleep(1);
                        if (spin_expired(spins, spin_t0, fa.spin_ticks)) {
                            *fa.err = 1;
                            ready = true;
                        }
                    }
                }
            }
        }
        while (!LOCAL && !__all(ready)) {
            if (!ready) {
                const unsigned long long *r = recs + (int64_t)u * (2 * 2 * NV);
                double pay[NV];
                bool ok = true;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const unsigned long long g0 = granule_load(r + 2 * j), g1 = granule_load(r + 2 * j + 1);
                    ok = ok && (unsigned)(g0 >> 32) == fa.epoch && (unsigned)(g1 >> 32) == fa.epoch;
                    pay[j] = __builtin_bit_cast(double, (g1 << 32) | (g0 & 0xFFFFFFFFull));
                }
                if (ok) {
#pragma unroll
                    for (int j = 0; j < N2; ++j) {
                        wr[j] = pay[j];
                        wi[j] = pay[N2 + j];
                    }
                    ready = true;
                } else {
                    __builtin_amdgcn_s_sleep(1);
                    if (spin_expired(spins, spin_t0, fa.spin_ticks)) {  // seconds: something is wrong; give up loudly
                        *fa.err = 1;
                        ready = true;
                    }
                }
            }
        }
        PH_FSTAMP(11);  // look-back: waiting for the predecessors' records
        double zero[N2], vr[N2], vi[N2];
#pragma unroll
        for (int j = 0; j < N2; ++j)
            zero[j] = 0.0;
        affine<S>(vr, zero, tj0, wr);
        affine<S>(vi, zero, tj0, wi);
        const int nd = fc.D < 32 ? fc.D : 32;
        for (int d = 0; d < nd; ++d) {  // uniform trip count, usually 1..3
#pragma unroll
            for (int j = 0; j < N2; ++j) {
                sr[j] += __shfl(vr[j], d, 32);
                si[j] += __shfl(vi[j], d, 32);
            }
        }
    } else {
        {
            int base = tile - 1;  // newest predecessor of the current window
            int dist0 = 0;        // how many predecessors lie between it and the tile (32 per window passed)
            bool done = !valid;
            unsigned spins = 0;
            unsigned long long spin_t0 = 0;
            while (!__all(done)) {
                const int u = base - l5;  // the predecessor this lane looks at (-1: the stage's own state)
                // tags first (the first granule of each record); the payload of the chosen record is
                // fetched after the decision: fewer registers in flight than both records whole
                const bool ask = !done && u >= 0 && dist0 + l5 < fc.D;
                const unsigned long long *r = recs + (int64_t)(ask ? u : 0) * (2 * 2 * NV);
                unsigned ta = 0, tp = 0;
                if (ask) {
                    ta = (unsigned)(granule_load(r) >> 32);
                    tp = (unsigned)(granule_load(r + 2 * NV) >> 32);
                }
                // 2: a P (or nothing to add), 1: an A, 0: not there yet
                const int st = !ask ? 2 : (tp == fa.epoch ? 2 : (ta == fa.epoch ? 1 : 0));
                const unsigned long long bp = __ballot(st == 2), b0 = __ballot(st == 0);
                const unsigned mp = (unsigned)(bp >> (32 * half)), m0 = (unsigned)(b0 >> (32 * half));
                const int jp = mp ? __builtin_ctz(mp) : 32, j0 = m0 ? __builtin_ctz(m0) : 32;
                const bool resolved = jp < j0;            // everything nearer than the first P is an A
                const bool whole = jp == 32 && j0 == 32;  // 32 A's: take them all and look further back
                if (!done && (resolved || whole)) {
                    const int jlim = resolved ? jp : 31;
                    double vr[N2], vi[N2];
    #pragma unroll
                    for (int j = 0; j < N2; ++j)
                        vr[j] = vi[j] = 0.0;
                    if (l5 <= jlim && (st == 1 || l5 == jp) && (ask || u == -1)) {
                        double wr[N2], wi[N2];
                        if (u == -1) {  // the series starts here: the biquad stage's own state
                            double pay[NV];
                            own_read<NV>(fa.own + series * (2 * 2 * NV), fa.epoch, pay);
    #pragma unroll
                            for (int j = 0; j < N2; ++j) {
                                wr[j] = pay[j];
                                wi[j] = pay[N2 + j];
                            }
                        } else {
                            const unsigned long long *rr = r + (st == 2 ? 2 * NV : 0);
                            double pay[NV];
                            for (unsigned tries = 0;; ++tries) {  // the other granules land with the first
                                bool ok = true;
    #pragma unroll
                                for (int j = 0; j < NV; ++j) {
                                    const unsigned long long g0 = granule_load(rr + 2 * j), g1 = granule_load(rr + 2 * j + 1);
                                    ok = ok && (unsigned)(g0 >> 32) == fa.epoch && (unsigned)(g1 >> 32) == fa.epoch;
                                    pay[j] = __builtin_bit_cast(double, (g1 << 32) | (g0 & 0xFFFFFFFFull));
                                }
                                if (ok)
                                    break;
                                if (tries > (1u << 20)) {
                                    *fa.err = 2;
                                    break;
                                }
                            }
    #pragma unroll
                            for (int j = 0; j < N2; ++j) {
                                wr[j] = pay[j];
                                wi[j] = pay[N2 + j];
                            }
                        }
                        double tj[N2][N2];
                        load_mat<S>(tj, fa.mats, kMatTj + l5);
                        double zero[N2];
    #pragma unroll
                        for (int j = 0; j < N2; ++j)
                            zero[j] = 0.0;
                        affine<S>(vr, zero, tj, wr);  // (M^L)^l5 applied to this predecessor's contribution
                        affine<S>(vi, zero, tj, wi);
                    }
                    // sum over the half-wave's lanes, then through the windows already passed
    #pragma unroll
                    for (int sh = 1; sh < 32; sh <<= 1) {
    #pragma unroll
                        for (int j = 0; j < N2; ++j) {
                            vr[j] += __shfl_xor(vr[j], sh, 32);
                            vi[j] += __shfl_xor(vi[j], sh, 32);
                        }
                    }
                    {   // through the windows already passed: (M^L)^(32 w), from the table
                        double R[N2][N2];
                        load_mat<S>(R, fa.mats, kMatWw + (dist0 >> 5));
                        affine<S>(sr, sr, R, vr);
                        affine<S>(si, si, R, vi);
                    }
                    if (resolved) {
                        done = true;
                    } else if ((dist0 >> 5) + 1 > kMaxWindows) {
                        *fa.err = 3;
                        done = true;
                    } else {
                        base -= 32;
                        dist0 += 32;
                    }
                } else if (!done) {
                    __builtin_amdgcn_s_sleep(2);
                    if (spin_expired(spins, spin_t0, fa.spin_ticks)) {  // seconds: something is wrong; give up loudly
                        *fa.err = 1;
                        done = true;
                    }
                }
            }
        }
    }
    PH_FSTAMP(4);  // look-back
    if constexpr (GENERAL) {  // P: the tile's true end state, for the tiles after it
        double ml[N2][N2], pr[N2], pi[N2];
        load_mat<S>(ml, fa.mats, kMatML);
        affine<S>(pr, Zr, ml, sr);
        affine<S>(pi, Zi, ml, si);
        publish(1, pr, pi);
    }
    PH_FSTAMP(5);  // publish P

    // ---- 5. the ordered recurrence from the true start states ------------------------------------
    // (lane k0 opens the tile: e = 0 and its table entry is the identity, so it starts from the
    // tile's start state; lanes below k0 hold zeros and produce nothing that is kept)
    double str[N2], sti[N2];
    {
        double pk[N2][N2];  // M^(32 (l5 - k0))
        load_mat<S>(pk, fa.mats, kMatPk + l5);
        affine<S>(str, er, pk, sr);
        affine<S>(sti, ei, pk, si);
    }
    // The Line's last tile: the state after the Line's LAST frame is the cascade's state for the
    // next call.  When the Line ends where a segment ends (frames % 32 == 0: every power-of-two
    // buffer size) that is lane kl's state after the loop below, and it goes to the series' state
    // slot from here.  Otherwise the frame sits in the middle of a segment (step jl of lane kl):
    // the kernel hands out the segment's true start state and chain_tail_kernel (chain_fused.hip)
    // walks the <= 32 frames from there.  No second shape of the loop, nothing inside it.
    const bool ends_here = valid && last_tile && l5 == ((a.HP + len - 1) >> 5);
    const bool on_boundary = ((a.HP + len) & 31) == 0;
    if (ends_here && !on_boundary) {
        double *sp = fa.seg_state + series * NV;
#pragma unroll
        for (int j = 0; j < N2; ++j) {
            sp[j] = str[j];
            sp[N2 + j] = sti[j];
        }
    }
    {
        // channel 0 out of registers, channel 1 out of the plane, the two chains interleaved;
        // results replace the inputs
        double xi[32];
#pragma unroll
        for (int c = 0; c < 32; ++c)
            xi[c] = PH_ROW(c);
#if PH_FUSE_ABLATE == 1 || PH_FUSE_ABLATE == 2 || PH_FUSE_ABLATE == 3
        // ablation builds (scripts/build_ablate_lib.sh chain_fused PH_FUSE_ABLATE fab 1 2 3 4 5 7 8; WRONG results;
        // profiles/r06_chain_fold_ab.txt, r06_chain_ablation.txt): what pass 5 would
        // cost with the numerator and the gain folded into the tap spectrum (1: three operations a sample), the numerator
        // alone (2: four), the gain alone (3: five)
#pragma unroll
        for (int c = 0; c < 32; ++c) {
#if PH_FUSE_ABLATE == 3
            xr[c] = biquad_step<S>(xr[c], str, fc);
            xi[c] = biquad_step<S>(xi[c], sti, fc);
#else
            {
                const double y = xr[c] + str[0];
                str[0] = __builtin_fma(-fc.c[0][3], y, str[1]);
                str[1] = -fc.c[0][4] * y;
                xr[c] = PH_FUSE_ABLATE == 2 ? y * fc.gain : y;
            }
            {
                const double y = xi[c] + sti[0];
                sti[0] = __builtin_fma(-fc.c[0][3], y, sti[1]);
                sti[1] = -fc.c[0][4] * y;
                xi[c] = PH_FUSE_ABLATE == 2 ? y * fc.gain : y;
            }
#endif
        }
#else
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            xr[c] = biquad_step<S>(xr[c], str, fc) * fc.gain;
            xi[c] = biquad_step<S>(xi[c], sti, fc) * fc.gain;
        }
#endif
        if (ends_here && on_boundary) {
            unsigned long long *dst = own_write_slot<NV>(fa.own + series * (2 * 2 * NV), fa.epoch);
#pragma unroll
            for (int j = 0; j < N2; ++j) {
                own_store(dst + 2 * j, fa.epoch, str[j]);
                own_store(dst + 2 * (N2 + j), fa.epoch, sti[j]);
            }
        }
        PH_FSTAMP(6);  // pass 3
        // ---- back to natural layout (channel 1 first: the plane is still its) ------------------
#pragma unroll
        for (int c = 0; c < 32; ++c)
            PH_ROW(c) = xi[c];
        wave_fence();
#pragma unroll
        for (int r = 0; r < 32; ++r)
            PH_NAT(r).im = PH_COL(r);
        wave_fence();
#pragma unroll
        for (int c = 0; c < 32; ++c)
            PH_ROW(c) = xr[c];
        wave_fence();
#pragma unroll
        for (int r = 0; r < 32; ++r)
            PH_NAT(r).re = PH_COL(r);
        wave_fence();
        PH_FSTAMP(7);  // back to natural
    }
}

// ---- cascades of two (or more) sections: one pass of the machinery above PER SECTION ---------------
// A cascade's zero-input transition is 2S x 2S; run as ONE recurrence (fused_epilogue<2>) its scan,
// look-back and pass-5 matrices do not fit a wave's registers (200+ spilled).  But a cascade is its
// sections one after the other: section k's output is section k + 1's input, and each section alone is
// the 2 x 2 problem.  So the tile goes to segment layout ONCE, every section runs zero-state pass, scan,
// look-back and the ordered recurrence over it in place (channel 0 in registers, channel 1 in the
// plane), with its own matrices, records and state slots, and the tile goes back to natural layout
// ONCE.  The arithmetic per section is exactly the one-section epilogue's: same contract
// (tests/test_gpu_chain_fused.py).  Forgetful filters only (every section forgets within a look-back
// window: all audio EQs; the host checks and takes the staged chain otherwise).
//   fa.mats : [S][kMatCount] 2 x 2 matrices, section-major
//   records : [series][tile][A | P][2 NV granules], NV = 4 S: section k's aggregate at doubles
//             2k, 2k + 1 (channel 0) and 2S + 2k, 2S + 2k + 1 (channel 1)
//   slots   : the same order; ring: [S][kLocalRing] LocalRec<4>
__device__ __forceinline__ double biquad_step1(double x, double (&st)[2], const double (&c)[5])
{
    const double y = __builtin_fma(c[0], x, st[0]);
    const double t = __builtin_fma(c[1], x, st[1]);
    st[0] = __builtin_fma(-c[3], y, t);
    const double u = c[2] * x;
    st[1] = __builtin_fma(-c[4], y, u);
    return y;
}

template <int S, bool LOCAL>
__device__ __forceinline__ void fused_epilogue_sections(cd (&lo)[16], cd (&hi)[16], double *pa, double *pb, const Args32 &a,
                                                        const FuseArgs &fa, const FuseConst<S> &fc, int line, int tile, int pair,
                                                        bool valid, int l5_in, int half, LocalRec<4> *ring_all, int gi)
{
    (void)half;
    int l5 = l5_in;
    asm volatile("" : "+v"(l5));
    constexpr int NV = 4 * S;  // doubles per series record / slot
    const int64_t len64 = a.frames - (int64_t)tile * a.L;
    const int len = (int)(len64 < a.L ? len64 : a.L);
    const bool last_tile = tile == a.tiles_per_line - 1;
    valid = valid && len > 0;
    const int64_t series = (int64_t)line * a.pairs + pair;
    unsigned long long *recs = fa.rec + (series * a.tiles_per_line) * (2 * 2 * NV);
    unsigned long long *own_base = fa.own + series * (2 * 2 * NV);

    // ---- to segment layout: channel 0 in registers, channel 1 in the plane's rows ---------------------
    double xr[32];
#pragma unroll
    for (int r = 0; r < 32; ++r)
        PH_COL(r) = PH_NAT(r).re;
    wave_fence();
#pragma unroll
    for (int c = 0; c < 32; ++c)
        xr[c] = PH_ROW(c);
    wave_fence();
#pragma unroll
    for (int r = 0; r < 32; ++r)
        PH_COL(r) = PH_NAT(r).im;
    wave_fence();

    // the state slot of this launch's readers / writer: the tags decide once for all sections
    const bool mine_first = valid && tile - 1 - l5 == -1 && l5 < fc.D;  // this lane stands for "tile -1"
    const bool ends_here = valid && last_tile && l5 == ((a.HP + len - 1) >> 5);
    const bool on_boundary = ((a.HP + len) & 31) == 0;

    for_each_section<S>([&](auto sec_c) {
        constexpr int SEC = decltype(sec_c)::value;
        constexpr bool LAST = SEC == S - 1;
        int l5s = l5;
        asm volatile("" : "+v"(l5s));  // (this section's table addresses are made here, not before the previous section)
        const double *mats = fa.mats + (size_t)SEC * kMatCount * 4;
        const double(&cf)[5] = fc.c[SEC];
        double m0[2][2];
        load_mat<1>(m0, mats, kMatAk + 1);

        // ---- zero-state pass: z = sum_i g_i x_i ---------------------------------------------------------
        double zr[2] = {0.0, 0.0}, zi[2] = {0.0, 0.0};
        {
            double xz[32];
#pragma unroll
            for (int c = 0; c < 32; ++c)
                xz[c] = PH_ROW(c);
            typedef const __attribute__((address_space(4))) double *const_f64;
            const const_f64 gz = (const_f64)(mats + kMatGz * 4);
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const double g0 = gz[2 * c], g1 = gz[2 * c + 1];
                zr[0] = __builtin_fma(g0, xr[c], zr[0]);
                zr[1] = __builtin_fma(g1, xr[c], zr[1]);
                zi[0] = __builtin_fma(g0, xz[c], zi[0]);
                zi[1] = __builtin_fma(g1, xz[c], zi[1]);
            }
        }
        if (l5s < fa.k0)
            zr[0] = zr[1] = zi[0] = zi[1] = 0.0;

        // ---- scan over the 32 segments ---------------------------------------------------------------------
        {
            double mn[2][2], tr[2], ti[2];
#define PH_SCAN_STEP(D_, NEXT_)                                  \
    load_mat<1>(mn, mats, (NEXT_));                              \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                \
    {                                                            \
        tr[j] = dpp_f64<0x110 + (D_), 0xF>(zr[j]);               \
        ti[j] = dpp_f64<0x110 + (D_), 0xF>(zi[j]);               \
    }                                                            \
    affine<1>(zr, zr, m0, tr);                                   \
    affine<1>(zi, zi, m0, ti);                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) m0[i][j] = mn[i][j];
            PH_SCAN_STEP(1, kMatAk + 2)
            PH_SCAN_STEP(2, kMatAk + 4)
            PH_SCAN_STEP(4, kMatAk + 8)
            PH_SCAN_STEP(8, kMatAk + (l5s & 15) + 1)
#undef PH_SCAN_STEP
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                tr[j] = dpp_f64<0x142, 0xA>(zr[j]);
                ti[j] = dpp_f64<0x142, 0xA>(zi[j]);
            }
            affine<1>(zr, zr, m0, tr);
            affine<1>(zi, zi, m0, ti);
        }
        double er[2], ei[2], Zr[2], Zi[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double sr1 = dpp_f64<0x111, 0xF>(zr[j]), si1 = dpp_f64<0x111, 0xF>(zi[j]);
            const double br = dpp_f64<0x142, 0xA>(zr[j]), bi = dpp_f64<0x142, 0xA>(zi[j]);
            er[j] = l5s == 16 ? br : sr1;
            ei[j] = l5s == 16 ? bi : si1;
            Zr[j] = __shfl(zr[j], 31, 32);
            Zi[j] = __shfl(zi[j], 31, 32);
        }

        // ---- publish this section's aggregate ------------------------------------------------------------
        LocalRec<4> *ring = ring_all + SEC * kLocalRing2;
        if (l5s == 31 && valid && !last_tile && !(fa.withhold >= 0 && tile == fa.withhold)) {
            if constexpr (LOCAL) {
                LocalRec<4> *r = ring + gi % kLocalRing2;
                r->v[0] = Zr[0];
                r->v[1] = Zr[1];
                r->v[2] = Zi[0];
                r->v[3] = Zi[1];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __hip_atomic_store(&r->tag, (unsigned long long)(gi / kLocalRing2 + 1), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                unsigned long long *dst = recs + (int64_t)tile * 2 * (2 * NV);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned long long br = __builtin_bit_cast(unsigned long long, Zr[j]);
                    const unsigned long long bi = __builtin_bit_cast(unsigned long long, Zi[j]);
                    granule_store(dst + 2 * (2 * SEC + j), fa.epoch, (unsigned)br);
                    granule_store(dst + 2 * (2 * SEC + j) + 1, fa.epoch, (unsigned)(br >> 32));
                    granule_store(dst + 2 * (2 * S + 2 * SEC + j), fa.epoch, (unsigned)bi);
                    granule_store(dst + 2 * (2 * S + 2 * SEC + j) + 1, fa.epoch, (unsigned)(bi >> 32));
                }
            }
        }

        // ---- the tile's true start state for this section: s = sum_{j < D} (M^L)^j A_{t-1-j} -----------------
        double tj0[2][2];
        load_mat<1>(tj0, mats, kMatTj + l5s);
        double wr[2] = {0.0, 0.0}, wi[2] = {0.0, 0.0};
        if (mine_first) {  // the cascade's state of the previous launch: this section's part of the newer slot
            const unsigned long long *r = own_base + own_newer_slot<NV>(own_base, fa.epoch) * (2 * NV);
            double pay[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = (j < 2 ? 2 * SEC + j : 2 * S + 2 * SEC + (j - 2));
                const unsigned long long g0 = granule_load(r + 2 * d), g1 = granule_load(r + 2 * d + 1);
                pay[j] = __builtin_bit_cast(double, (g1 << 32) | (g0 & 0xFFFFFFFFull));
            }
            wr[0] = pay[0];
            wr[1] = pay[1];
            wi[0] = pay[2];
            wi[1] = pay[3];
        }
        {
            const int u = tile - 1 - l5s;
            const bool need = valid && l5s < fc.D && u >= -1;
            bool ready = !(need && u >= 0);
            unsigned spins = 0;
            unsigned long long spin_t0 = 0;
            if constexpr (LOCAL) {
                const int gp0 = gi - (l5s + 1) * a.pairs;
                const int gp = gp0 < 0 ? 0 : gp0;  // (lanes without a predecessor do not look)
                const LocalRec<4> *r = ring + gp % kLocalRing2;
                const unsigned long long want = (unsigned long long)(gp / kLocalRing2 + 1);
                while (!__all(ready)) {
                    if (!ready) {
                        if (__hip_atomic_load(&r->tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == want) {
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                            wr[0] = r->v[0];
                            wr[1] = r->v[1];
                            wi[0] = r->v[2];
                            wi[1] = r->v[3];
                            ready = true;
                        } else {
                            __builtin_amdgcn_s_sleep(1);
                            if (spin_expired(spins, spin_t0, fa.spin_ticks)) {
                                *fa.err = 1;
                                ready = true;
                            }
                        }
                    }
                }
            } else {
                while (!__all(ready)) {
                    if (!ready) {
                        const unsigned long long *r = recs + (int64_t)u * 2 * (2 * NV);
                        double pay[4];
                        bool ok = true;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int d = (j < 2 ? 2 * SEC + j : 2 * S + 2 * SEC + (j - 2));
                            const unsigned long long g0 = granule_load(r + 2 * d), g1 = granule_load(r + 2 * d + 1);
                            ok = ok && (unsigned)(g0 >> 32) == fa.epoch && (unsigned)(g1 >> 32) == fa.epoch;
                            pay[j] = __builtin_bit_cast(double, (g1 << 32) | (g0 & 0xFFFFFFFFull));
                        }
                        if (ok) {
                            wr[0] = pay[0];
                            wr[1] = pay[1];
                            wi[0] = pay[2];
                            wi[1] = pay[3];
                            ready = true;
                        } else {
                            __builtin_amdgcn_s_sleep(1);
                            if (spin_expired(spins, spin_t0, fa.spin_ticks)) {
                                *fa.err = 1;
                                ready = true;
                            }
                        }
                    }
                }
            }
        }
        double sr[2] = {0.0, 0.0}, si[2] = {0.0, 0.0};
        {
            double zero[2] = {0.0, 0.0}, vr[2], vi[2];
            affine<1>(vr, zero, tj0, wr);
            affine<1>(vi, zero, tj0, wi);
            const int nd = fc.D < 32 ? fc.D : 32;
            for (int d = 0; d < nd; ++d) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    sr[j] += __shfl(vr[j], d, 32);
                    si[j] += __shfl(vi[j], d, 32);
                }
            }
        }

        // ---- the ordered recurrence from the true start states, in place ----------------------------------
        double str[2], sti[2];
        {
            double pk[2][2];
            load_mat<1>(pk, mats, kMatPk + l5s);
            affine<1>(str, er, pk, sr);
            affine<1>(sti, ei, pk, si);
        }
        if (ends_here && !on_boundary) {  // the Line ends inside this segment: chain_tail_kernel walks it from here
            double *sp = fa.seg_state + series * NV;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                sp[2 * SEC + j] = str[j];
                sp[2 * S + 2 * SEC + j] = sti[j];
            }
        }
        {
            double xi[32];
#pragma unroll
            for (int c = 0; c < 32; ++c)
                xi[c] = PH_ROW(c);
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                if constexpr (LAST) {
                    xr[c] = biquad_step1(xr[c], str, cf) * fc.gain;
                    xi[c] = biquad_step1(xi[c], sti, cf) * fc.gain;
                } else {
                    xr[c] = biquad_step1(xr[c], str, cf);
                    xi[c] = biquad_step1(xi[c], sti, cf);
                }
            }
            if (ends_here && on_boundary) {  // the Line ends where this segment ends: its state after the loop
                unsigned long long *dst = own_write_slot<NV>(own_base, fa.epoch);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    own_store(dst + 2 * (2 * SEC + j), fa.epoch, str[j]);
                    own_store(dst + 2 * (2 * S + 2 * SEC + j), fa.epoch, sti[j]);
                }
            }
            // channel 1 back into the plane's rows: the next section's input, or on its way out
#pragma unroll
            for (int c = 0; c < 32; ++c)
                PH_ROW(c) = xi[c];
            wave_fence();
        }
    });

    // ---- back to natural layout ------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < 32; ++r)
        PH_NAT(r).im = PH_COL(r);
    wave_fence();
#pragma unroll
    for (int c = 0; c < 32; ++c)
        PH_ROW(c) = xr[c];
    wave_fence();
#pragma unroll
    for (int r = 0; r < 32; ++r)
        PH_NAT(r).re = PH_COL(r);
    wave_fence();
}

// element type of the FIR history a kernel reads and writes
template <typename TIn, int S>
struct HistOf {
    using type = double;
};
template <int S>
struct HistOf<float, S> {
    using type = typename std::conditional<(S > 0), float, double>::type;
};

// S = 0: the FIR alone.  S = 1, 2: the FIR's tile goes through an S-section biquad cascade and a
// gain before it is stored (chain_fused.hip; fa / fc are then the epilogue's arguments).
// MONO (S = 0, one channel): a half-wave's complex sequence carries TWO TILES of the one channel -- tile t as
// the real part, tile t + tiles_per_line as the imaginary part, `mono_shift` frames later in the same Line --
// instead of a channel and its neighbour: the Line is run as if it had two channels, the second being its own
// second half.  tiles_per_line is then HALF the Line's tiles.
template <typename TIn, typename TOut, int S = 0, bool GENERAL = false, bool LOCAL = false, bool MONO = false>
__global__ void __launch_bounds__(kWaves32 * 64)
fir_ols32_kernel(const TIn *__restrict__ in_base, TOut *__restrict__ out_base,
                 const typename HistOf<TIn, S>::type *__restrict__ hist_base,
                 const double2 *__restrict__ tw_g, const double2 *__restrict__ hperm_g, const Args32 a,
                 const FuseArgs fa, const FuseConst<(S > 0 ? S : 1)> fc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // (S >= 2: one ring of kLocalRing2 records per section -- the LDS has 2000 bytes to spare, two rings of
    // 64 need 2560 more.  The tap spectrum read from global memory instead would free 8 KB, and costs the
    // transform 35 registers: 95 spilled.)
    constexpr bool kSpecInLds = true;
    double2 *hspec_lds = reinterpret_cast<double2 *>(smem_raw);       // H[0..512] (+ pad)
    double2 *tws = kSpecInLds ? hspec_lds + kHalf32 + 1 : hspec_lds;  // W1024^(k n), k = 1..31, n = 0..31
    const double2 *hspec = kSpecInLds ? hspec_lds : hperm_g;
    double *planes = reinterpret_cast<double *>(tws + 31 * 32);   // [waves][2][kPlane32]
    constexpr int kRingNV = S >= 2 ? 4 : 4 * (S > 0 ? S : 1);      // doubles per record (S >= 2: one ring per section)
    constexpr int kRings = S >= 2 ? S : 1;
    constexpr int kRingLen = S >= 2 ? kLocalRing2 : kLocalRing;
    LocalRec<kRingNV> *ring = reinterpret_cast<LocalRec<kRingNV> *>(planes + (size_t)kWaves32 * 2 * kPlane32);  // LOCAL only
    unsigned *round_done = reinterpret_cast<unsigned *>(ring + kRings * kRingLen);  // [4], LOCAL only
    if constexpr (LOCAL) {
        for (int i = threadIdx.x; i < kRings * kRingLen; i += kWaves32 * 64)
            ring[i].tag = 0;
        if (threadIdx.x < 4)
            round_done[threadIdx.x] = 0;
    }

    fir_history_carry(in_base, hist_base, static_cast<typename HistOf<TIn, S>::type *>(a.hist_new), a.frames, a.line_stride,
                      a.H, a.C, a.lines);
    if constexpr (kSpecInLds)
        for (int i = threadIdx.x; i < kHalf32; i += kWaves32 * 64)
            hspec_lds[i] = hperm_g[i];
    for (int i = threadIdx.x; i < 31 * 32; i += kWaves32 * 64)
        tws[i] = tw_g[32 + i];
    __syncthreads();

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, l5 = lane & 31;
    if constexpr (S > 0) {
        // The two waves of a SIMD (w and w + 4) out of step by part of a unit: one wave's epilogue --
        // chains of dependent operations and round trips -- then runs under the other's dense transform
        // instead of beside its epilogue.
        if (a.stagger > 0 && __builtin_amdgcn_readfirstlane(wave) >= kWaves32 / 2)
            for (int i = 0; i < a.stagger; ++i)
                __builtin_amdgcn_s_sleep(127);
    }
    double *plane = planes + (wave * 2 + half) * kPlane32;
    double *pa = plane + l5;       // (row r, this lane's column): pa[33 r]
    double *pb = plane + 33 * l5;  // (this lane's row, column c): pb[c]
    const double2 *__restrict__ twl = tws + l5 - 32;  // row k at twl[32 k]
    const double2 *__restrict__ hlo = hspec + l5;     // H[32 k1 + l5]
    const double2 *__restrict__ hhi = hspec - l5;     // conj side: H[1024 - 32 k1 - l5]

    using In2 = typename Pair<TIn>::type;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int nb = (int)gridDim.x;
    // consecutive units go to consecutive blocks of the same XCD (block b runs on XCD b % 8):
    // neighbouring tiles share their overlap through that XCD's L2
    const int xb = nb % 8 == 0 ? ((int)blockIdx.x % 8) * (nb / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
    // LOCAL: this workgroup runs Lines blockIdx, blockIdx + nb, ... whole; its units are those Lines'
    // units one after the other, eight (one per wave) to a round
    const int my_lines = LOCAL ? (a.lines - (int)blockIdx.x + nb - 1) / nb : 0;
    const int64_t my_units = (int64_t)my_lines * a.upl;
    const int64_t wave_global = LOCAL ? (int64_t)wave_u : (int64_t)wave_u * nb + xb;
    const int64_t wave_stride = LOCAL ? (int64_t)kWaves32 : (int64_t)nb * kWaves32;
    const int64_t unit_end = LOCAL ? (my_units + kWaves32 - 1) / kWaves32 * kWaves32 : a.nunits;  // LOCAL: whole rounds
    int line = 0, slot = 0;
    if (LOCAL) {
        line = (int)blockIdx.x + __builtin_amdgcn_readfirstlane((int)(wave_global / a.upl)) * nb;
        slot = __builtin_amdgcn_readfirstlane((int)(wave_global % a.upl));
    } else if (wave_global < a.nunits) {
        line = __builtin_amdgcn_readfirstlane((int)(wave_global / a.upl));
        slot = __builtin_amdgcn_readfirstlane((int)(wave_global % a.upl));
    }
    const unsigned in_step = (unsigned)(32 * a.C * sizeof(TIn));    // 32 frames
    const unsigned out_step = (unsigned)(32 * a.C * sizeof(TOut));
    auto bytes31 = [](int64_t n) { return (int)(n < 0x7FFFFFFF ? n : 0x7FFFFFFF); };
    [[maybe_unused]] auto bytes31c = [](int64_t n) { return (int)(n < 0 ? 0 : (n < 0x7FFFFFFF ? n : 0x7FFFFFFF)); };

#ifdef PH_FUSE_PROF
    unsigned long long fprof_acc[kFuseProfPhases] = {};
    unsigned long long fprof_last = __builtin_amdgcn_s_memtime();
#endif
    int round = 0;
    auto end_round = [&]() {
        if constexpr (LOCAL) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0)
                __hip_atomic_fetch_add(&round_done[round & 3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ++round;
        }
    };
#ifdef PH_FUSE_PROF
    int tl_unit = -1;
#endif
    for (int64_t unit = wave_global; unit < unit_end; unit += wave_stride) {
#ifdef PH_FUSE_PROF
        ++tl_unit;
        if constexpr (S > 0)
            PH_FTIME(0);
#endif
        if constexpr (LOCAL) {
            if (round >= 2) {  // everybody is through round - 2 (its count: 8 per pass over the four counters)
                const unsigned want = (unsigned)kWaves32 * (unsigned)((round - 2) / 4 + 1);
                while (__hip_atomic_load(&round_done[(round - 2) & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want)
                    __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            if (unit >= my_units) {  // a wave without a unit in the workgroup's last round
                end_round();
                continue;
            }
        }
#ifdef PH_FUSE_PROF
        if constexpr (S > 0)
            PH_FTIME(1);
#endif
        // ---- the unit's two items: item0 = 2 slot (half 0), item0 + 1 (half 1) -----------------
        const int item0 = 2 * slot;
        const int tile0 = __builtin_amdgcn_readfirstlane(item0 / a.pairs);
        const int pair0 = __builtin_amdgcn_readfirstlane(item0 - tile0 * a.pairs);
        int tile = tile0, pair = pair0 + half;
        if (pair >= a.pairs) {
            pair = 0;
            tile = tile0 + 1;
        }
        const bool valid = item0 + half < a.ipl;
        const int c0 = 2 * pair;
        const int64_t fr00 = (int64_t)tile0 * a.L - a.HP;  // first window frame of half 0's item

        cd lo[16], hi[16];
        // ---- the window: lane l5, register r -> window index l5 + 32 r ---------------------------
        if (tile0 > 0) {
            // both windows start inside the Line: one buffer resource based at half 0's window,
            // 32-bit lane offsets; frames past the end of the Line read as zero
            const TIn *base = in_base + (int64_t)line * a.line_stride + fr00 * a.C;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<TIn *>(base), 0, bytes31((a.frames - fr00) * a.C * (int64_t)sizeof(TIn)), 0x00020000);
#if PH_FUSE_ABLATE == 8  // (8: no window is fetched -- every load lands beyond the buffer and reads zero)
            const unsigned v0 = kOut32;
#else
            const unsigned v0 = valid ? (unsigned)((((tile - tile0) * a.L + l5) * a.C + c0) * (int)sizeof(TIn)) : kOut32;
#endif
            In2 pf[32];
            if constexpr (MONO) {
                const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<TIn *>(base + a.mono_shift), 0, bytes31c((a.frames - fr00 - a.mono_shift) * (int64_t)sizeof(TIn)), 0x00020000);
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    pf[r].x = buf_load_one<TIn>(rs, v0 + (unsigned)r * in_step);
                    pf[r].y = buf_load_one<TIn>(rb, v0 + (unsigned)r * in_step);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 32; ++r)
                    pf[r] = buf_load_pair<TIn>(rs, v0 + (unsigned)r * in_step);
            }
#pragma unroll
            for (int r = 0; r < 32; ++r)
                PH_NAT(r) = cd{(double)pf[r].x, (double)pf[r].y};
        } else {
            // a Line's first tile: its head is the history.  Branch-free, two batches of requests (a
            // window's first 512 indices may reach back into the history, the rest never do): frames
            // outside the Line and the history read as zero through the resources' bounds.  (One load
            // per branch of an if / else ladder was 32 dependent round trips: the first tile of a Line
            // heads the fused chain's look-back order, and everybody behind it waited.)
            using THist = typename HistOf<TIn, S>::type;
            using H2 = typename Pair<THist>::type;
            const TIn *in0 = in_base + (int64_t)line * a.line_stride;
            const THist *hist0 = hist_base + (int64_t)line * a.H * a.C;
            const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<TIn *>(in0), 0, bytes31(a.frames * a.C * (int64_t)sizeof(TIn)), 0x00020000);
            const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<THist *>(hist0), 0, bytes31((int64_t)a.H * a.C * (int64_t)sizeof(THist)), 0x00020000);
            const int fr0 = tile * a.L - a.HP;  // (first tiles: small)
            if constexpr (MONO) {
                // the real part: history then input; the imaginary part: the Line `mono_shift` frames on (always input)
                const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<TIn *>(in0 + a.mono_shift - a.HP), 0, bytes31c((a.frames - a.mono_shift + a.HP) * (int64_t)sizeof(TIn)), 0x00020000);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    TIn pa[16], pb[16];
                    THist pc[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int g = fr0 + l5 + 32 * (16 * b + k);
                        pa[k] = buf_load_one<TIn>(rin, valid && g >= 0 ? (unsigned)(g * (int)sizeof(TIn)) : kOut32);
                        pb[k] = buf_load_one<TIn>(rb, valid ? (unsigned)((g + a.HP) * (int)sizeof(TIn)) : kOut32);
                        if (b == 0)
                            pc[k] = buf_load_one<THist>(rh, valid && g < 0 && g >= -a.H ? (unsigned)((g + a.H) * (int)sizeof(THist)) : kOut32);
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int g = fr0 + l5 + 32 * (16 * b + k);
                        PH_NAT(16 * b + k) = cd{b == 0 && g < 0 ? (double)pc[k] : (double)pa[k], (double)pb[k]};
                    }
                }
            } else {
            {
                In2 pi[16];
                H2 ph[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int g = fr0 + l5 + 32 * r;
                    pi[r] = buf_load_pair<TIn>(rin, valid && g >= 0 ? (unsigned)((g * a.C + c0) * (int)sizeof(TIn)) : kOut32);
                    ph[r] = buf_load_pair<THist>(rh, valid && g < 0 && g >= -a.H ? (unsigned)(((g + a.H) * a.C + c0) * (int)sizeof(THist)) : kOut32);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int g = fr0 + l5 + 32 * r;
                    PH_NAT(r) = g >= 0 ? cd{(double)pi[r].x, (double)pi[r].y} : cd{(double)ph[r].x, (double)ph[r].y};
                }
            }
            {
                In2 pi[16];
#pragma unroll
                for (int r = 16; r < 32; ++r) {
                    const int g = fr0 + l5 + 32 * r;  // >= 0: HP <= 511
                    pi[r - 16] = buf_load_pair<TIn>(rin, valid ? (unsigned)((g * a.C + c0) * (int)sizeof(TIn)) : kOut32);
                }
#pragma unroll
                for (int r = 16; r < 32; ++r)
                    PH_NAT(r) = cd{(double)pi[r - 16].x, (double)pi[r - 16].y};
            }
            }
        }
        // the next unit's coordinates (uniform)
        const int cur_line = line;
        if constexpr (LOCAL) {
            const int64_t nu = unit + wave_stride;
            line = (int)blockIdx.x + __builtin_amdgcn_readfirstlane((int)(nu / a.upl)) * nb;
            slot = __builtin_amdgcn_readfirstlane((int)(nu % a.upl));
        } else {
            slot += a.d_slot;
            if (slot >= a.upl) {
                slot -= a.upl;
                ++line;
            }
            line += a.d_line;
        }

        ols32_transform(lo, hi, pa, pb, twl, hlo, hhi);
        if constexpr (S > 0) {
            PH_FSTAMP(0);  // window + FIR transform
            PH_FTIME(2);
            // The epilogue is chains of dependent fma and round trips, a few instructions each: they
            // go ahead of the other wave's dense transform on this SIMD, which loses nothing by it.
            __builtin_amdgcn_s_setprio(3);
            __builtin_amdgcn_sched_barrier(0);  // the epilogue's early loads stay out of the transform's registers
            // (LOCAL: item index in the workgroup's list = two per unit, Lines padded to whole units)
#if PH_FUSE_ABLATE != 4  // (4: the transform and the stores alone -- what the epilogue costs in all)
            if constexpr (S >= 2)
                fused_epilogue_sections<S, LOCAL>(lo, hi, pa, pb, a, fa, fc, cur_line, tile, c0 >> 1, valid, l5, half, ring,
                                                  (int)(2 * unit) + half);
            else
                fused_epilogue<S, GENERAL, LOCAL>(lo, hi, pa, pb, a, fa, fc, cur_line, tile, c0 >> 1, valid, l5, half, ring,
                                                  (int)(2 * unit) + half PH_FPROF_ARGS);
#endif
            __builtin_amdgcn_sched_barrier(0);  // ... and the store addresses are not computed ahead of it
            __builtin_amdgcn_s_setprio(0);
            PH_FTIME(3);
        }

        // ---- store the valid part: window index i >= H is frame t0 + i - H --------------------
        {
            const int64_t t00 = (int64_t)tile0 * a.L;
            TOut *base = out_base + (int64_t)cur_line * a.line_stride + t00 * a.C;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                base, 0, bytes31((a.frames - t00) * a.C * (int64_t)sizeof(TOut)), 0x00020000);
            const int o0 = (((tile - tile0) * a.L + l5 - a.HP) * a.C + c0) * (int)sizeof(TOut);
            const int i0 = valid ? l5 - a.HP : -2048;  // window index - HP of register 0: outputs need >= 0
            const bool lone = a.odd && c0 + 1 == a.C;  // (uniform over a half-wave)
            if constexpr (MONO) {
                const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
                    base + a.mono_shift, 0, bytes31c((a.frames - t00 - a.mono_shift) * (int64_t)sizeof(TOut)), 0x00020000);
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const unsigned voff = i0 + 32 * r >= 0 ? (unsigned)(o0 + r * (int)out_step) : kOut32;
                    buf_store_one<TOut>(rs, voff, PH_NAT(r).re);
                    buf_store_one<TOut>(rb, voff, PH_NAT(r).im);
                }
            } else if (lone) {
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const int off = o0 + r * (int)out_step;
                    buf_store_one<TOut>(rs, i0 + 32 * r >= 0 ? (unsigned)off : kOut32, PH_NAT(r).re);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const int off = o0 + r * (int)out_step;
#if PH_FUSE_ABLATE == 7  // (7: nothing is stored -- every store lands beyond the buffer)
                    buf_store_pair<TOut>(rs, kOut32 | (unsigned)(off & 0xFFFF), PH_NAT(r).re, PH_NAT(r).im);
#else
                    buf_store_pair<TOut>(rs, i0 + 32 * r >= 0 ? (unsigned)off : kOut32, PH_NAT(r).re, PH_NAT(r).im);
#endif
                }
            }
        }
#ifdef PH_FUSE_PROF
        if constexpr (S > 0) {
            PH_FSTAMP(9);  // stores
            PH_FTIME(4);
        }
#endif
        end_round();
    }
#ifdef PH_FUSE_PROF
    if constexpr (S > 0) {
        if (fa.prof && lane == 0) {
            unsigned long long *dst = fa.prof + ((size_t)blockIdx.x * kWaves32 + wave) * kFuseProfPhases;
            for (int i = 0; i < kFuseProfPhases; ++i)
                dst[i] = fprof_acc[i];
        }
    }
#endif
}


}  // namespace ols
}  // namespace pipehip
