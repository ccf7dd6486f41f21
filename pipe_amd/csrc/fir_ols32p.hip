// Partitioned overlap-save: FIRs of 513 .. 4096 taps on the 32 x 32 transform of fir_ols32.hip.
//
// The taps are cut into P partitions of Np <= 512 taps, h = sum_p h_p delayed by p Np frames, so
//     y[n] = sum_p (h_p * x)[n - p Np].
// For an output tile, partition p's contribution is the circular convolution of h_p with the 1024-frame
// window that starts p Np frames EARLIER than partition 0's -- and the valid outputs of every partition
// sit at the SAME window indices.  Their spectra therefore add: Y = sum_p X_p H_p, ONE inverse
// transform per tile, no float64 intermediate through HBM.  Two kernels:
//
//   fir_ols32d_kernel (shipped): hop = partition = 512 frames.  Partition p's window of tile t is then
//     partition 0's window of tile t - p, so a half-wave that runs consecutive tiles of one series needs
//     ONE forward transform per tile and reads the other P - 1 spectra from a ring it wrote itself:
//     two transforms per tile for any P.
//   fir_ols32p_kernel (PIPE_HIP_FIR_PARTITION_SUM, A/B): partitions of ceil(N / P) taps, tiles of
//     1025 - Np outputs dealt like the one-spectrum kernel's, P forward transforms per tile, the running
//     sum parked in a per-wave scratch area between them (a transform needs all 128 of a lane's data
//     registers).
//
// Ring and scratch are device memory, [slot][register][lane] so that the accesses coalesce: L2 /
// Infinity-Cache traffic.  Same contract as the one-spectrum form: float64 arithmetic, the float32
// result within one ulp of the oracle's ordered sum at the filter's full scale
// (tests/test_gpu_fir_ols.py).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_ext.h>

#include "common.hpp"
#include "fir_hist.hpp"
#include "fir_ols.hpp"
#include "fir_ols_impl.hpp"
#include "ols32_kernel.hpp"

namespace pipehip {
namespace ols {
namespace {

struct ArgsP {
    Args32 a;            // geometry with H = N - 1 (the whole history), HP = Np - 1, L = 1025 - Np
    int P, Np;
    const double2 *hpart;  // [P][kHalf32 + 1] tap spectra of the partitions (scaled by 1/1024)
    double2 *scratch;      // [waves][32 registers][64 lanes]
};

template <typename TIn, typename TOut>
__global__ void __launch_bounds__(kWaves32 * 64)
fir_ols32p_kernel(const TIn *__restrict__ in_base, TOut *__restrict__ out_base, const double *__restrict__ hist_base,
                  const double2 *__restrict__ tw_g, const ArgsP t)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2 *tws = reinterpret_cast<double2 *>(smem_raw);         // W1024^(k n), k = 1..31, n = 0..31
    double *planes = reinterpret_cast<double *>(tws + 31 * 32);   // [waves][2][kPlane32]
    const Args32 &a = t.a;

    fir_history_carry(in_base, hist_base, static_cast<double *>(a.hist_new), a.frames, a.line_stride, a.H, a.C, a.lines);
    for (int i = threadIdx.x; i < 31 * 32; i += kWaves32 * 64)
        tws[i] = tw_g[32 + i];
    __syncthreads();

    const int wave = threadIdx.x >> 6;
    const int lane_ = threadIdx.x & 63;
    const int l5_ = lane_ & 31;

    using In2 = typename Pair<TIn>::type;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int nb = (int)gridDim.x;
    const int xb = nb % 8 == 0 ? ((int)blockIdx.x % 8) * (nb / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
    const int64_t wave_global = (int64_t)wave_u * nb + xb;
    const int64_t wave_stride = (int64_t)nb * kWaves32;
    double2 *wave_scratch = t.scratch + ((int64_t)blockIdx.x * kWaves32 + wave_u) * (32 * 64);  // [r][lane]
    int line = 0, slot = 0;
    if (wave_global < a.nunits) {
        line = __builtin_amdgcn_readfirstlane((int)(wave_global / a.upl));
        slot = __builtin_amdgcn_readfirstlane((int)(wave_global % a.upl));
    }
    const unsigned in_step = (unsigned)(32 * a.C * sizeof(TIn));
    const unsigned out_step = (unsigned)(32 * a.C * sizeof(TOut));
    auto bytes31 = [](int64_t n) { return (int)(n < 0x7FFFFFFF ? n : 0x7FFFFFFF); };
    const int64_t last = a.frames - 1;

    for (int64_t unit = wave_global; unit < a.nunits; unit += wave_stride) {
        const int item0 = 2 * slot;
        const int tile0 = __builtin_amdgcn_readfirstlane(item0 / a.pairs);
        const int pair0 = __builtin_amdgcn_readfirstlane(item0 - tile0 * a.pairs);

        cd lo[16], hi[16];
        // one partition: FIRST starts the tile's sum, LAST leaves it in the registers for the inverse
        auto partition = [&](int p, auto first_c, auto last_c) {
            constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
            // Every per-lane quantity of the loop body is derived again from opaque copies of the lane
            // indices: whatever is invariant over the partitions would otherwise be computed once per
            // unit and held in registers through all the transforms, which have none to give (the
            // library ships no kernel that spills: scripts/check_spills.sh).
            int l5 = l5_, lane = lane_;
            asm volatile("" : "+v"(l5), "+v"(lane));
            const int half = lane >> 5;
            double *plane = planes + (wave_u * 2 + half) * kPlane32;
            double *pa = plane + l5;
            double *pb = plane + 33 * l5;
            const double2 *__restrict__ twl = tws + l5 - 32;
            int tile = tile0, pair = pair0 + half;
            if (pair >= a.pairs) {
                pair = 0;
                tile = tile0 + 1;
            }
            const bool valid = item0 + half < a.ipl;
            const int c0 = 2 * pair;
            // ---- partition p's window: it starts p Np frames before partition 0's ------------------
            const int64_t fr00 = (int64_t)tile0 * a.L - a.HP - (int64_t)p * t.Np;  // half 0's first window frame
            if (fr00 >= 0) {
                const TIn *base = in_base + (int64_t)line * a.line_stride + fr00 * a.C;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<TIn *>(base), 0, bytes31((a.frames - fr00) * a.C * (int64_t)sizeof(TIn)), 0x00020000);
                unsigned v0 = valid ? (unsigned)((((tile - tile0) * a.L + l5) * a.C + c0) * (int)sizeof(TIn)) : kOut32;
                asm volatile("" : "+v"(v0));  // (the 32 lane offsets are made per partition, not kept across the transforms)
                In2 pf[32];
#pragma unroll
                for (int r = 0; r < 32; ++r)
                    pf[r] = buf_load_pair<TIn>(rs, v0 + (unsigned)r * in_step);
#pragma unroll
                for (int r = 0; r < 32; ++r)
                    PH_NAT(r) = cd{(double)pf[r].x, (double)pf[r].y};
            } else {
                // the window reaches back into the history (N - 1 frames deep)
                const TIn *__restrict__ in = in_base + (int64_t)line * a.line_stride;
                const double *__restrict__ hist = hist_base + (int64_t)line * a.H * a.C;
                const int64_t fr0 = (int64_t)tile * a.L - a.HP - (int64_t)p * t.Np;
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const int64_t g = fr0 + l5 + 32 * r;
                    double re = 0.0, im = 0.0;
                    if (valid) {
                        if (g >= 0) {
                            if (g <= last) {
                                re = (double)in[g * a.C + c0];
                                im = (double)in[g * a.C + c0 + 1];
                            }
                        } else if (g >= -(int64_t)a.H) {
                            re = hist[(g + a.H) * a.C + c0];
                            im = hist[(g + a.H) * a.C + c0 + 1];
                        }
                    }
                    PH_NAT(r) = cd{re, im};
                }
            }
            ols32_forward(lo, hi, pa, pb, twl);
            __builtin_amdgcn_sched_barrier(0);  // the product's loads and addresses stay out of the transform's registers
            // ---- times H_p: lane k2 = l5, register k1 (split layout) holds X[32 k1 + l5]; the upper
            //      half of the spectrum is the conjugate mirror (real taps) ----------------------------
            // Four registers at a time, fenced: all 32 spectrum entries (and 32 pieces of the running
            // sum) requested at once would sit in 256 registers next to the tile's 128.
            // (The per-lane addresses of the spectrum and of the scratch area are invariant over both
            // loops; hoisted out they would sit in ~130 registers through the transform, which has
            // none to give.  Opaque copies of the lane indices keep them inside.)
            const int l5o = l5;
            const double2 *__restrict__ hp = t.hpart + (size_t)p * (kHalf32 + 1);
            double2 *__restrict__ mine = wave_scratch + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                double2 h[4], sum[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k1 = 4 * g + i;
                    h[i] = k1 < 16 ? hp[32 * k1 + l5o] : hp[1024 - 32 * k1 - l5o];
                }
                // (FIRST / LAST are compile-time: a run-time branch around these loads made the compiler keep
                // two versions of the loop body's registers -- 51 spilled)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    sum[i] = FIRST ? double2{0.0, 0.0} : mine[(4 * g + i) * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k1 = 4 * g + i;
                    const cd w{h[i].x, k1 < 16 ? h[i].y : -h[i].y};
                    cd v = cmul(PH_SPL(k1), w);
                    v.re += sum[i].x;
                    v.im += sum[i].y;
                    if constexpr (LAST)
                        PH_SPL(k1) = v;
                    else
                        mine[k1 * 64] = double2{v.re, v.im};
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        partition(0, std::true_type{}, std::false_type{});
#pragma unroll 1
        for (int p = 1; p + 1 < t.P; ++p)
            partition(p, std::false_type{}, std::false_type{});
        partition(t.P - 1, std::false_type{}, std::true_type{});
        // the next unit's coordinates (uniform)
        const int cur_line = line;
        slot += a.d_slot;
        if (slot >= a.upl) {
            slot -= a.upl;
            ++line;
        }
        line += a.d_line;

        int l5 = l5_, lane = lane_;
        asm volatile("" : "+v"(l5), "+v"(lane));
        const int half = lane >> 5;
        double *plane = planes + (wave_u * 2 + half) * kPlane32;
        double *pa = plane + l5;
        double *pb = plane + 33 * l5;
        const double2 *__restrict__ twl = tws + l5 - 32;
        int tile = tile0, pair = pair0 + half;
        if (pair >= a.pairs) {
            pair = 0;
            tile = tile0 + 1;
        }
        const bool valid = item0 + half < a.ipl;
        const int c0 = 2 * pair;
        ols32_inverse_plain(lo, hi, pa, pb, twl);

        // ---- store the valid part: window index i >= HP is frame tile L + i - HP ---------------------
        {
            const int64_t t00 = (int64_t)tile0 * a.L;
            TOut *base = out_base + (int64_t)cur_line * a.line_stride + t00 * a.C;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                base, 0, bytes31((a.frames - t00) * a.C * (int64_t)sizeof(TOut)), 0x00020000);
            const int o0 = (((tile - tile0) * a.L + l5 - a.HP) * a.C + c0) * (int)sizeof(TOut);
            const int i0 = valid ? l5 - a.HP : -2048;
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int off = o0 + r * (int)out_step;
                buf_store_pair<TOut>(rs, i0 + 32 * r >= 0 ? (unsigned)off : kOut32, PH_NAT(r).re, PH_NAT(r).im);
            }
        }
    }
}

// ---- frequency-domain delay line: two transforms per tile for ANY number of partitions ---------------
// With hop = partition = 512 frames (taps zero-padded to P x 512) partition p's window of tile t IS
// partition 0's window of tile t - p: X_{t,p} = X_{t-p}.  A half-wave that runs R consecutive tiles of
// one series therefore needs ONE forward transform per tile, keeps the last P spectra in a ring
// ([slot][register][lane] in device memory, L2 / Infinity-Cache traffic) and forms
//     Y_t = sum_p X_{t-p} H_p                       then ONE inverse transform.
// A run starts with P - 1 forward transforms of the windows before its first tile (history / earlier
// input), so runs are long compared with P (the host picks R).  Both halves of a wave run consecutive
// runs of the same series: one buffer resource serves both windows.
struct ArgsD {
    Args32 a;             // HP = L = 512; tiles_per_line = ceil(frames / 512); H = N - 1 (the whole history)
    int P, R;             // partitions; tiles per run (a multiple of P)
    int upl;              // units (pairs of runs) per (Line, pair) series
    int64_t nunits;
    const double2 *hpart; // [P][kHalf32 + 1]
    double2 *ring;        // [waves][P][32 registers][64 lanes]
    unsigned long long *prof;  // PH_OLSD_PROF builds: [wave][kProfD] s_memtime ticks per phase
};
[[maybe_unused]] constexpr int kProfD = 6;
// measurement builds (scripts/build_ablate_lib.sh fir_ols32p PH_OLSD_ABLATE ...): bit 0 no output stores,
// bit 1 no ring stores, bit 2 ring loads replaced by arithmetic, bit 3 tap-spectrum loads likewise,
// bit 4 no window loads.  Wrong results by construction.
#ifndef PH_OLSD_ABLATE
#define PH_OLSD_ABLATE 0
#endif
#ifdef PH_OLSD_PROF
#define PH_D_STAMP(i)                                                 \
    do {                                                              \
        const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime(); \
        dprof[i] += now_ - dlast;                                     \
        dlast = now_;                                                 \
    } while (0)
#else
#define PH_D_STAMP(i) \
    do {              \
    } while (0)
#endif

// MONO (one channel): the Line is run as if it had two channels, the second being its own second half,
// a.mono_shift frames on (ols32_kernel.hpp: the same device in the one-spectrum kernel); a.tiles_per_line is
// then half the Line's tiles, and every piece of window / output is two element accesses through two resources.
template <typename TIn, typename TOut, bool MONO = false>
__global__ void __launch_bounds__(kWaves32 * 64)
fir_ols32d_kernel(const TIn *__restrict__ in_base, TOut *__restrict__ out_base, const double *__restrict__ hist_base,
                  const double2 *__restrict__ tw_g, const ArgsD t)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2 *tws = reinterpret_cast<double2 *>(smem_raw);
    double *planes = reinterpret_cast<double *>(tws + 31 * 32);
    // partition 0's tap spectrum: every tile of every wave multiplies by it, and it is all the LDS
    // left over holds (the other partitions' come through L2)
    double2 *h0s = reinterpret_cast<double2 *>(planes + (size_t)kWaves32 * 2 * kPlane32);
    const Args32 &a = t.a;

    fir_history_carry(in_base, hist_base, static_cast<double *>(a.hist_new), a.frames, a.line_stride, a.H, a.C, a.lines);
    for (int i = threadIdx.x; i < 31 * 32; i += kWaves32 * 64)
        tws[i] = tw_g[32 + i];
    for (int i = threadIdx.x; i < kHalf32; i += kWaves32 * 64)
        h0s[i] = t.hpart[i];
    __syncthreads();

    const int wave_u = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane_ = threadIdx.x & 63;
    const int l5_ = lane_ & 31;
    using In2 = typename Pair<TIn>::type;
    const int nb = (int)gridDim.x;
    const int64_t wave_global = (int64_t)wave_u * nb + (int)blockIdx.x;
    const int64_t wave_stride = (int64_t)nb * kWaves32;
    double2 *wave_ring = t.ring + ((int64_t)blockIdx.x * kWaves32 + wave_u) * (int64_t)t.P * (32 * 64);
    const unsigned in_step = (unsigned)(32 * a.C * sizeof(TIn));
    const unsigned out_step = (unsigned)(32 * a.C * sizeof(TOut));
    // (a step past the end of the Line bases its resources beyond the Line: zero records, so that the
    // out-of-range offset of its lanes IS out of range)
    auto bytes31 = [](int64_t n) { return (int)(n < 0 ? 0 : (n < 0x7FFFFFFF ? n : 0x7FFFFFFF)); };
    constexpr int kL = 512;  // hop = partition = window overlap
#ifdef PH_OLSD_PROF
    unsigned dprof[kProfD] = {}, dlast = (unsigned)__builtin_amdgcn_s_memtime();
#endif

    for (int64_t unit = wave_global; unit < t.nunits; unit += wave_stride) {
        // unit -> (series, pair of runs): half h runs tiles [tb, te) of series (line, pair)
        const int series = __builtin_amdgcn_readfirstlane((int)(unit / t.upl));
        const int uidx = __builtin_amdgcn_readfirstlane((int)(unit - (int64_t)series * t.upl));
        const int line = series / a.pairs, c0 = 2 * (series - line * a.pairs);
        const int tb0 = 2 * uidx * t.R;                         // half 0's first tile
        cd lo[16], hi[16];
        // one step: WARM = a window before the run's first tile (its spectrum is all it is for)
        auto step = [&](int rel, auto warm_c) {
            constexpr bool WARM = decltype(warm_c)::value;
            // (per-lane quantities from opaque copies of the lane indices: nothing invariant over the
            // tiles is kept in registers through the transforms)
            int l5 = l5_, lane = lane_;
            asm volatile("" : "+v"(l5), "+v"(lane));
            const int half = lane >> 5;
            double *plane = planes + (wave_u * 2 + half) * kPlane32;
            double *pa = plane + l5;
            double *pb = plane + 33 * l5;
            const double2 *__restrict__ twl = tws + l5 - 32;
            const int tile0 = tb0 + rel;                        // half 0's tile of this step (warm-up: may be < 0)
            const int tile = tile0 + half * t.R;
            // a half whose run lies past the series' last tile does nothing; a warm-up step of a run
            // only transforms; a step past the end of a (shorter, last) run does nothing either
            const bool run_exists = tb0 + half * t.R < a.tiles_per_line;
            const bool out_step_ok = rel >= 0 && tile < a.tiles_per_line;
            const bool valid = run_exists && (rel < 0 || tile < a.tiles_per_line);
            const int cur = ((tile0 % t.P) + t.P) % t.P;        // ring slot of this step's spectrum (R % P == 0: the same for both halves)

            // ---- the window of tile `tile`: frames [tile * 512 - 512, tile * 512 + 512) ----------------
            const int64_t fr00 = (int64_t)tile0 * kL - kL;
            if (fr00 >= 0) {
                const TIn *base = in_base + (int64_t)line * a.line_stride + fr00 * a.C;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<TIn *>(base), 0, bytes31((a.frames - fr00) * a.C * (int64_t)sizeof(TIn)), 0x00020000);
                unsigned v0 = valid ? (unsigned)(((half * t.R * kL + l5) * a.C + c0) * (int)sizeof(TIn)) : kOut32;
                asm volatile("" : "+v"(v0));
                if constexpr (MONO) {
                    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<TIn *>(base + a.mono_shift), 0, bytes31((a.frames - fr00 - a.mono_shift) * (int64_t)sizeof(TIn)), 0x00020000);
                    TIn pa[32], pb[32];
#pragma unroll
                    for (int r = 0; r < 32; ++r) {
                        pa[r] = buf_load_one<TIn>(rs, v0 + (unsigned)r * in_step);
                        pb[r] = buf_load_one<TIn>(rb, v0 + (unsigned)r * in_step);
                    }
#pragma unroll
                    for (int r = 0; r < 32; ++r)
                        PH_NAT(r) = cd{(double)pa[r], (double)pb[r]};
                } else {
                In2 pf[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    if (PH_OLSD_ABLATE & 16) {
                        pf[r].x = (TIn)(v0 + r);
                        pf[r].y = (TIn)(v0 - r);
                    } else {
                        pf[r] = buf_load_pair<TIn>(rs, v0 + (unsigned)r * in_step);
                    }
                }
#pragma unroll
                for (int r = 0; r < 32; ++r)
                    PH_NAT(r) = cd{(double)pf[r].x, (double)pf[r].y};
                }
            } else if constexpr (MONO) {
                // the window reaches back into the history: the real part at frame g, the imaginary part at frame
                // g + mono_shift of the same Line (history below 0 for either): four requests per index, branch-free
                const TIn *in0 = in_base + (int64_t)line * a.line_stride;
                const double *hist0 = hist_base + (int64_t)line * a.H;
                constexpr int kBack = 8192;  // (a window starts at most 4096 + 512 frames before its run)
                const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<TIn *>(in0), 0, bytes31(a.frames * (int64_t)sizeof(TIn)), 0x00020000);
                const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<TIn *>(in0 + a.mono_shift - kBack), 0, bytes31((a.frames - a.mono_shift + kBack) * (int64_t)sizeof(TIn)), 0x00020000);
                const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<double *>(hist0), 0, bytes31((int64_t)a.H * (int64_t)sizeof(double)), 0x00020000);
                const int fr0 = tile * kL - kL;
                const int64_t sh = a.mono_shift;
#pragma unroll 1
                for (int b = 0; b < 32; b += 32) {  // (one trip: keeps the block's addresses out of the transform's registers)
                    int l5w = l5;
                    asm volatile("" : "+v"(l5w));
#pragma unroll
                    for (int r = 0; r < 32; ++r) {
                        const int g = fr0 + l5w + 32 * r;
                        const int64_t g2 = g + sh;
                        const TIn xr = buf_load_one<TIn>(rin, valid && g >= 0 ? (unsigned)(g * (int)sizeof(TIn)) : kOut32);
                        const double hr = buf_load_one<double>(rh, valid && g < 0 && g >= -a.H ? (unsigned)((g + a.H) * 8) : kOut32);
                        const TIn xi = buf_load_one<TIn>(rb, valid && g2 >= 0 ? (unsigned)((g + kBack) * (int)sizeof(TIn)) : kOut32);
                        const double hi2 = buf_load_one<double>(rh, valid && g2 < 0 && g2 >= -(int64_t)a.H ? (unsigned)((int)(g2 + a.H) * 8) : kOut32);
                        PH_NAT(r) = cd{g >= 0 ? (double)xr : hr, g2 >= 0 ? (double)xi : hi2};
                    }
                }
            } else if constexpr (sizeof(TIn) == 8) {
                // the window reaches back into the history; float64 input: the plain ladder (the batched
                // form below costs this variant, whose window alone is 128 registers, 40 bytes of scratch)
                const TIn *__restrict__ in = in_base + (int64_t)line * a.line_stride;
                const double *__restrict__ hist = hist_base + (int64_t)line * a.H * a.C;
                const int64_t fr0 = (int64_t)tile * kL - kL;
                const int64_t last = a.frames - 1;
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const int64_t g = fr0 + l5 + 32 * r;
                    double re = 0.0, im = 0.0;
                    if (valid) {
                        if (g >= 0) {
                            if (g <= last) {
                                re = (double)in[g * a.C + c0];
                                im = (double)in[g * a.C + c0 + 1];
                            }
                        } else if (g >= -(int64_t)a.H) {  // (older frames meet zero-padded taps only)
                            re = hist[(g + a.H) * a.C + c0];
                            im = hist[(g + a.H) * a.C + c0 + 1];
                        }
                    }
                    PH_NAT(r) = cd{re, im};
                }
            } else {
                // the window reaches back into the history (warm-up steps, a Line's first tile): branch-free,
                // batches of requests -- input AND history for every index, the one out of range
                // reading zero through its resource's bounds, then a select (a load per branch of an
                // if / else ladder is 32 dependent round trips, and short Lines start with P of these)
                const TIn *in0 = in_base + (int64_t)line * a.line_stride;
                const double *hist0 = hist_base + (int64_t)line * a.H * a.C;
                const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<TIn *>(in0), 0, bytes31(a.frames * a.C * (int64_t)sizeof(TIn)), 0x00020000);
                const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<double *>(hist0), 0, bytes31((int64_t)a.H * a.C * (int64_t)sizeof(double)), 0x00020000);
                const int fr0 = tile * kL - kL;  // (runs that start at a Line's head: small)
                constexpr int kB = 8;
#pragma unroll
                for (int b = 0; b < 32 / kB; ++b) {
                    In2 pi[kB];
                    double2 ph[kB];
#pragma unroll
                    for (int i = 0; i < kB; ++i) {
                        const int g = fr0 + l5 + 32 * (kB * b + i);
                        pi[i] = buf_load_pair<TIn>(rin, valid && g >= 0 ? (unsigned)((g * a.C + c0) * (int)sizeof(TIn)) : kOut32);
                        ph[i] = buf_load_pair<double>(rh, valid && g < 0 && g >= -a.H ? (unsigned)(((g + a.H) * a.C + c0) * (int)sizeof(double)) : kOut32);
                    }
#pragma unroll
                    for (int i = 0; i < kB; ++i) {
                        const int g = fr0 + l5 + 32 * (kB * b + i);
                        PH_NAT(kB * b + i) = g >= 0 ? cd{(double)pi[i].x, (double)pi[i].y} : cd{ph[i].x, ph[i].y};
                    }
                }
            }
            // (no stamp between the window and its transform: one there costs the build its registers)
            ols32_forward(lo, hi, pa, pb, twl);
            __builtin_amdgcn_sched_barrier(0);
            PH_D_STAMP(1);
            // ---- into the ring; for a tile of the run: Y = sum_p X_{t-p} H_p ----------------------------
            double2 *__restrict__ slot_cur = wave_ring + (int64_t)cur * (32 * 64) + lane;
            const double2 *hp0 = h0s;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                double2 h[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k1 = 4 * g + i;
                    if (PH_OLSD_ABLATE & 8)
                        h[i] = double2{(double)(l5 + k1), 0.5};
                    else
                        h[i] = k1 < 16 ? hp0[32 * k1 + l5] : hp0[1024 - 32 * k1 - l5];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k1 = 4 * g + i;
                    if (!(PH_OLSD_ABLATE & 2) || PH_SPL(k1).re == 1234.5)
                        slot_cur[k1 * 64] = double2{PH_SPL(k1).re, PH_SPL(k1).im};
                    if constexpr (!WARM) {
                        const cd w{h[i].x, k1 < 16 ? h[i].y : -h[i].y};
                        PH_SPL(k1) = cmul(PH_SPL(k1), w);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            PH_D_STAMP(2);
            if constexpr (WARM)
                return;
#pragma unroll 1
            for (int p = 1; p < t.P; ++p) {
                int l5p = l5, lanep = lane;
                asm volatile("" : "+v"(l5p), "+v"(lanep));
                const int sp = cur - p < 0 ? cur - p + t.P : cur - p;
                const double2 *__restrict__ xs = wave_ring + (int64_t)sp * (32 * 64) + lanep;
                const double2 *__restrict__ hp = t.hpart + (size_t)p * (kHalf32 + 1);
                // (group g + 1 is requested before group g's products: two round trips in flight)
                double2 hq[2][4], xq[2][4];
                auto fetch = [&](int g, double2 *hd, double2 *xd) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k1 = 4 * g + i;
                        if (PH_OLSD_ABLATE & 8)
                            hd[i] = double2{(double)(l5p + k1), 0.5};
                        else
                            hd[i] = k1 < 16 ? hp[32 * k1 + l5p] : hp[1024 - 32 * k1 - l5p];
                        if (PH_OLSD_ABLATE & 4)
                            xd[i] = double2{(double)(lanep - k1), 0.25};
                        else
                            xd[i] = xs[k1 * 64];
                    }
                };
                fetch(0, hq[0], xq[0]);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    if (g + 1 < 8)
                        fetch(g + 1, hq[(g + 1) & 1], xq[(g + 1) & 1]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k1 = 4 * g + i;
                        const double2 hh = hq[g & 1][i], xx = xq[g & 1][i];
                        const cd w{hh.x, k1 < 16 ? hh.y : -hh.y};
                        const cd v = cmul(cd{xx.x, xx.y}, w);
                        PH_SPL(k1).re += v.re;
                        PH_SPL(k1).im += v.im;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            PH_D_STAMP(3);
            ols32_inverse_plain(lo, hi, pa, pb, twl);
            PH_D_STAMP(4);
            // ---- store: window index i >= 512 is frame tile * 512 + i - 512 -----------------------------
            {
                const int64_t t00 = (int64_t)tile0 * kL;
                TOut *base = out_base + (int64_t)line * a.line_stride + t00 * a.C;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    base, 0, bytes31((a.frames - t00) * a.C * (int64_t)sizeof(TOut)), 0x00020000);
                const int o0 = ((half * t.R * kL + l5 - kL) * a.C + c0) * (int)sizeof(TOut);
                const int i0 = out_step_ok ? l5 - kL : -2048;
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const int off = o0 + r * (int)out_step;
                    if ((PH_OLSD_ABLATE & 1) && PH_NAT(r).re != 1234.5)
                        continue;
                    if constexpr (MONO) {
                        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
                            base + a.mono_shift, 0, bytes31((a.frames - t00 - a.mono_shift) * (int64_t)sizeof(TOut)), 0x00020000);
                        const unsigned voff = i0 + 32 * r >= 0 ? (unsigned)off : kOut32;
                        buf_store_one<TOut>(rs, voff, PH_NAT(r).re);
                        buf_store_one<TOut>(rb, voff, PH_NAT(r).im);
                    } else {
                        buf_store_pair<TOut>(rs, i0 + 32 * r >= 0 ? (unsigned)off : kOut32, PH_NAT(r).re, PH_NAT(r).im);
                    }
                }
            }
            PH_D_STAMP(5);
        };
#pragma unroll 1
        for (int rel = 1 - t.P; rel < 0; ++rel)
            step(rel, std::true_type{});
#pragma unroll 1
        for (int rel = 0; rel < t.R; ++rel)
            step(rel, std::false_type{});
    }
#ifdef PH_OLSD_PROF
    if (t.prof && lane_ == 0) {
        unsigned long long *dst = t.prof + ((size_t)blockIdx.x * kWaves32 + wave_u) * kProfD;
        for (int i = 0; i < kProfD; ++i)
            dst[i] = dprof[i];
    }
#endif
}

template <typename TIn, typename TOut, bool MONO = false>
int launch32d(const Plan::Impl &I, const void *d_in, void *d_out, const double *hist, ArgsD t, hipStream_t s,
              KernelTimer *timer)
{
    auto kfn = fir_ols32d_kernel<TIn, TOut, MONO>;
    const size_t lds = sizeof(double2) * (31 * 32 + kHalf32 + 1) + sizeof(double) * (size_t)kPlane32 * 2 * kWaves32;
    PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    const int64_t wanted = (t.nunits + kWaves32 - 1) / kWaves32;
    const unsigned grid = (unsigned)(wanted < I.cus ? wanted : I.cus);
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    if (timer)
        PH_TRY(timer->pair(&ev_a, &ev_b));
#ifdef PH_OLSD_PROF
    static DevBuf prof;
    const size_t nw = (size_t)grid * kWaves32;
    if (!prof.p)
        PH_TRY(prof.alloc(sizeof(unsigned long long) * kProfD * kWaves32 * 4096));
    t.prof = static_cast<unsigned long long *>(prof.p);
#endif
    hipExtLaunchKernelGGL(kfn, dim3(grid), dim3(kWaves32 * 64), lds, s, ev_a, ev_b, 0, static_cast<const TIn *>(d_in),
                          static_cast<TOut *>(d_out), hist, static_cast<const double2 *>(I.tw32.p), t);
    PH_HIP(hipGetLastError());
#ifdef PH_OLSD_PROF
    static int shown = 0;
    if (shown++ == 3) {  // (a warm launch)
        PH_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(nw * kProfD);
        PH_HIP(hipMemcpy(h.data(), t.prof, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
        static const char *names[kProfD] = {"-", "window + forward", "ring store + H0", "ring loads + Hp", "inverse transform", "output stores"};
        double sum[kProfD] = {}, tot = 0;
        for (size_t w = 0; w < nw; ++w)
            for (int i = 0; i < kProfD; ++i)
                sum[i] += (double)h[w * kProfD + i];
        for (int i = 0; i < kProfD; ++i)
            tot += sum[i];
        std::fprintf(stderr, "[fir delay-line prof] s_memtime ticks per wave, %zu waves, P = %d, R = %d\n", nw, t.P, t.R);
        for (int i = 0; i < kProfD; ++i)
            std::fprintf(stderr, "[fir delay-line prof]   %-20s %10.1f  %5.1f %%\n", names[i], sum[i] / (double)nw, 100.0 * sum[i] / tot);
    }
#endif
    return PIPE_HIP_OK;
}

template <typename TIn, typename TOut>
int launch32p(const Plan::Impl &I, const void *d_in, void *d_out, const double *hist, ArgsP t, hipStream_t s,
              KernelTimer *timer)
{
    auto kfn = fir_ols32p_kernel<TIn, TOut>;
    const size_t lds = sizeof(double2) * (31 * 32) + sizeof(double) * (size_t)kPlane32 * 2 * kWaves32;
    PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    const int64_t resident = I.cus;  // one 512-thread workgroup per CU
    const int64_t wanted = (t.a.nunits + kWaves32 - 1) / kWaves32;
    const unsigned grid = (unsigned)(wanted < resident ? wanted : resident);
    const int64_t stride = (int64_t)grid * kWaves32;
    t.a.d_slot = (int)(stride % t.a.upl);
    t.a.d_line = (int)(stride / t.a.upl);
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    if (timer)
        PH_TRY(timer->pair(&ev_a, &ev_b));
    hipExtLaunchKernelGGL(kfn, dim3(grid), dim3(kWaves32 * 64), lds, s, ev_a, ev_b, 0, static_cast<const TIn *>(d_in),
                          static_cast<TOut *>(d_out), hist, static_cast<const double2 *>(I.tw32.p), t);
    PH_HIP(hipGetLastError());
    return PIPE_HIP_OK;
}

}  // namespace

int run_ols32p(Plan::Impl &I, const void *d_in, int in_dtype, void *d_out, int out_dtype, const double *hist,
               double *hist_new, int64_t frames, int channels, int lines, hipStream_t s, const char **kernel_name,
               KernelTimer *timer)
{
    // the frequency-domain delay line (two transforms per tile): the plan's spectra are cut every 512
    // taps for it (Plan::init); PIPE_HIP_FIR_PARTITION_SUM selects the sum-of-partitions kernel (A/B)
    if (I.Np == 512 && !PH_ENV_AB("PIPE_HIP_FIR_PARTITION_SUM")) {
        ArgsD d{};
        Args32 &a = d.a;
        a.frames = frames;
        a.hist_new = hist_new;
        a.line_stride = frames * channels;
        a.C = channels;
        a.N = I.N;
        a.H = I.N - 1;
        a.HP = 512;
        a.L = 512;
        a.pairs = channels / 2;
        a.lines = lines;
        a.tiles_per_line = (int)((frames + 511) / 512);
        const bool mono = channels == 1;
        if (mono) {  // two tiles of the one channel per complex sequence: tile t and tile t + half the tiles
            a.pairs = 1;
            a.tiles_per_line = (a.tiles_per_line + 1) / 2;
            a.mono_shift = (int64_t)a.tiles_per_line * 512;
        }
        d.P = I.P;
        const int64_t series = (int64_t)lines * a.pairs;
        const int64_t waves = (int64_t)kWaves32 * I.cus;
        // tiles per run: enough runs for every half-wave of the chip, a multiple of P (both halves of a
        // wave then use the same ring slot).  A run opens with P - 1 warm-up transforms, but a wave is
        // bound by its own chain of round trips: as long as there are idle waves, shorter runs on more
        // of them finish sooner (512 Lines of 8 tiles, 1024 taps: runs of 2 tiles on 1024 waves against
        // runs of 8 on 256).  PIPE_HIP_FIR_RUN_FLOOR=n restores a floor of n P tiles (A/B).
        int64_t R = ((int64_t)a.tiles_per_line * series + 2 * waves - 1) / (2 * waves);
        const char *rf = PH_ENV_AB("PIPE_HIP_FIR_RUN_FLOOR");
        const int run_floor = rf && std::atoi(rf) > 0 ? std::atoi(rf) : 1;
        if (R < run_floor * (int64_t)d.P)
            R = run_floor * (int64_t)d.P;
        R = (R + d.P - 1) / d.P * d.P;
        d.R = (int)R;
        d.upl = (int)((a.tiles_per_line + 2 * R - 1) / (2 * R));
        d.nunits = (int64_t)d.upl * series;
        d.hpart = static_cast<const double2 *>(I.hpart[I.cur].p);
        const size_t need = sizeof(double2) * 32 * 64 * (size_t)kWaves32 * (size_t)I.cus * (size_t)d.P;
        if (I.scratch.bytes < need)
            PH_TRY(I.scratch.alloc(need));
        d.ring = static_cast<double2 *>(I.scratch.p);
        if (mono) {
            if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32) {
                *kernel_name = "fir_ols_kernel<f32,f32,32x32,partitioned>";
                return launch32d<float, float, true>(I, d_in, d_out, hist, d, s, timer);
            }
            if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F32) {
                *kernel_name = "fir_ols_kernel<f64,f32,32x32,partitioned>";
                return launch32d<double, float, true>(I, d_in, d_out, hist, d, s, timer);
            }
            if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F64) {
                *kernel_name = "fir_ols_kernel<f32,f64,32x32,partitioned>";
                return launch32d<float, double, true>(I, d_in, d_out, hist, d, s, timer);
            }
            *kernel_name = "fir_ols_kernel<f64,f64,32x32,partitioned>";
            return launch32d<double, double, true>(I, d_in, d_out, hist, d, s, timer);
        }
        if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32) {
            *kernel_name = "fir_ols_kernel<f32,f32,32x32,partitioned>";
            return launch32d<float, float>(I, d_in, d_out, hist, d, s, timer);
        }
        if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F32) {
            *kernel_name = "fir_ols_kernel<f64,f32,32x32,partitioned>";
            return launch32d<double, float>(I, d_in, d_out, hist, d, s, timer);
        }
        if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F64) {
            *kernel_name = "fir_ols_kernel<f32,f64,32x32,partitioned>";
            return launch32d<float, double>(I, d_in, d_out, hist, d, s, timer);
        }
        *kernel_name = "fir_ols_kernel<f64,f64,32x32,partitioned>";
        return launch32d<double, double>(I, d_in, d_out, hist, d, s, timer);
    }
    ArgsP t{};
    Args32 &a = t.a;
    a.frames = frames;
    a.hist_new = hist_new;
    a.line_stride = frames * channels;
    a.C = channels;
    a.N = I.N;
    a.H = I.N - 1;
    a.HP = I.Np - 1;
    a.L = kM32 - a.HP;
    a.pairs = channels / 2;
    a.lines = lines;
    a.tiles_per_line = (int)((frames + a.L - 1) / a.L);
    a.ipl = a.tiles_per_line * a.pairs;
    a.upl = (a.ipl + 1) / 2;
    a.nunits = (int64_t)a.upl * lines;
    t.P = I.P;
    t.Np = I.Np;
    t.hpart = static_cast<const double2 *>(I.hpart[I.cur].p);
    const size_t need = sizeof(double2) * 32 * 64 * (size_t)kWaves32 * (size_t)I.cus;
    if (I.scratch.bytes < need)
        PH_TRY(I.scratch.alloc(need));
    t.scratch = static_cast<double2 *>(I.scratch.p);
    if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32) {
        *kernel_name = "fir_ols_kernel<f32,f32,32x32,partitioned>";
        return launch32p<float, float>(I, d_in, d_out, hist, t, s, timer);
    }
    if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F32) {
        *kernel_name = "fir_ols_kernel<f64,f32,32x32,partitioned>";
        return launch32p<double, float>(I, d_in, d_out, hist, t, s, timer);
    }
    if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F64) {
        *kernel_name = "fir_ols_kernel<f32,f64,32x32,partitioned>";
        return launch32p<float, double>(I, d_in, d_out, hist, t, s, timer);
    }
    *kernel_name = "fir_ols_kernel<f64,f64,32x32,partitioned>";
    return launch32p<double, double>(I, d_in, d_out, hist, t, s, timer);
}

}  // namespace ols
}  // namespace pipehip
