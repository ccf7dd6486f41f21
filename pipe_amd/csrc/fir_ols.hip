// Overlap-save FIR for gfx950: the O(log N) form of the FIR Processor.
//
// Why: the direct form costs 2*N f64 flop per scalar sample and is pinned at the
// f64 VALU roof (~15 % of the HBM roofline at N = 256).  Overlap-save
// with a 1024-point float64 FFT costs ~0.7 DP instructions per sample and lane (1064 per
// 1538-sample item) instead of 256 FMAs and moves the kernel towards the memory roof.
//
// Numerics: everything is float64.  Two real channels ride one complex sequence
// (z = ch0 + i*ch1; the taps are real, so Re/Im of the filtered sequence are the
// two filtered channels).  The result differs from the oracle's ordered fma chain
// by O(1e-16) absolute -- far below one float32 ulp -- so float32 output equals
// the correctly rounded oracle value except where the float64 value sits within
// ~1e-15 of a rounding boundary (then it is the neighbouring float32: 1 ulp).
// The path is therefore used for float32 buffers only, never for float64 ones and
// never when the handle asks for bit-exactness (PIPE_HIP_FIR_EXACT).
//
// Mapping: ONE WAVE = one 1024-point complex FFT, held as 16 complex values per
// lane.  1024 = 16 x 16 x 4:
//     A  : 16-point DFT in registers over n2           (n = n1 + 64*n2, lane = n1)
//     B  : twiddle W1024^(n1*k2)                   (LDS table, exactly rounded entries)
//     X1 : wave-private LDS exchange, stride 65    (re and im through one float64 plane)
//     C1 : 16-point DFT in registers over b             (n1 = a + 4*b)
//     C2 : twiddle W64^(a*d)                                            (LDS table)
//     X2 : 4x4 (16-lane row a) x (register) transposes with v_permlane32/16_swap: no LDS
//     C3 : four 4-point DFTs in registers
// then the spectrum is multiplied by the tap spectrum (LDS, H[0..512] + conjugate mirror,
// pre-scaled by 1/M) and the same steps run backwards with conjugate twiddles, ending in
// natural order, lane n1 holding y[n1 + 64*n2].  No workgroup barrier is needed after the
// tables are loaded; waves walk (Line, channel pair, tile) items on their own, 16 per CU.
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>

#include <hip/hip_ext.h>

#include "common.hpp"
#include "fir_hist.hpp"
#include "fir_ols.hpp"
#include "fir_ols_impl.hpp"
#include "ols_math.hpp"

namespace pipehip {
namespace ols {
namespace {

constexpr int kM = 1024;         // FFT size
constexpr int kVecWaves = 16;    // waves per workgroup (= per CU) of the even-channel kernels: 4 per SIMD
constexpr int kHalf = 513;        // H[0..512]; padded to 514 entries in LDS for alignment of what follows
constexpr int kEx = 1040;        // complex elements per wave-private exchange buffer (65 x 16)

// The W64^(a*d) twiddles, applied where a already sits in the register index (r = 4i + a) and
// j = d & 3 in the 16-lane row: t[r] = W64^(a*(4i + j)) for this lane's row.  The four registers
// with a = 0 are multiplied by one: twelve products instead of fifteen.
template <bool CONJ>
__device__ __forceinline__ void twiddle_rows(cd (&v)[16], const double2 *__restrict__ t)
{
    constexpr int G = 4;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        double2 w[G];
#pragma unroll
        for (int i = 0; i < G; ++i)
            w[i] = t[4 * i + g + 1];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int r = 4 * i + g + 1;
            const cd ww{w[i].x, w[i].y};
            v[r] = CONJ ? cmulc(v[r], ww) : cmul(v[r], ww);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// X2 without LDS.  Before: lane = k2 + 16*a (a = the 16-lane row), register d = 4i + j.  The last
// radix-4 step runs over a, so a has to come into registers: for every i the 4x4 block
// (row a) x (register j) is transposed with the cross-row swap instructions of gfx950 --
// v_permlane32_swap (rows {0,1} <-> {2,3} between two registers) then v_permlane16_swap (even <->
// odd rows) -- 64 32-bit swaps for the 16 complex registers.  After: lane = k2 + 16*j, register
// 4i + a.  The transpose is two involutions, so the way back applies them in reverse order.
__device__ __forceinline__ void swap_rows(cd &x, cd &y, bool by32)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    v2u xr = __builtin_bit_cast(v2u, x.re), xi = __builtin_bit_cast(v2u, x.im);
    v2u yr = __builtin_bit_cast(v2u, y.re), yi = __builtin_bit_cast(v2u, y.im);
#define PH_SWAP(A, B)                                                                          \
    do {                                                                                       \
        const auto r_ = by32 ? __builtin_amdgcn_permlane32_swap((A), (B), false, false)        \
                             : __builtin_amdgcn_permlane16_swap((A), (B), false, false);       \
        (A) = r_[0];                                                                           \
        (B) = r_[1];                                                                           \
    } while (0)
    PH_SWAP(xr[0], yr[0]);
    PH_SWAP(xr[1], yr[1]);
    PH_SWAP(xi[0], yi[0]);
    PH_SWAP(xi[1], yi[1]);
#undef PH_SWAP
    x.re = __builtin_bit_cast(double, xr);
    x.im = __builtin_bit_cast(double, xi);
    y.re = __builtin_bit_cast(double, yr);
    y.im = __builtin_bit_cast(double, yi);
}
template <bool FWD>
__device__ __forceinline__ void rows_to_regs(cd (&v)[16])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (FWD) {
            swap_rows(v[4 * i + 0], v[4 * i + 2], true);
            swap_rows(v[4 * i + 1], v[4 * i + 3], true);
            swap_rows(v[4 * i + 0], v[4 * i + 1], false);
            swap_rows(v[4 * i + 2], v[4 * i + 3], false);
        } else {
            swap_rows(v[4 * i + 0], v[4 * i + 1], false);
            swap_rows(v[4 * i + 2], v[4 * i + 3], false);
            swap_rows(v[4 * i + 0], v[4 * i + 2], true);
            swap_rows(v[4 * i + 1], v[4 * i + 3], true);
        }
    }
}

struct Args {
    int64_t frames;       // frames per Line in this call
    int64_t line_stride;  // elements between Lines
    int C, N, H;          // channels, taps, history frames
    int L;                // valid outputs per tile = kM - H
    int pairs;            // ceil(C / 2)
    int lines;
    int tiles_per_line;
    int64_t nitems;       // lines * pairs * tiles_per_line
    int d_pair, d_tile, d_line;  // the wave stride of the launch as (pair, tile, Line) digits
    int group;                   // waves of a block that take consecutive items
    double *hist_new;            // the other half of the history double buffer (written here)
};

// exchange addresses (in doubles, inside the wave-private buffer)
// (strides are odd so that the 16-lane groups of ds_read2_b64 / ds_write_b64 fall on 16 distinct
// 8-byte slots)
__device__ __forceinline__ int ex1_addr(int n1, int k2) { return 65 * k2 + n1; }

// VEC: the channel count is even, so a channel pair is one naturally aligned 8/16-byte
// element: window loads and result stores move whole pairs through buffer resources.
// WAVES = 16 (four waves per SIMD, the even-channel kernels): the exchange buffer is one
// float64 plane per wave (real and imaginary parts go through it one after the other), and there
// is no register prefetch of the next window -- the other three waves hide that latency.
// WAVES = 8 (odd channel counts): complex exchange slabs, next window prefetched into registers.
template <typename TIn, typename TOut, int WAVES, bool VEC>
__global__ void __launch_bounds__(WAVES * 64)
fir_ols_kernel(const TIn *__restrict__ in_base, TOut *__restrict__ out_base,
               const double *__restrict__ hist_base, const double2 *__restrict__ tw1_g,
               const double2 *__restrict__ tw2_g, const double2 *__restrict__ hperm_g, const Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // LDS: tap spectrum H[0..512] (the other half is its conjugate mirror: real taps), the
    // twiddle tables W1024^(n1*k2) [15][64] and W64^(a*d) [4][16], then the exchange planes
    double2 *hspec = reinterpret_cast<double2 *>(smem_raw);
    double2 *tw1s = hspec + kHalf + 1;
    double2 *tw2s = tw1s + 15 * 64;
    constexpr bool SPLIT = WAVES > 8;
    constexpr bool PREFETCH = WAVES <= 8;
    double2 *exbase = tw2s + 4 * 16;                         // [WAVES][kEx] wave-private exchange

    fir_history_carry(in_base, hist_base, a.hist_new, a.frames, a.line_stride, a.H, a.C, a.lines);
    for (int i = threadIdx.x; i < kHalf; i += WAVES * 64)
        hspec[i] = hperm_g[i];
    for (int i = threadIdx.x; i < 15 * 64; i += WAVES * 64)
        tw1s[i] = tw1_g[64 + i];
    if (threadIdx.x < 64)
        tw2s[threadIdx.x] = tw2_g[threadIdx.x];
    __syncthreads();

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    double2 *E = SPLIT ? reinterpret_cast<double2 *>(reinterpret_cast<double *>(exbase) + wave * kEx)
                       : exbase + wave * kEx;
    double *Ed = reinterpret_cast<double *>(E);
    // lane roles
    const int n1 = lane;                        // L0
    const int k2l = lane & 15, al = lane >> 4;  // L1: lane = k2 + 16*a
    const int64_t last = a.frames - 1;
    // twiddle rows: step B reads tw1s[(k2-1)*64 + n1] (consecutive lanes, consecutive slots),
    // step C2 reads tw2s[j*16 + r] (j = this lane's 16-lane row after the transpose: a broadcast)
    const double2 *__restrict__ twB = tw1s + n1 - 64;  // indexed with k2*64
    const double2 *__restrict__ twC = tw2s + al * 16;
    // spectrum entry of register r = 4i + c is frequency k = 256c + 64i + lane; for c >= 2 it is
    // read as conj(H[1024 - k])
    const double2 *__restrict__ hlo = hspec + lane;
    const double2 *__restrict__ hhi = hspec - lane;  // indexed with 1024 - 256c - 64i

    using In2 = typename Pair<TIn>::type;
    // Item coordinates are wave-uniform and live in SGPRs: (Line, tile, channel pair), channel
    // pair fastest -- the pairs of one tile read the same cache lines, and neighbouring waves of
    // a workgroup work on them at the same time.  They advance by the launch's wave stride
    // without a division (a.d_pair / a.d_tile / a.d_line are that stride's digits).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // wave w of block b starts at item (w / G)*(blocks*G) + xb*G + w % G: groups of G = a.group
    // neighbouring waves take neighbouring items (the channel pairs of one tile), and the groups
    // are dealt block after block -- so when the items do not divide evenly by the resident
    // waves, the waves with one item more are spread over all CUs (and SIMDs) instead of filling
    // the first blocks.
    // Blocks are dealt to the 8 XCDs round robin (block b runs on XCD b % 8) and each XCD has its
    // own L2: neighbouring tiles share H of their 1024 frames, so consecutive item groups go to
    // blocks of the SAME XCD (xb = the block's rank in XCD-major order) and the overlap is read
    // from HBM once.
    const int nb = (int)gridDim.x;
    const int xb = nb % 8 == 0 ? ((int)blockIdx.x % 8) * (nb / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
    const int64_t wave_global = (int64_t)(wave_u / a.group) * ((int64_t)nb * a.group) +
                                (int64_t)xb * a.group + wave_u % a.group;
    const int64_t wave_stride = (int64_t)gridDim.x * WAVES;
    struct Item {
        int line, tile, pair;
    };
    Item cur{0, 0, 0};
    if (wave_global < a.nitems) {
        cur.pair = __builtin_amdgcn_readfirstlane((int)(wave_global % a.pairs));
        const int64_t rest = wave_global / a.pairs;
        cur.tile = __builtin_amdgcn_readfirstlane((int)(rest % a.tiles_per_line));
        cur.line = __builtin_amdgcn_readfirstlane((int)(rest / a.tiles_per_line));
    }
    auto advance = [&](Item it) {
        it.pair += a.d_pair;
        if (it.pair >= a.pairs) {
            it.pair -= a.pairs;
            ++it.tile;
        }
        it.tile += a.d_tile;
        if (it.tile >= a.tiles_per_line) {
            it.tile -= a.tiles_per_line;
            ++it.line;
        }
        it.line += a.d_line;
        return it;
    };
    // VEC windows that start inside the Line (every tile but a Line's first) are read through a
    // buffer resource based at the window: 32-bit per-lane offsets, no address arithmetic in the
    // loop, and the frames past the end of the Line read as zero (hardware range check)
    auto interior = [&](const Item &it) { return VEC && (int64_t)it.tile * a.L - a.H >= 0; };
    const unsigned in_lane = (unsigned)((n1 * a.C) * sizeof(TIn));    // + c0*sizeof(TIn) per item
    const unsigned in_step = (unsigned)(64 * a.C * sizeof(TIn));      // 64 frames
    const unsigned out_step = (unsigned)(64 * a.C * sizeof(TOut));
    auto bytes31 = [](int64_t n) { return (int)(n < 0x7FFFFFFF ? n : 0x7FFFFFFF); };
    In2 pf[16];
    auto issue = [&](const Item &it) {
        const int64_t fr0 = (int64_t)it.tile * a.L - a.H;
        const TIn *base = in_base + (int64_t)it.line * a.line_stride + fr0 * a.C;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<TIn *>(base), 0, bytes31((a.frames - fr0) * a.C * (int64_t)sizeof(TIn)), 0x00020000);
        const unsigned v0 = in_lane + (unsigned)(it.pair * 2 * sizeof(TIn));
#pragma unroll
        for (int r = 0; r < 16; ++r)
            pf[r] = buf_load_pair<TIn>(rs, v0 + (unsigned)r * in_step);
    };

    // one exchange: v[r] goes to wr(r), then v[r] is re-read from rd(r)
    auto exchange = [&](cd (&v)[16], auto wr, auto rd) {
        if constexpr (!SPLIT) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                E[wr(r)] = double2{v[r].re, v[r].im};
            wave_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double2 t = E[rd(r)];
                v[r] = cd{t.x, t.y};
            }
            wave_fence();
        } else {
            double re[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ed[wr(r)] = v[r].re;
            wave_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r)
                re[r] = Ed[rd(r)];
            wave_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ed[wr(r)] = v[r].im;
            wave_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r)
                v[r] = cd{re[r], Ed[rd(r)]};
            wave_fence();
        }
    };
    auto x1_lane = [&](int r) { return ex1_addr(n1, r); };                     // (lane n1, reg k2)
    auto x1_grp = [&](int r) { return ex1_addr(al + 4 * r, k2l); };            // (lane k2 + 16a, reg b)

    bool have_pf = false;
    if (PREFETCH && wave_global < a.nitems && interior(cur)) {
        issue(cur);
        have_pf = true;
    }
    for (int64_t item = wave_global; item < a.nitems; item += wave_stride) {
        // the seeds are loop-invariant; without this the compiler hoists all 60 of their
        // powers out of the item loop and spills them
        const int line = cur.line, c0 = cur.pair * 2;
        const bool two = c0 + 1 < a.C;
        const int64_t t0 = (int64_t)cur.tile * a.L, fr0 = t0 - a.H;

        // ---- the window: lane n1, register n2 -> element n1 + 64*n2 -------------
        cd v[16];
        if (!PREFETCH && interior(cur)) {
            issue(cur);
            have_pf = true;
        }
        if (have_pf) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                v[r] = cd{(double)pf[r].x, (double)pf[r].y};
        } else {
            // a Line's first tile (its head is the history) and odd / unaligned layouts
            const TIn *__restrict__ in = in_base + (int64_t)line * a.line_stride;
            const double *__restrict__ hist = hist_base + (int64_t)line * a.H * a.C;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t g = fr0 + n1 + 64 * r;
                double re = 0.0, im = 0.0;
                if (g >= 0) {
                    if (g <= last) {
                        re = (double)in[g * a.C + c0];
                        if (two)
                            im = (double)in[g * a.C + c0 + 1];
                    }
                } else if (g >= -(int64_t)a.H) {
                    re = hist[(g + a.H) * a.C + c0];
                    if (two)
                        im = hist[(g + a.H) * a.C + c0 + 1];
                }
                v[r] = cd{re, im};
            }
        }
        // next item's window: in flight while this one is transformed
        have_pf = false;
        if (item + wave_stride < a.nitems) {
            cur = advance(cur);
            if (PREFETCH && interior(cur)) {
                issue(cur);
                have_pf = true;
            }
        }

        // ---- forward transform -------------------------------------------------
        dft16<-1>(v);    // A: over n2 -> k2
        twiddle<false>(v, twB, 64);  // B: W1024^(n1*k2)
        // X1: (lane n1, reg k2) -> (lane k2 + 16a, reg b) holding element (a + 4b, k2)
        exchange(v, x1_lane, x1_grp);
        dft16<-1>(v);    // C1: over b -> d
        // X2: (lane k2 + 16a, reg d = 4i + j) -> (lane k2 + 16j, reg 4i + a), in registers
        rows_to_regs<true>(v);
        twiddle_rows<false>(v, twC);  // C2: W64^(a*d), d = 4i + j; a = 0 needs none
#pragma unroll
        for (int q = 0; q < 4; ++q)  // C3: over a -> c
            dft4<-1>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);

        // ---- spectrum of the taps (scaled by 1/M) -------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = r >> 2, c = r & 3;
            if (c < 2) {
                const double2 h = hlo[256 * c + 64 * i];
                v[r] = cmul(v[r], cd{h.x, h.y});
            } else {
                const double2 h = hhi[1024 - 256 * c - 64 * i];
                v[r] = cmulc(v[r], cd{h.x, h.y});
            }
        }

        // ---- inverse transform: the same steps backwards, conjugate twiddles -----
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dft4<+1>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        twiddle_rows<true>(v, twC);
        rows_to_regs<false>(v);
        dft16<+1>(v);    // over d -> b
        exchange(v, x1_grp, x1_lane);
        twiddle<true>(v, twB, 64);
        dft16<+1>(v);    // over k2 -> n2 : v[r] = y_circ[n1 + 64 r]

        // ---- store the valid part: window index i >= H is frame t0 + i - H ------
        if constexpr (VEC) {
            // buffer resource based at the tile's first output frame: the frames past the end of
            // the Line fall outside it and are dropped; window indices below H get an offset
            // outside it too
            TOut *base = out_base + (int64_t)line * a.line_stride + t0 * a.C;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                base, 0, bytes31((a.frames - t0) * a.C * (int64_t)sizeof(TOut)), 0x00020000);
            const int o0 = ((n1 - a.H) * a.C + c0) * (int)sizeof(TOut);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int off = o0 + r * (int)out_step;
                buf_store_pair<TOut>(rs, off >= 0 ? (unsigned)off : 0xFFFFFFFFu, v[r].re, v[r].im);
            }
        } else {
            TOut *__restrict__ out = out_base + (int64_t)line * a.line_stride;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = n1 + 64 * r;
                const int64_t g = t0 + i - a.H;
                if (i >= a.H && g <= last) {
                    out[g * a.C + c0] = (TOut)v[r].re;
                    if (two)
                        out[g * a.C + c0 + 1] = (TOut)v[r].im;
                }
            }
        }
    }
}

}  // namespace

// ---- host side ---------------------------------------------------------------
Plan::Plan() : impl_(new Impl) {}
Plan::~Plan() { delete impl_; }

bool Plan::supports(int ntaps, int channels)
{
    (void)channels;
    if (std::getenv("PIPE_HIP_FIR_EXACT"))
        return false;
    // up to 512 taps: one spectrum; 513 .. 4096: partitioned (even channel counts: the 32 x 32 kernel)
    return ntaps >= 16 && (ntaps <= 512 || (ntaps <= 4096 && (channels % 2 == 0 || channels == 1) && !PH_ENV_AB("PIPE_HIP_FIR_NO_PARTITION")));
}

// H[k] = (1/M) sum_n h[n] exp(-2 pi i n k / M) for k = 0..M/2 (the kernels take the upper half from
// H[M - k] = conj(H[k]): real taps), written as interleaved re/im doubles.  A radix-2 transform of
// the zero-padded taps in long double: microseconds, where the defining double sum (513 x N cosl /
// sinl) took milliseconds inside a mutation.
static void tap_spectrum(const double *taps, int N, double *out)
{
    typedef long double ld;
    static std::vector<ld> wr, wi;  // exp(-2 pi i j / M), j < M / 2 (computed once; handles are
                                    // created and mutated from the pipe's own threads one at a time
                                    // per component, but any thread may get here first)
    static std::once_flag once;
    std::call_once(once, [] {
        wr.resize(kM / 2);
        wi.resize(kM / 2);
        for (int j = 0; j < kM / 2; ++j) {
            const ld ang = -2.0L * (ld)kPi * j / kM;
            wr[j] = cosl(ang);
            wi[j] = sinl(ang);
        }
    });
    std::vector<ld> re(kM, 0.0L), im(kM, 0.0L);
    for (int n = 0; n < kM; ++n) {  // bit-reversed input order
        int r = 0;
        for (int b = 0; b < 10; ++b)
            r |= ((n >> b) & 1) << (9 - b);
        re[r] = n < N ? (ld)taps[n] : 0.0L;
    }
    for (int len = 2; len <= kM; len <<= 1) {
        const int half = len / 2, step = kM / len;
        for (int i = 0; i < kM; i += len)
            for (int j = 0; j < half; ++j) {
                const ld cr = wr[j * step], ci = wi[j * step];
                const ld xr = re[i + j + half], xi = im[i + j + half];
                const ld tr = xr * cr - xi * ci, ti = xr * ci + xi * cr;
                re[i + j + half] = re[i + j] - tr;
                im[i + j + half] = im[i + j] - ti;
                re[i + j] += tr;
                im[i + j] += ti;
            }
    }
    for (int k = 0; k < kHalf; ++k) {
        out[2 * k] = (double)(re[k] / kM);
        out[2 * k + 1] = (double)(im[k] / kM);
    }
    out[2 * kHalf] = out[2 * kHalf + 1] = 0.0;  // the pad entry
}

int Plan::init(int device, const double *taps, int ntaps, int channels)
{
    impl_->N = ntaps;
    hipDeviceProp_t prop;
    PH_HIP(hipGetDeviceProperties(&prop, device));
    impl_->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    std::vector<double> t1(2 * 16 * 64), t2(2 * 64);
    for (int k2 = 0; k2 < 16; ++k2)
        for (int n1 = 0; n1 < 64; ++n1) {
            const long double ang = -2.0L * (long double)kPi * (n1 * k2) / kM;
            t1[2 * (k2 * 64 + n1)] = (double)cosl(ang);
            t1[2 * (k2 * 64 + n1) + 1] = (double)sinl(ang);
        }
    // W64^(a*d) laid out for the point where the kernel applies it: row j = d & 3 of the wave,
    // register r = 4i + a (d = 4i + j)
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) {
            const int i = r >> 2, a = r & 3;
            const long double ang = -2.0L * (long double)kPi * (a * (4 * i + j)) / 64;
            t2[2 * (j * 16 + r)] = (double)cosl(ang);
            t2[2 * (j * 16 + r) + 1] = (double)sinl(ang);
        }
    PH_TRY(impl_->tw1.alloc(sizeof(double) * t1.size()));
    PH_TRY(impl_->tw2.alloc(sizeof(double) * t2.size()));
    PH_TRY(impl_->hperm[0].alloc(sizeof(double) * 2 * (kHalf + 1)));
    PH_TRY(impl_->hperm[1].alloc(sizeof(double) * 2 * (kHalf + 1)));
    PH_HIP(hipMemcpy(impl_->tw1.p, t1.data(), sizeof(double) * t1.size(), hipMemcpyHostToDevice));
    PH_HIP(hipMemcpy(impl_->tw2.p, t2.data(), sizeof(double) * t2.size(), hipMemcpyHostToDevice));
    PH_TRY(init_ols32_tables(impl_));
    if (ntaps > 512) {
        // partitions of exactly 512 taps (the last one zero-padded): the hop of the frequency-domain
        // delay line; PIPE_HIP_FIR_PARTITION_SUM cuts them evenly for the sum-of-partitions kernel (A/B)
        impl_->P = (ntaps + 511) / 512;
        // (the A/B kernel pairs channels: a one-channel stream has no pair to ride with and keeps the delay line)
        impl_->Np = PH_ENV_AB("PIPE_HIP_FIR_PARTITION_SUM") && channels != 1 ? (ntaps + impl_->P - 1) / impl_->P : 512;
        const size_t pb = sizeof(double) * 2 * (kHalf + 1) * (size_t)impl_->P;
        PH_TRY(impl_->hpart[0].alloc(pb));
        PH_TRY(impl_->hpart[1].alloc(pb));
    }
    PH_TRY(set_taps(taps, nullptr));
    PH_HIP(hipStreamSynchronize(nullptr));
    return PIPE_HIP_OK;
}

int Plan::set_taps(const double *taps, hipStream_t s)
{
    // double-buffered on the device (launches already queued keep the old spectrum), staged
    // through pinned memory and copied on the handle's stream: no device-wide wait
    const size_t one = sizeof(double) * 2 * (kHalf + 1);
    const int nxt = impl_->cur ^ 1;
    void *host = nullptr;
    if (impl_->P > 1) {  // one spectrum per partition of Np taps (the last one may be shorter)
        PH_TRY(impl_->upload.stage(one * (size_t)impl_->P, &host));
        for (int p = 0; p < impl_->P; ++p) {
            const int first = p * impl_->Np;
            const int n = impl_->N - first < impl_->Np ? impl_->N - first : impl_->Np;
            tap_spectrum(taps + first, n > 0 ? n : 0, static_cast<double *>(host) + (size_t)p * 2 * (kHalf + 1));
        }
        PH_TRY(impl_->upload.commit(impl_->hpart[nxt].p, one * (size_t)impl_->P, s));
        impl_->cur = nxt;
        return PIPE_HIP_OK;
    }
    PH_TRY(impl_->upload.stage(one, &host));
    tap_spectrum(taps, impl_->N, static_cast<double *>(host));
    PH_TRY(impl_->upload.commit(impl_->hperm[nxt].p, one, s));
    impl_->cur = nxt;
    return PIPE_HIP_OK;
}

bool Plan::partitioned() const { return impl_->P > 1; }
int Plan::partitions() const { return impl_->P; }

int64_t Plan::items(int64_t frames, int channels, int lines) const
{
    const int L = kM - ((impl_->P > 1 ? impl_->Np : impl_->N) - 1);
    return ((frames + L - 1) / L) * ((channels + 1) / 2) * (int64_t)lines;
}

template <typename TIn, typename TOut, int WAVES, bool VEC>
static int launch_ols(const Plan::Impl &I, const void *d_in, void *d_out, const double *hist, Args a,
                      hipStream_t s, KernelTimer *timer)
{
    auto kfn = fir_ols_kernel<TIn, TOut, WAVES, VEC>;
    const size_t lds = sizeof(double2) * (kHalf + 1 + 15 * 64 + 4 * 16) + (WAVES > 8 ? sizeof(double) : sizeof(double2)) * (size_t)kEx * WAVES;
    if (lds > 64 * 1024)
        PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, WAVES * 64, lds) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    // every CU gets a block (or as many blocks as there are items for); waves walk the items
    // with the grid's wave stride
    const int64_t resident = (int64_t)per_cu * I.cus;
    const int64_t wanted = (a.nitems + WAVES - 1) / WAVES;
    const unsigned grid = (unsigned)(wanted < resident ? wanted : resident);
    int group = 1;
    while (group < a.pairs && group < WAVES)
        group *= 2;
    a.group = group;
    const int64_t stride = (int64_t)grid * WAVES;
    a.d_pair = (int)(stride % a.pairs);
    a.d_tile = (int)((stride / a.pairs) % a.tiles_per_line);
    a.d_line = (int)(stride / a.pairs / a.tiles_per_line);
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    if (timer)
        PH_TRY(timer->pair(&ev_a, &ev_b));
    hipExtLaunchKernelGGL(kfn, dim3(grid), dim3(WAVES * 64), lds, s, ev_a, ev_b, 0, static_cast<const TIn *>(d_in),
                       static_cast<TOut *>(d_out), hist, static_cast<const double2 *>(I.tw1.p),
                       static_cast<const double2 *>(I.tw2.p), static_cast<const double2 *>(I.hperm[I.cur].p), a);
    PH_HIP(hipGetLastError());
    return PIPE_HIP_OK;
}

int Plan::run(const void *d_in, int in_dtype, void *d_out, int out_dtype, const double *hist, double *hist_new,
              int64_t frames,
              int channels, int lines, hipStream_t s, const char **kernel_name, KernelTimer *timer)
{
    if (impl_->P > 1) {
        const size_t piece = channels == 1 ? 1 : 2;  // (a channel pair, or the one channel's elements)
        if ((channels % 2 != 0 && channels != 1) || reinterpret_cast<uintptr_t>(d_in) % (piece * dtype_size(in_dtype)) != 0 ||
            reinterpret_cast<uintptr_t>(d_out) % (piece * dtype_size(out_dtype)) != 0)
            return PIPE_HIP_EINVAL;  // (the caller asked partitioned_ok() first)
        return run_ols32p(*impl_, d_in, in_dtype, d_out, out_dtype, hist, hist_new, frames, channels, lines, s, kernel_name,
                          timer);
    }
    Args a{};
    a.frames = frames;
    a.hist_new = hist_new;
    a.line_stride = frames * channels;
    a.C = channels;
    a.N = impl_->N;
    a.H = impl_->N - 1;
    a.L = kM - a.H;
    a.pairs = (channels + 1) / 2;
    a.lines = lines;
    a.tiles_per_line = (int)((frames + a.L - 1) / a.L);
    a.nitems = (int64_t)a.tiles_per_line * a.pairs * lines;
    // channel pairs are accessed as one 8- / 16-byte piece (buffer loads and stores: any pair-aligned address
    // will do, e.g. a stream that starts at an odd frame of its buffer)
    // (an odd channel count: the last channel alone in its "pair", pieces then aligned to an element only)
    const size_t pin = (channels % 2 ? 1 : 2) * dtype_size(in_dtype), pout = (channels % 2 ? 1 : 2) * dtype_size(out_dtype);
    const bool vec = reinterpret_cast<uintptr_t>(d_in) % pin == 0 && reinterpret_cast<uintptr_t>(d_out) % pout == 0;
    const bool vec16 = channels % 2 == 0 && reinterpret_cast<uintptr_t>(d_in) % 16 == 0 && reinterpret_cast<uintptr_t>(d_out) % 16 == 0;
    // the 32 x 32 decomposition (one transform per half-wave, fir_ols32.hip): even channel counts
    // (default; PIPE_HIP_OLS_VARIANT=16 selects the 16 x 16 x 4 kernel of this file for A/B runs)
    static const int variant = PH_ENV_AB("PIPE_HIP_OLS_VARIANT") ? std::atoi(PH_ENV_AB("PIPE_HIP_OLS_VARIANT")) : 32;
    if (vec && variant == 32)
        return run_ols32(*impl_, d_in, in_dtype, d_out, out_dtype, hist, hist_new, frames, channels, lines, s,
                         kernel_name, timer);
    if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32) {
        *kernel_name = "fir_ols_kernel<f32,f32>";
        if (vec16)
            return launch_ols<float, float, kVecWaves, true>(*impl_, d_in, d_out, hist, a, s, timer);
        return launch_ols<float, float, 8, false>(*impl_, d_in, d_out, hist, a, s, timer);
    }
    if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F32) {
        *kernel_name = "fir_ols_kernel<f64,f32>";
        if (vec16)
            return launch_ols<double, float, kVecWaves, true>(*impl_, d_in, d_out, hist, a, s, timer);
        return launch_ols<double, float, 8, false>(*impl_, d_in, d_out, hist, a, s, timer);
    }
    // float64 output: only as an intermediate of a chain that ends in float32
    if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F64) {
        *kernel_name = "fir_ols_kernel<f32,f64>";
        if (vec16)
            return launch_ols<float, double, kVecWaves, true>(*impl_, d_in, d_out, hist, a, s, timer);
        return launch_ols<float, double, 8, false>(*impl_, d_in, d_out, hist, a, s, timer);
    }
    if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F64) {
        *kernel_name = "fir_ols_kernel<f64,f64>";
        if (vec16)
            return launch_ols<double, double, kVecWaves, true>(*impl_, d_in, d_out, hist, a, s, timer);
        return launch_ols<double, double, 8, false>(*impl_, d_in, d_out, hist, a, s, timer);
    }
    return PIPE_HIP_EINVAL;
}

}  // namespace ols
}  // namespace pipehip
