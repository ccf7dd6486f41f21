// Direct-form FIR Processor for gfx950.
//
// Arithmetic contract (identical to oracle/dsp_oracle.h so that float64 output is
// bit-exact and float32 output is the correctly rounded float64 result):
//     acc = +0.0;  for k = 0..N-1:  acc = fma(h[k], x[n-k], acc)      (binary64)
//
// Mapping to CDNA4:
//   * a tile = (Line, TF frames, channel group).  Workgroups (256 lanes, one wave
//     per SIMD, two workgroups per CU) are PERSISTENT: each walks tiles id,
//     id+grid, ...  While tile t is being computed, the global loads of tile
//     t+grid are already in flight into registers (issue-early / write-late), so
//     HBM/L2 latency hides under the tap loop instead of in front of it.
//   * the tile's input window (tile + history) sits in LDS, de-interleaved into
//     per-channel planes and widened to f64, one pad element every R elements and
//     a plane stride chosen so that the lanes of each 32-lane group fall on
//     distinct banks for ds_read_b64;
//   * one lane = R consecutive frames of one channel: R independent f64
//     accumulators and a 2R-deep register window that slides one frame per tap,
//     so each tap costs one ds_read_b64 per R v_fmac_f64;
//   * taps are wave-uniform: read through the constant address space
//     (s_load_dwordx16) they enter v_fmac_f64 as its SGPR operand -- no VGPR or
//     LDS traffic for coefficients;
//   * a wave owns a contiguous span of output frames for all channels of the
//     group, so results are transposed through a wave-private LDS slab (no
//     workgroup barrier) and leave as fully coalesced stores.
// The kernel is bound by the f64 VALU rate (2*N flop per scalar sample against
// 8 B of HBM traffic): DESIGN.md "Kernels and their rooflines".
#include <cmath>
#include <cstdlib>

#include <hip/hip_ext.h>

#include "common.hpp"
#include "fir_hist.hpp"
#include "fir_mfma.hpp"
#include "fir_ols.hpp"

namespace pipehip {
namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxTaps = 4096;
constexpr size_t kMaxLds = 160 * 1024;
constexpr int kPF = 24;  // prefetch registers per lane (rows of the next tile)
#ifndef PIPE_HIP_FIR_LDS_TAPS
#define PIPE_HIP_FIR_LDS_TAPS 0
#endif
constexpr bool kLdsTaps = PIPE_HIP_FIR_LDS_TAPS != 0;

struct FirArgs {
    int64_t frames;       // frames per Line in this call
    int64_t line_stride;  // elements between consecutive Lines (= frames*C)
    int C;                // channels
    int CG;               // channels per tile
    int lines;
    int tiles_per_line;
    int ntiles;           // tiles_per_line * lines * channel groups
    int N, H;             // taps, history frames (N-1)
    int HP;               // staged history frames: multiple of R, >= H
    int lpc_log;          // log2(LPC), LPC = lanes per channel within a wave
    int cx_log;           // log2(CX), CX = pow2 >= CG = staging columns
    int TF;               // frames per tile = kWaves * LPC * R
    int plane;            // plane stride in elements (bank-spread)
    int rows;             // staging rows per lane = ceil((TF+HP) / (256/CX))
    int prefetch;         // rows <= kPF: next tile's loads are issued before compute
    int nfull, rem;       // N = nfull*R + rem
    int out_off;          // byte offset of the wave-private output slabs in LDS
    int out_slab;         // elements per slab (padded)
    int taps_off;         // byte offset of the LDS copy of the taps
    double *hist_new;     // the other half of the history double buffer (written by this launch)
};

// Taps are wave-uniform and immutable during a launch: reading them through the
// constant address space makes hipcc emit s_load_dwordx16 into SGPRs, which then
// feed v_fmac_f64 directly as the scalar operand.
typedef const __attribute__((address_space(4))) double *const_f64_ptr;

template <int R>
__device__ __forceinline__ int pad_index(int f)
{
    return R > 1 ? f + f / R : f;
}

// One block of R taps: tap kk of the block multiplies window element (r - kk),
// which lives in `cur` for r >= kk and in `nxt` (the R frames before) otherwise.
// All register indices are compile-time constants; a TAIL block guards each tap
// with a wave-uniform (scalar) branch instead of padding with zero taps, so NaN /
// Inf handling stays identical to the oracle.
template <int R, bool TAIL, typename TapPtr>
__device__ __forceinline__ void fir_tap_block(double (&acc)[R], const double (&cur)[R],
                                              const double (&nxt)[R], TapPtr taps, int rem)
{
#pragma unroll
    for (int kk = 0; kk < R; ++kk) {
        if (!TAIL || kk < rem) {
            const double h = taps[kk];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double x = (r >= kk) ? cur[r - kk] : nxt[r - kk + R];
                acc[r] = __builtin_fma(h, x, acc[r]);
            }
        }
    }
}

// the same block with its R taps already in registers (the small-call kernel reads them from
// LDS one block ahead)
template <int R>
__device__ __forceinline__ void fir_tap_block_regs(double (&acc)[R], const double (&cur)[R],
                                                   const double (&nxt)[R], const double (&h)[R])
{
#pragma unroll
    for (int kk = 0; kk < R; ++kk) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double x = (r >= kk) ? cur[r - kk] : nxt[r - kk + R];
            acc[r] = __builtin_fma(h[kk], x, acc[r]);
        }
    }
}

struct TileCoord {
    int line, c0, cg;
    int64_t t0;
};

__device__ __forceinline__ TileCoord decode_tile(const FirArgs &a, int id)
{
    TileCoord t;
    const int tile = id % a.tiles_per_line;
    const int rest = id / a.tiles_per_line;
    t.line = rest % a.lines;
    const int g = rest / a.lines;
    t.c0 = g * a.CG;
    t.cg = min(a.CG, a.C - t.c0);
    t.t0 = (int64_t)tile * a.TF;
    return t;
}

// LT (R = 4 only): the variant for calls too small to fill the chip -- one pipe buffer per
// ProcessFunc call is 8 workgroups, one wave per SIMD, nothing to hide a latency behind.  The
// taps sit in LDS and the tap loop is software-pipelined by hand: the window block and the taps
// of block k + 1 are requested before block k's fma run (in-order LDS reads, counted waits).
template <int R, typename TIn, typename TOut, bool LT = false>
__global__ void __launch_bounds__(kThreads)
fir_direct_kernel(const TIn *__restrict__ in_base, TOut *__restrict__ out_base,
                  const double *__restrict__ hist_base, const double *__restrict__ taps_base,
                  const FirArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *xs = reinterpret_cast<double *>(smem_raw);
    constexpr int kStep = R > 1 ? R + 1 : 1;

    // staging map: lane -> (column tx = channel, row ty = frame) by shifts
    const int tx = threadIdx.x & ((1 << a.cx_log) - 1);
    const int ty = threadIdx.x >> a.cx_log;
    const int FY = kThreads >> a.cx_log;
    const int nfr = a.TF + a.HP;
    const int64_t last = a.frames - 1;

    // compute map: wave w, lane l -> channel cl = l / LPC, frame block w*LPC + l % LPC
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int cl = lane >> a.lpc_log;
    const int fbl = lane & ((1 << a.lpc_log) - 1);
    const int LPC = 1 << a.lpc_log;
    const int fb = wave * LPC + fbl;  // frame block within the tile
    TOut *slab = reinterpret_cast<TOut *>(smem_raw + a.out_off) + (size_t)wave * a.out_slab;

    // taps live in LDS for the whole (persistent) workgroup: inside the tap loop
    // every load is then an in-order LDS read (broadcast for the tap), so the
    // waits are counted instead of the full drain that SMEM loads would force
    const double *ltaps = reinterpret_cast<const double *>(smem_raw + a.taps_off);
    if constexpr (kLdsTaps || LT) {
        double *lt = reinterpret_cast<double *>(smem_raw + a.taps_off);
        for (int k = threadIdx.x; k < a.N; k += kThreads)
            lt[k] = taps_base[k];
    }

    TIn pf[kPF];
    int id = blockIdx.x;
    TileCoord tc = decode_tile(a, id < a.ntiles ? id : 0);

    // Interior tiles (no history, no end of stream inside the staged rows) are
    // addressed linearly: one base pointer per lane plus u*FY*C, no clamps, so the
    // prefetch costs kPF registers and little else.  Boundary tiles take
    // stage_sync() below when their turn comes.
    auto interior = [&](const TileCoord &t) -> bool {
        const int64_t first = t.t0 - a.HP;
        return a.prefetch && first >= 0 && first + (int64_t)a.rows * FY - 1 <= last;
    };
    auto issue = [&](const TileCoord &t) {
        const TIn *__restrict__ src = in_base + (int64_t)t.line * a.line_stride +
                                      (t.t0 - a.HP + ty) * a.C + t.c0 + (tx < t.cg ? tx : 0);
        const int64_t stride = (int64_t)FY * a.C;
#pragma unroll
        for (int u = 0; u < kPF; ++u)
            if (u < a.rows)
                pf[u] = src[u * stride];
    };
    // write pf[] into the planes
    auto commit = [&](const TileCoord &t) {
        if (tx < t.cg) {
            double *__restrict__ dst = xs + tx * a.plane;
#pragma unroll
            for (int u = 0; u < kPF; ++u) {
                const int f = ty + u * FY;
                if (u < a.rows && f < nfr)
                    dst[pad_index<R>(f)] = (double)pf[u];
            }
        }
    };
    // boundary tiles, and windows too long to prefetch in registers: clamped
    // loads, written in place
    auto stage_sync = [&](const TileCoord &t) {
        const bool ok = tx < t.cg;
        const TIn *__restrict__ src =
            in_base + (int64_t)t.line * a.line_stride + t.c0 + (ok ? tx : 0);
        const double *__restrict__ hist = hist_base + (int64_t)t.line * a.H * a.C;
        const double *__restrict__ hsrc = hist + t.c0 + (ok ? tx : 0);
        const bool has_hist = a.H > 0 && t.t0 - a.HP < 0;  // uniform: only a Line's first tiles
        for (int fb0 = ty; fb0 < nfr; fb0 += 8 * FY) {
            TIn v[8];
            double hv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int64_t g = t.t0 - a.HP + fb0 + u * FY;
                g = g < 0 ? 0 : (g > last ? last : g);
                v[u] = src[g * a.C];
            }
            // the history rows are requested with the input rows (one round trip, not two:
            // a single pipe buffer per call reads its input over PCIe)
            if (has_hist) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int64_t gh = t.t0 - a.HP + fb0 + u * FY + a.H;
                    gh = gh < 0 ? 0 : (gh > a.H - 1 ? a.H - 1 : gh);
                    hv[u] = hsrc[gh * a.C];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int f = fb0 + u * FY;
                const int64_t g = t.t0 - a.HP + f;
                const double loaded = (double)v[u];  // unconditional use: no load stays pending
                if (ok && f < nfr) {
                    double w = 0.0;
                    if (g >= 0)
                        w = g <= last ? loaded : 0.0;
                    else if (g >= -(int64_t)a.H)
                        w = hv[u];
                    xs[tx * a.plane + pad_index<R>(f)] = w;
                }
            }
        }
    };

    bool fast = id < a.ntiles && interior(tc);
    if (fast)
        issue(tc);

    while (id < a.ntiles) {
        if (fast)
            commit(tc);
        else
            stage_sync(tc);
        __syncthreads();  // planes of tile `id` are complete

        const TileCoord cur_t = tc;
        // The history carry: new history = last H frames of (old history ++ this call's input).
        // A Line's last tile has exactly those frames in its planes (HP >= H, and the first tile
        // staged the old history), so it writes the other half of the double buffer from LDS --
        // no second read of the input.
        if (a.H > 0 && cur_t.t0 + a.TF >= a.frames) {
            double *__restrict__ hn = a.hist_new + (int64_t)cur_t.line * a.H * a.C + cur_t.c0;
            const int64_t f_off = a.frames - a.H - (cur_t.t0 - a.HP);  // plane frame of history row 0
            const int n = a.H * cur_t.cg;
            for (int e = threadIdx.x; e < n; e += kThreads) {
                const int j = e / cur_t.cg;
                const int c = e - j * cur_t.cg;
                hn[(int64_t)j * a.C + c] = xs[c * a.plane + pad_index<R>((int)(f_off + j))];
            }
        }
        const int next = id + gridDim.x;
        fast = false;
        if (next < a.ntiles) {
            tc = decode_tile(a, next);
            fast = interior(tc);
            if (fast)
                issue(tc);  // in flight during the tap loop below
        }

        // ---- tap loop ---------------------------------------------------------
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
            acc[r] = 0.0;
        const bool active = cl < cur_t.cg;
        if (active) {
            // group g of a plane holds frames [g*R, g*R+R) at elements g*(R+1)..+R-1
            const double *w = xs + cl * a.plane + (fb + a.HP / R) * kStep;
            double cur[R], nxt[R];
#pragma unroll
            for (int j = 0; j < R; ++j)
                cur[j] = w[j];
            auto tp = [&] {
                if constexpr (kLdsTaps || LT)
                    return ltaps;
                else
                    return (const_f64_ptr)taps_base;
            }();
            int kb = 0;
            if constexpr (LT) {
                // three window sets and three tap sets take turns (block k uses window blocks
                // k and k + 1 and tap block k; block k + 1's operands are requested first), three
                // blocks per trip so that no register is copied.  The last blocks and the
                // remainder run through the plain loop below.
                double s1[R], s2[R], h0[R], h1[R], h2[R];
                const double *wq = w - kStep;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    s1[j] = wq[j];
                    h0[j] = tp[j];
                }
                for (; kb + 5 <= a.nfull; kb += 3) {
                    const double *tq = tp + (kb + 1) * R;
                    wq -= kStep;
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        s2[j] = wq[j];
                        h1[j] = tq[j];
                    }
                    fir_tap_block_regs<R>(acc, cur, s1, h0);
                    wq -= kStep;
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        cur[j] = wq[j];
                        h2[j] = tq[R + j];
                    }
                    fir_tap_block_regs<R>(acc, s1, s2, h1);
                    wq -= kStep;
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        s1[j] = wq[j];
                        h0[j] = tq[2 * R + j];
                    }
                    fir_tap_block_regs<R>(acc, s2, cur, h2);
                }
                w -= kb * kStep;  // `cur` holds window block kb
                tp += kb * R;
            }
            // two tap blocks per trip so the window registers swap roles
            // instead of being copied
            for (; kb + 2 <= a.nfull; kb += 2) {
                w -= kStep;
#pragma unroll
                for (int j = 0; j < R; ++j)
                    nxt[j] = w[j];
                fir_tap_block<R, false>(acc, cur, nxt, tp, R);
                w -= kStep;
#pragma unroll
                for (int j = 0; j < R; ++j)
                    cur[j] = w[j];
                fir_tap_block<R, false>(acc, nxt, cur, tp + R, R);
                tp += 2 * R;
            }
            if (kb < a.nfull) {
                w -= kStep;
#pragma unroll
                for (int j = 0; j < R; ++j)
                    nxt[j] = w[j];
                fir_tap_block<R, false>(acc, cur, nxt, tp, R);
                tp += R;
#pragma unroll
                for (int j = 0; j < R; ++j)
                    cur[j] = nxt[j];
            }
            if (a.rem) {
                w -= kStep;
#pragma unroll
                for (int j = 0; j < R; ++j)
                    nxt[j] = w[j];
                fir_tap_block<R, true>(acc, cur, nxt, tp, a.rem);
            }
        }

        // ---- wave-private transpose + coalesced store -------------------------
        // the wave owns frames [t0 + wave*LPC*R, +LPC*R) for every channel of the
        // group: slab element (frame offset, channel) -> o = foff*cg + c
        const int cg = cur_t.cg;
        if (active) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int o = (fbl * R + r) * cg + cl;
                slab[o + (o >> 5)] = (TOut)acc[r];
            }
        }
        // same-wave LDS ordering only: no workgroup barrier needed
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const int64_t f0 = cur_t.t0 + (int64_t)wave * LPC * R;  // first frame of the wave
            TOut *__restrict__ out = out_base + (int64_t)cur_t.line * a.line_stride;
            const int nel = LPC * R * cg;
            int64_t lim = (a.frames - f0) * cg;  // elements that fall inside the call
            if (lim > nel)
                lim = nel;
            if (cg == a.C) {
                TOut *__restrict__ dst = out + f0 * a.C;
                for (int e = lane; e < lim; e += 64)
                    dst[e] = slab[e + (e >> 5)];
            } else {
                for (int e = lane; e < lim; e += 64) {
                    const int f = e / cg;
                    const int c = e - f * cg;
                    out[(f0 + f) * a.C + cur_t.c0 + c] = slab[e + (e >> 5)];
                }
            }
        }
        __syncthreads();  // every wave is done with the planes before they are rewritten
        id = next;
    }
}

struct Geometry {
    int R, CG, ngroups, cgp, lpc_log, cx_log, TF, HP, plane, rows, prefetch, out_slab;
    size_t out_off, taps_off, lds;
    int64_t tiles_per_line, ntiles;
    bool lt = false;  // the small-call variant of the kernel (taps in LDS, hand-pipelined tap loop)
};

int ilog2(int v)
{
    int l = 0;
    while ((1 << l) < v)
        ++l;
    return l;
}

// smallest plane stride >= minlen whose residue spreads the (channel, frame
// block) lanes of each 32-lane ds_read_b64 group over distinct 8-byte slots
int pick_plane_stride(int minlen, int R, int lpc_log)
{
    const int step = R > 1 ? R + 1 : 1;
    const int LPC = 1 << lpc_log;
    for (int extra = 0; extra < 64; ++extra) {
        const int ps = minlen + extra;
        bool ok = true;
        for (int half = 0; half < 2 && ok; ++half) {
            unsigned seen = 0;
            for (int l = half * 32; l < half * 32 + 32; ++l) {
                const int cl = l >> lpc_log, fbl = l & (LPC - 1);
                const int slot = (int)(((int64_t)cl * ps + (int64_t)fbl * step) & 31);
                if (seen & (1u << slot)) {
                    ok = false;
                    break;
                }
                seen |= 1u << slot;
            }
        }
        if (ok)
            return ps;
    }
    return minlen;
}

// The fused chain kernel keeps a float32 stream's history as float32 (half the bytes it moves per
// launch); every other kernel reads float64.  A chain that changes form converts: exact either way.
template <typename TSrc, typename TDst>
__global__ void fir_hist_convert_kernel(const TSrc *__restrict__ src, TDst *__restrict__ dst, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (TDst)src[i];
}

class Fir final : public pipe_hip_processor {
public:
    ~Fir() override
    {
        if (taps_uploaded_)
            (void)hipEventDestroy(taps_uploaded_);
    }
    int init(const double *taps, int32_t ntaps)
    {
        N_ = ntaps;
        H_ = ntaps - 1;
        PH_TRY(taps_[0].alloc(sizeof(double) * (size_t)N_));
        PH_TRY(taps_[1].alloc(sizeof(double) * (size_t)N_));
        PH_HIP(hipMemcpy(taps_[0].p, taps, sizeof(double) * (size_t)N_, hipMemcpyHostToDevice));
        const size_t hb = sizeof(double) * (size_t)cfg.lines * (size_t)H_ * (size_t)cfg.channels;
        PH_TRY(hist_[0].alloc(hb));
        PH_TRY(hist_[1].alloc(hb));
        hist_bytes_ = hb;
        hipDeviceProp_t prop;
        PH_HIP(hipGetDeviceProperties(&prop, cfg.device));
        cus_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        Geometry g;
        if (!choose(1, &g))
            return PIPE_HIP_EINVAL;  // window does not fit in LDS even at R = 1
        if (ols::Plan::supports(N_, cfg.channels)) {
            ols_.reset(new ols::Plan());
            PH_TRY(ols_->init(cfg.device, taps, N_, cfg.channels));
        }
        return start(stream);
    }

    int start(hipStream_t s) override
    {
        if (hist_bytes_)
            PH_HIP(hipMemsetAsync(hist_[cur_hist_].p, 0, hist_bytes_, s));
        hist_f32_ = false;  // (all zeros: either layout)
        return PIPE_HIP_OK;
    }

    int start_lines(int first, int count, hipStream_t s) override
    {
        const size_t per = (hist_f32_ ? sizeof(float) : sizeof(double)) * (size_t)H_ * (size_t)cfg.channels;
        if (per && count > 0)
            PH_HIP(hipMemsetAsync(static_cast<char *>(hist_[cur_hist_].p) + per * (size_t)first, 0, per * (size_t)count, s));
        return PIPE_HIP_OK;
    }

    int set_param(int32_t param, const double *values, int32_t count) override
    {
        if (param == PIPE_HIP_PARAM_EXACT && count == 1 && values) {
            exact_ = values[0] != 0.0;
            return PIPE_HIP_OK;
        }
        if (param == PIPE_HIP_PARAM_RELAXED_F64 && count == 1 && values) {  // float64 results may take the overlap-save form too
            relaxed_f64_ = values[0] != 0.0;
            return PIPE_HIP_OK;
        }
        if (param != PIPE_HIP_PARAM_TAPS || count != N_ || !values)
            return PIPE_HIP_EINVAL;
        // Double-buffered on the device: launches already queued keep reading the old copy, the
        // next launch reads the new one.  The upload is a hipMemcpyAsync from pinned staging on
        // the stream this handle's launches go to -- stream order does the rest; nothing waits
        // for the device, and other handles on it (other Lines) do not notice (mutable.go:40-94,
        // pipe.go:433).  A caller that moves the handle between streams orders them itself.
        const size_t bytes = sizeof(double) * (size_t)N_;
        void *host = nullptr;
        PH_TRY(upload_.stage(bytes, &host));
        std::memcpy(host, values, bytes);
        const int nxt = cur_taps_ ^ 1;
        // The caller's stream of the last device-resident call may be gone by now (its owner
        // destroyed it): ask before queueing on it, and use the handle's own stream if so.
        if (last_stream_ && last_stream_ != stream) {
            const hipError_t q = hipStreamQuery(last_stream_);
            if (q != hipSuccess && q != hipErrorNotReady) {
                (void)hipGetLastError();
                last_stream_ = nullptr;
            }
        }
        hipStream_t us = last_stream();
        PH_TRY(upload_.commit(taps_[nxt].p, bytes, us));
        cur_taps_ = nxt;
        if (ols_)
            PH_TRY(ols_->set_taps(values, us));
        // a launch that goes to ANOTHER stream next waits for this upload (run / fuse_view_fir)
        if (!taps_uploaded_)
            PH_HIP(hipEventCreateWithFlags(&taps_uploaded_, hipEventDisableTiming));
        PH_HIP(hipEventRecord(taps_uploaded_, us));
        upload_stream_ = us;
        upload_pending_ = true;
        return PIPE_HIP_OK;
    }

    int run(const void *d_in, int in_dtype, void *d_out, int out_dtype, int64_t frames,
            hipStream_t s) override
    {
        last_flipped_ = false;
        if (frames <= 0)
            return PIPE_HIP_OK;
        PH_TRY(order_after_upload(s));
        last_stream_ = s;
        PH_TRY(ensure_hist_type(false, s));  // these kernels read a float64 history
        // a window of Lines (pipe_hip_process_lines with ragged lengths): the per-Line history
        // slices of exactly those Lines
        const int nl = active_lines();
        const size_t hoff = (size_t)win_first * (size_t)H_ * (size_t)cfg.channels;
        const double *hist = static_cast<const double *>(hist_[cur_hist_].p) + hoff;
        // Large float32 batches take the overlap-save FFT form (<= 1 ulp f32 of the
        // oracle); float64 output, small calls and exact mode keep the ordered-fma
        // direct form (bit-exact).
        // (a run that is queued behind a doorbell keeps to the direct form: nothing in it allocates or synchronises)
        if (ols_ && !exact_ && !queued_run && (out_dtype == PIPE_HIP_F32 || relaxed_f64_out || relaxed_f64_) &&
            ols_wanted(frames, nl) &&
            (!ols_->partitioned() || (reinterpret_cast<uintptr_t>(d_in) % ((cfg.channels == 1 ? 1 : 2) * dtype_size(in_dtype)) == 0 &&
                                       reinterpret_cast<uintptr_t>(d_out) % ((cfg.channels == 1 ? 1 : 2) * dtype_size(out_dtype)) == 0))) {
            PH_TRY(ols_->run(d_in, in_dtype, d_out, out_dtype, hist, hist_next() + hoff, frames, cfg.channels,
                             nl, s, &last_kernel, &timer));
            return flip_history(s);
        }
        const double *taps = static_cast<const double *>(taps_[cur_taps_].p);
        // the ordered-fma form: large calls on the float64 matrix pipe (fir_mfma.hip: the same chain
        // of fused multiply-adds, 1024 of them per instruction), the rest on the VALU
        if (!queued_run && fir_mfma_takes(N_, frames, cfg.channels, nl, cus_, knobs.fir_mfma_min_passes)) {
            hipEvent_t *done = windowed() ? nullptr : &completion;
            PH_TRY(run_fir_mfma(d_in, in_dtype, d_out, out_dtype, hist, hist_next() + hoff, taps, N_, frames, cfg.channels, nl,
                                cus_, s, &last_kernel, &timer, done));
            return flip_history(s);
        }
        Geometry g;
        if (!choose(frames, &g))
            return PIPE_HIP_EINVAL;
        FirArgs a{};
        a.frames = frames;
        a.line_stride = frames * cfg.channels;
        a.C = cfg.channels;
        a.CG = g.CG;
        a.lines = nl;
        a.tiles_per_line = (int)g.tiles_per_line;
        a.ntiles = (int)g.ntiles;
        a.N = N_;
        a.H = H_;
        a.HP = g.HP;
        a.lpc_log = g.lpc_log;
        a.cx_log = g.cx_log;
        a.TF = g.TF;
        a.plane = g.plane;
        a.rows = g.rows;
        a.prefetch = g.prefetch;
        a.nfull = N_ / g.R;
        a.rem = N_ % g.R;
        a.out_off = (int)g.out_off;
        a.out_slab = g.out_slab;
        a.taps_off = (int)g.taps_off;
        a.hist_new = hist_next() + hoff;
        PH_TRY(launch(g, in_dtype, out_dtype, d_in, d_out, hist, taps, a, s));
        return flip_history(s);
    }

    // PIPE_HIP_PARAM_RESIDENT: a queued launch wrote the OTHER half of the history double buffer and
    // flip_history() pointed the stage at it; taking the launch back is pointing it at the old half again
    // (only a run() that got as far as flip_history() has anything to take back: last_flipped_)
    bool armable() const override { return true; }
    void rollback_launch() override
    {
        if (H_ > 0 && last_flipped_)
            cur_hist_ ^= 1;
        last_flipped_ = false;
    }

    bool fuse_view_fir(FirFuseView *v, hipStream_t s, bool prepare) override
    {
        if (!ols_ || windowed())
            return false;
        if (prepare && (order_after_upload(s) != PIPE_HIP_OK ||
                        ensure_hist_type(!v->f64_stream, s) != PIPE_HIP_OK))  // the fused kernel: the history in the stream's own type
            return false;
        v->hist = hist_[cur_hist_].p;
        v->hist_new = hist_next();
        v->plan = &ols_->impl();
        v->taps = static_cast<const double *>(taps_[cur_taps_].p);
        v->ntaps = N_;
        v->relaxed = !exact_;
        v->relaxed_f64 = relaxed_f64_;
        // (the fused chain replaces THREE launches and 4x the traffic: where it pays is the chain's rule, chain.hip)
        v->min_items = knobs.fir_ols_min_items >= 0 ? knobs.fir_ols_min_items : -1;
        v->cus = cus_;
        return true;
    }
    int fuse_commit_fir(hipStream_t s) override
    {
        last_stream_ = s;
        return flip_history(s);
    }

private:
    // Where a device-resident call crosses from the ordered form on the matrix pipe to overlap-save
    // (profiles/r06_fir_small_calls.txt: scripts/fir_small_calls_probe.py over 32 ... 4096 taps, one long Line and many
    // one-buffer Lines).  The ordered form is 4.5 us + 2.8 ps a sample + 0.060 ps a sample and tap at every tap count.
    // An overlap-save launch has a floor, whatever it carries up to a unit a wave (2048 transforms on 256 CUs: a lone
    // wave's unit is that long): 15 - 18 us with one spectrum; partitioned (P = ceil(taps / 512), every run starts with
    // P - 1 transforms of warm-up) 13.8 P^1.5 us for one long Line ... 18.8 P^1.5 for 128 Lines and more (39 / 105 / 310
    // and 53 / 150 / 420 us at 1024 / 2048 / 4096 taps).  Overlap-save where the ordered form's estimate passes that:
    //     one spectrum:  frames x channel pairs x (taps + 45) >= 2.0e8     (256 taps: 860 transforms; 32: 2600; 512: 700)
    //     partitioned:   4.5 + frames x channel pairs x (2.76e-6 + 6.0e-8 taps) >= the floor
    // -- until round 6 the rule was "8 transforms a CU" whatever the taps and the shape, set against the VALU direct form
    // before the matrix-pipe form existed: calls of 1024 - 2047 transforms of a 256-tap filter took 19 - 34 us where
    // overlap-save takes 16 - 17, 256 one-buffer Lines of a 4096-tap filter 418 us where the ordered form takes 330.
    // PIPE_HIP_FIR_OLS_MIN_ITEMS set: that count of transforms alone.
    bool ols_wanted(int64_t frames, int nl) const
    {
        const int64_t items = ols_->items(frames, cfg.channels, nl);
        if (knobs.fir_ols_min_items >= 0)
            return items >= knobs.fir_ols_min_items;
        const double chip = (double)cus_ / 256.0;  // (the constants are a 256-CU chip's)
        const double pair_frames = (double)frames * (double)nl * (double)((cfg.channels + 1) / 2);
        if (!ols_->partitioned())
            return pair_frames * (double)(N_ + 45) >= 2.0e8 * chip;
        const double P = (double)ols_->partitions();
        const double spread = nl <= 1 ? 0.0 : (nl >= 128 ? 1.0 : (double)(nl - 1) / 127.0);
        const double floor_us = (13.8 + 5.0 * spread) * P * std::sqrt(P);
        return 4.5 + pair_frames * (2.76e-6 + 6.0e-8 * (double)N_) / chip >= floor_us;
    }

    // the history of all Lines into the other half of the double buffer in the other element type
    int ensure_hist_type(bool f32, hipStream_t s)
    {
        if (hist_f32_ == f32 || H_ <= 0)
            return PIPE_HIP_OK;
        const int64_t n = (int64_t)cfg.lines * H_ * cfg.channels;
        const unsigned grid = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        if (f32)
            hipLaunchKernelGGL((fir_hist_convert_kernel<double, float>), dim3(grid), dim3(256), 0, s,
                               static_cast<const double *>(hist_[cur_hist_].p), static_cast<float *>(hist_[cur_hist_ ^ 1].p), n);
        else
            hipLaunchKernelGGL((fir_hist_convert_kernel<float, double>), dim3(grid), dim3(256), 0, s,
                               static_cast<const float *>(hist_[cur_hist_].p), static_cast<double *>(hist_[cur_hist_ ^ 1].p), n);
        PH_HIP(hipGetLastError());
        cur_hist_ ^= 1;
        hist_f32_ = f32;
        return PIPE_HIP_OK;
    }

    // new history = last N-1 frames of (old history ++ this call's input)
    // the launch just queued wrote the other half of the history double buffer
    double *hist_next() const { return static_cast<double *>(hist_[cur_hist_ ^ 1].p); }
    int flip_history(hipStream_t s)
    {
        if (H_ <= 0)
            return PIPE_HIP_OK;
        if (windowed()) {
            // only the window's Lines were advanced: bring their new history back into the
            // current half instead of flipping the halves of every Line
            const size_t per = sizeof(double) * (size_t)H_ * (size_t)cfg.channels;
            PH_HIP(hipMemcpyAsync(static_cast<char *>(hist_[cur_hist_].p) + per * (size_t)win_first,
                                  static_cast<const char *>(hist_[cur_hist_ ^ 1].p) + per * (size_t)win_first,
                                  per * (size_t)win_count, hipMemcpyDeviceToDevice, s));
            return PIPE_HIP_OK;
        }
        cur_hist_ ^= 1;
        last_flipped_ = true;
        return PIPE_HIP_OK;
    }
    bool last_flipped_ = false;  // the last run() flipped the history halves (rollback_launch flips them back)

    // tile geometry for register blocking R with the channels split `split` ways
    bool geometry(int R, int split, int64_t frames, Geometry *g) const
    {
        const int C = cfg.channels;
        const int CG = (C + split - 1) / split;
        g->R = R;
        g->CG = CG;
        g->ngroups = (C + CG - 1) / CG;
        g->cx_log = ilog2(CG);
        g->cgp = 1 << g->cx_log;  // <= 64 because channels <= 64
        g->lpc_log = 6 - g->cx_log;
        const int LPC = 1 << g->lpc_log;
        g->TF = kWaves * LPC * R;
        g->HP = ((N_ + R - 1) / R) * R;
        const int groups = (g->TF + g->HP) / R;
        g->plane = pick_plane_stride(R > 1 ? groups * (R + 1) : groups, R, g->lpc_log);
        const int FY = kThreads >> g->cx_log;
        g->rows = (g->TF + g->HP + FY - 1) / FY;
        g->prefetch = g->rows <= kPF ? 1 : 0;
        const size_t in_bytes = sizeof(double) * (size_t)g->plane * (size_t)g->cgp;
        const int slab_el = LPC * R * CG;
        g->out_slab = slab_el + (slab_el >> 5) + 2;
        g->out_off = (in_bytes + 15) & ~(size_t)15;
        g->taps_off = g->out_off + sizeof(double) * (size_t)g->out_slab * kWaves;
        g->lds = g->taps_off + sizeof(double) * (size_t)N_;
        g->tiles_per_line = (frames + g->TF - 1) / g->TF;
        g->ntiles = g->tiles_per_line * active_lines() * g->ngroups;
        return g->lds <= kMaxLds && g->ntiles < (int64_t)1 << 30;
    }

    // Largest register blocking R that still gives the chip >= 2 workgroups per
    // CU; small calls fall back to smaller R (more, shorter lanes).  A tile that
    // would not leave LDS room for 2 workgroups per CU is split across channel
    // groups.
    bool choose(int64_t frames, Geometry *best) const
    {
        static const int Rs[] = {16, 8, 4, 2, 1};
        bool have = false;
        // tuning knob for experiments: PIPE_HIP_FIR_R pins the register blocking
        const char *force = PH_ENV_AB("PIPE_HIP_FIR_R");
        const int forced = force ? std::atoi(force) : 0;
        const int64_t want = 2 * (int64_t)cus_;
        for (int R : Rs) {
            if (forced && R != forced)
                continue;
            Geometry g;
            int split = 1;
            bool ok = geometry(R, split, frames, &g);
            while ((!ok || g.lds > 80 * 1024) && split < cfg.channels)
                ok = geometry(R, ++split, frames, &g);
            if (!ok)
                continue;
            if (!have || g.ntiles > best->ntiles)
                *best = g;
            have = true;
            if (g.ntiles >= want)
                break;
        }
        // A call too small to fill the chip even with one frame per lane (one pipe buffer in
        // the ProcessFunc form) is bound by its launch and staging latencies, not by the number
        // of tiles: R = 4 has the shortest critical path there (measured 29.7 us per 4096x2
        // buffer against 32.7 at R = 1 and 41.4 at R = 16).
        if (have && !forced && best->ntiles < want) {
            Geometry g;
            int split = 1;
            bool ok = geometry(4, split, frames, &g);
            while ((!ok || g.lds > 80 * 1024) && split < cfg.channels)
                ok = geometry(4, ++split, frames, &g);
            if (ok) {
                *best = g;
                static const bool no_lt = PH_ENV_AB("PIPE_HIP_FIR_NO_LT") != nullptr;  // A/B knob
                best->lt = !kLdsTaps && !no_lt;
            }
        }
        return have;
    }

    template <int R>
    int launch_r(int in_dtype, int out_dtype, const Geometry &g, const void *d_in, void *d_out,
                 const double *hist, const double *taps, const FirArgs &a, hipStream_t s)
    {
#define PH_FIR_LAUNCH(TI, TO, NAME)                                                                  \
    do {                                                                                             \
        auto kfn = fir_direct_kernel<R, TI, TO, false>;                                              \
        if constexpr (R == 4)                                                                        \
            if (g.lt)                                                                                \
                kfn = fir_direct_kernel<R, TI, TO, true>;                                            \
        if (g.lds > 64 * 1024)                                                                       \
            PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),                          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));     \
        const dim3 grid = persistent_grid(reinterpret_cast<const void *>(kfn), g);                   \
        hipEvent_t ev_a = nullptr, ev_b = nullptr;                                                   \
        PH_TRY(timer.pair(&ev_a, &ev_b));                                                            \
        if (!ev_b && completion && !windowed()) { /* the launch signals the buffer's completion */   \
            ev_b = completion;                                                                       \
            completion = nullptr;                                                                    \
        }                                                                                            \
        hipExtLaunchKernelGGL(kfn, grid, dim3(kThreads), g.lds, s, ev_a, ev_b, 0,                    \
                              static_cast<const TI *>(d_in), static_cast<TO *>(d_out), hist, taps, a); \
        last_kernel = NAME;                                                                          \
    } while (0)
        if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32)
            PH_FIR_LAUNCH(float, float, "fir_direct_kernel<f32,f32>");
        else if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F64)
            PH_FIR_LAUNCH(double, double, "fir_direct_kernel<f64,f64>");
        else if (in_dtype == PIPE_HIP_F32)
            PH_FIR_LAUNCH(float, double, "fir_direct_kernel<f32,f64>");
        else
            PH_FIR_LAUNCH(double, float, "fir_direct_kernel<f64,f32>");
#undef PH_FIR_LAUNCH
        PH_HIP(hipGetLastError());
        return PIPE_HIP_OK;
    }

    // Persistent grid: as many workgroups as the CUs can hold at once (VGPR / LDS
    // occupancy of this kernel variant), tiles dealt round-robin and balanced so
    // that every workgroup walks the same number of tiles.
    dim3 persistent_grid(const void *kfn, const Geometry &g)
    {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, kThreads, g.lds) != hipSuccess ||
            per_cu < 1) {
            (void)hipGetLastError();
            per_cu = 2;
        }
        if (const char *f = PH_ENV_AB("PIPE_HIP_FIR_WGS_PER_CU"))  // tuning knob
            per_cu = std::atoi(f) > 0 ? std::atoi(f) : per_cu;
        const int64_t slots = (int64_t)per_cu * cus_;
        const int64_t per = (g.ntiles + slots - 1) / slots;
        return dim3((unsigned)((g.ntiles + per - 1) / per));
    }

    int launch(const Geometry &g, int in_dtype, int out_dtype, const void *d_in, void *d_out,
               const double *hist, const double *taps, const FirArgs &a, hipStream_t s)
    {
        switch (g.R) {
        case 16: return launch_r<16>(in_dtype, out_dtype, g, d_in, d_out, hist, taps, a, s);
        case 8: return launch_r<8>(in_dtype, out_dtype, g, d_in, d_out, hist, taps, a, s);
        case 4: return launch_r<4>(in_dtype, out_dtype, g, d_in, d_out, hist, taps, a, s);
        case 2: return launch_r<2>(in_dtype, out_dtype, g, d_in, d_out, hist, taps, a, s);
        default: return launch_r<1>(in_dtype, out_dtype, g, d_in, d_out, hist, taps, a, s);
        }
    }

    hipStream_t last_stream() const { return last_stream_ ? last_stream_ : stream; }
    // a tap upload queued on one stream, the next launch on another: the launch waits for it
    int order_after_upload(hipStream_t s)
    {
        if (upload_pending_) {
            if (s != upload_stream_)
                PH_HIP(hipStreamWaitEvent(s, taps_uploaded_, 0));
            upload_pending_ = false;
        }
        return PIPE_HIP_OK;
    }
    hipEvent_t taps_uploaded_ = nullptr;
    hipStream_t upload_stream_ = nullptr;
    bool upload_pending_ = false;

    int N_ = 0, H_ = 0;
    int cus_ = 256;
    hipStream_t last_stream_ = nullptr;  // where the last launch went: parameter uploads follow it
    AsyncUpload upload_;
    DevBuf taps_[2];
    DevBuf hist_[2];
    size_t hist_bytes_ = 0;
    int cur_taps_ = 0, cur_hist_ = 0;
    bool hist_f32_ = false;  // the history is in the fused chain kernel's float32 layout
    bool exact_ = std::getenv("PIPE_HIP_FIR_EXACT") != nullptr;
    bool relaxed_f64_ = false;  // PIPE_HIP_PARAM_RELAXED_F64: float64 buffers (pipe.go:394,437) may take the overlap-save form
    std::unique_ptr<ols::Plan> ols_;
};

}  // namespace

int make_fir(const pipe_hip_config *cfg, const double *taps, int32_t ntaps, pipe_hip_processor **out)
{
    if (!taps || ntaps < 1 || ntaps > kMaxTaps)
        return PIPE_HIP_EINVAL;
    auto p = std::make_unique<Fir>();
    PH_TRY(p->init_common(cfg));
    PH_TRY(p->init(taps, ntaps));
    *out = p.release();
    return PIPE_HIP_OK;
}

}  // namespace pipehip
