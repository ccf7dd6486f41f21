// Direct-form FIR Processor for gfx950.
//
// Arithmetic contract (identical to oracle/dsp_oracle.h so that float64 output is
// bit-exact and float32 output is the correctly rounded float64 result):
//     acc = +0.0;  for k = 0..N-1:  acc = fma(h[k], x[n-k], acc)      (binary64)
//
// Mapping to CDNA4:
//   * one workgroup = one (Line, frame tile, channel group); the tile's input
//     window (tile + history) is staged ONCE from HBM into LDS, de-interleaved
//     into per-channel planes and widened to f64, with one pad element every R
//     elements so that the lanes of a wave (stride R) fall on distinct LDS banks;
//   * one lane = R consecutive frames of one channel: R independent f64
//     accumulators, a 2R-deep register window that slides one frame per tap, so
//     each tap costs one ds_read_b64 per R v_fma_f64;
//   * taps are wave-uniform: they are read through the scalar cache (s_load) and
//     enter v_fma_f64 as an SGPR operand -- no VGPR / LDS traffic for taps;
//   * results go back through LDS so the global store is fully coalesced.
// The kernel is bound by the f64 VALU rate (2*N flop per scalar sample against
// 8 B of HBM traffic): DESIGN.md "Roofline".
#include <cstdlib>

#include "common.hpp"

namespace pipehip {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxTaps = 4096;
constexpr size_t kMaxLds = 160 * 1024;

struct FirArgs {
    int64_t frames;      // frames per Line in this call
    int64_t line_stride; // elements between consecutive Lines (= frames*C)
    int C;               // channels
    int CG;              // channels per workgroup
    int N, H;            // taps, history frames (N-1)
    int HP;              // staged history frames, multiple of R, >= H
    int FB;              // frame blocks (lanes) per channel per tile
    int TF;              // frames per tile = FB*R
    int plane;           // padded plane length (elements)
    int nfull, rem;      // N = nfull*R + rem
    int cx_log;          // log2 of the staging column count (2^cx_log >= CG)
    int ablate;          // tuning only (PIPE_HIP_FIR_ABLATE): 1 = skip staging loads, 2 = skip taps
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
template <int R, bool TAIL>
__device__ __forceinline__ void fir_tap_block(double (&acc)[R], const double (&cur)[R],
                                              const double (&nxt)[R], const_f64_ptr taps, int rem)
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

template <int R, typename TIn, typename TOut>
__global__ void __launch_bounds__(kThreads)
fir_direct_kernel(const TIn *__restrict__ in_base, TOut *__restrict__ out_base,
                  const double *__restrict__ hist_base, const double *__restrict__ taps_base,
                  const FirArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *xs = reinterpret_cast<double *>(smem_raw);

    const int line = blockIdx.y;
    const int c0 = blockIdx.z * a.CG;
    const int cg = min(a.CG, a.C - c0);
    const int64_t t0 = (int64_t)blockIdx.x * a.TF;
    const TIn *__restrict__ in = in_base + (int64_t)line * a.line_stride;
    const double *__restrict__ hist = hist_base + (int64_t)line * a.H * a.C;

    // ---- stage [t0-HP, t0+TF) x cg channels into per-channel planes --------
    // lane -> (channel tx, frame row ty) by shifts (CX = 2^cx_log >= cg); kUnroll
    // independent loads are issued before the first is consumed, so a wave keeps
    // kUnroll HBM/L2 requests in flight instead of one.
    const int cx_log = a.cx_log;
    const int tx = threadIdx.x & ((1 << cx_log) - 1);
    const int ty = threadIdx.x >> cx_log;
    const int FY = kThreads >> cx_log;
    const bool tx_ok = tx < cg;
    const int nfr = a.TF + a.HP;
    // frames f < nh precede this call: history (or zeros before it)
    const int64_t nh64 = (int64_t)a.HP - t0;
    const int nh = nh64 > 0 ? (int)nh64 : 0;
    for (int f = ty; f < nh; f += FY) {
        const int64_t g = t0 - a.HP + f;  // < 0
        double v = 0.0;
        if (tx_ok && g >= -(int64_t)a.H)
            v = hist[(g + a.H) * a.C + c0 + tx];
        if (tx_ok)
            xs[tx * a.plane + pad_index<R>(f)] = v;
    }
    {
        constexpr int kUnroll = 8;
        const int64_t last = a.frames - 1;
        // lanes beyond the channel group read channel 0 and discard it: keeps the
        // loads unconditional so that they can all be issued before the first wait
        const TIn *__restrict__ src = in + c0 + (tx_ok ? tx : 0);
        // first row >= nh that this lane owns
        int f0 = ty;
        if (f0 < nh)
            f0 += ((nh - f0 + FY - 1) / FY) * FY;
        for (int fb = f0; fb < nfr && !(a.ablate & 1); fb += kUnroll * FY) {
            TIn v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                int64_t g = t0 - a.HP + fb + u * FY;
                g = g > last ? last : g;  // clamped: the value is discarded below
                v[u] = src[g * a.C];
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int f = fb + u * FY;
                const int64_t g = t0 - a.HP + f;
                if (tx_ok && f < nfr)
                    xs[tx * a.plane + pad_index<R>(f)] = g <= last ? (double)v[u] : 0.0;
            }
        }
    }
    __syncthreads();

    // ---- compute ---------------------------------------------------------------
    const int items = a.FB * cg;
    const int passes = (items + kThreads - 1) / kThreads;
    TOut *os = reinterpret_cast<TOut *>(smem_raw);
    TOut *__restrict__ out = out_base + (int64_t)line * a.line_stride;
    constexpr int kStep = R > 1 ? R + 1 : 1;
    for (int pass = 0; pass < passes; ++pass) {
        const int item = pass * kThreads + threadIdx.x;
        const bool active = item < items;
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
            acc[r] = 0.0;
        int cl = 0, fb = 0;
        if (active) {
            cl = item / a.FB;
            fb = item - cl * a.FB;
            // group g of a plane holds frames [g*R, g*R+R) at elements g*(R+1)..+R-1
            const double *w = xs + cl * a.plane + (fb + a.HP / R) * kStep;
            double cur[R], nxt[R];
#pragma unroll
            for (int j = 0; j < R; ++j)
                cur[j] = w[j];
            const_f64_ptr tp = (const_f64_ptr)taps_base;
            // two tap blocks per trip so the window registers swap roles
            // instead of being copied
            int kb = 0;
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
        // ---- results through LDS for a coalesced store -------------------------
        // (single pass is the common case; with several passes the planes are
        // still needed, so store straight from registers instead)
        if (passes == 1) {
            __syncthreads();  // every lane is done reading the planes
            if (active) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int o = (fb * R + r) * cg + cl;
                    os[o + (o >> 5)] = (TOut)acc[r];
                }
            }
            __syncthreads();
            for (int f = ty; f < a.TF; f += FY) {
                const int64_t g = t0 + f;
                const int i = f * cg + tx;
                if (tx_ok && g < a.frames)
                    out[g * a.C + c0 + tx] = os[i + (i >> 5)];
            }
        } else if (active) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t g = t0 + (int64_t)fb * R + r;
                if (g < a.frames)
                    out[g * a.C + c0 + cl] = (TOut)acc[r];
            }
        }
    }
}

// new history = last H frames of (old history ++ this call's input)
template <typename TIn>
__global__ void fir_hist_update_kernel(const TIn *__restrict__ in, const double *__restrict__ hist_old,
                                       double *__restrict__ hist_new, int64_t frames,
                                       int64_t line_stride, int H, int C)
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
        v = (double)in[(int64_t)line * line_stride + s * C + c];
    else
        v = hist_old[((int64_t)line * H + (s + H)) * C + c];
    hist_new[((int64_t)line * H + j) * C + c] = v;
}

struct Geometry {
    int R, CG, FB, TF, HP, plane;
    size_t lds;
    int64_t blocks;
};

class Fir final : public pipe_hip_processor {
public:
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
        // the largest tile geometry must fit in LDS for at least R = 1
        Geometry g;
        if (!geometry(1, 1, &g))
            return PIPE_HIP_EINVAL;
        return start(stream);
    }

    int start(hipStream_t s) override
    {
        if (hist_bytes_)
            PH_HIP(hipMemsetAsync(hist_[cur_hist_].p, 0, hist_bytes_, s));
        return PIPE_HIP_OK;
    }

    int set_param(int32_t param, const double *values, int32_t count) override
    {
        if (param != PIPE_HIP_PARAM_TAPS || count != N_ || !values)
            return PIPE_HIP_EINVAL;
        // double-buffered: launches already queued keep reading the old copy
        PH_HIP(hipDeviceSynchronize());
        const int nxt = cur_taps_ ^ 1;
        PH_HIP(hipMemcpy(taps_[nxt].p, values, sizeof(double) * (size_t)N_, hipMemcpyHostToDevice));
        cur_taps_ = nxt;
        return PIPE_HIP_OK;
    }

    int run(const void *d_in, int in_dtype, void *d_out, int out_dtype, int64_t frames,
            hipStream_t s) override
    {
        if (frames <= 0)
            return PIPE_HIP_OK;
        Geometry g;
        if (!choose(frames, &g))
            return PIPE_HIP_EINVAL;
        FirArgs a{};
        const double *hist = static_cast<const double *>(hist_[cur_hist_].p);
        const double *taps = static_cast<const double *>(taps_[cur_taps_].p);
        a.frames = frames;
        a.line_stride = frames * cfg.channels;
        a.C = cfg.channels;
        a.CG = g.CG;
        a.N = N_;
        a.H = H_;
        a.HP = g.HP;
        a.FB = g.FB;
        a.TF = g.TF;
        a.plane = g.plane;
        a.nfull = N_ / g.R;
        a.rem = N_ % g.R;
        a.cx_log = 0;
        if (const char *ab = std::getenv("PIPE_HIP_FIR_ABLATE")) {
            a.ablate = std::atoi(ab);
            if (a.ablate & 2) {
                a.nfull = 0;
                a.rem = 0;
            }
        }
        while ((1 << a.cx_log) < g.CG)
            ++a.cx_log;
        const dim3 grid((unsigned)((frames + g.TF - 1) / g.TF), (unsigned)cfg.lines,
                        (unsigned)((cfg.channels + g.CG - 1) / g.CG));
        PH_TRY(timer.begin(s));
        PH_TRY(launch(g, in_dtype, out_dtype, grid, d_in, d_out, hist, taps, a, s));
        PH_TRY(timer.end(s));
        if (H_ > 0) {
            const int n = H_ * cfg.channels;
            const dim3 hg((unsigned)((n + 255) / 256), (unsigned)cfg.lines);
            double *hn = static_cast<double *>(hist_[cur_hist_ ^ 1].p);
            if (in_dtype == PIPE_HIP_F32)
                hipLaunchKernelGGL(fir_hist_update_kernel<float>, hg, dim3(256), 0, s,
                                   static_cast<const float *>(d_in), hist, hn, frames,
                                   a.line_stride, H_, cfg.channels);
            else
                hipLaunchKernelGGL(fir_hist_update_kernel<double>, hg, dim3(256), 0, s,
                                   static_cast<const double *>(d_in), hist, hn, frames,
                                   a.line_stride, H_, cfg.channels);
            PH_HIP(hipGetLastError());
            cur_hist_ ^= 1;
        }
        return PIPE_HIP_OK;
    }

private:
    // tile geometry for register blocking R and a channel-group divisor
    bool geometry(int R, int split, Geometry *g) const
    {
        const int C = cfg.channels;
        int CG = (C + split - 1) / split;
        int FB = (kThreads / CG) / 32 * 32;
        if (FB < 32)
            FB = 32;
        g->R = R;
        g->CG = CG;
        g->FB = FB;
        g->TF = FB * R;
        g->HP = ((N_ + R - 1) / R) * R;
        const int groups = (g->TF + g->HP) / R;
        g->plane = R > 1 ? groups * (R + 1) : groups;
        size_t in_bytes = sizeof(double) * (size_t)g->plane * (size_t)CG;
        size_t out_elems = (size_t)g->TF * (size_t)CG;
        size_t out_bytes = sizeof(double) * (out_elems + (out_elems >> 5) + 1);
        g->lds = in_bytes > out_bytes ? in_bytes : out_bytes;
        return g->lds <= kMaxLds;
    }

    // Largest register blocking R that still gives the chip >= 2 workgroups per
    // CU; small calls fall back to smaller R (more, shorter lanes).  A tile that
    // would not leave room for 2 workgroups per CU in LDS is split across
    // channel groups.
    bool choose(int64_t frames, Geometry *best) const
    {
        static const int Rs[] = {16, 8, 4, 2, 1};
        bool have = false;
        // tuning knob for experiments: PIPE_HIP_FIR_R pins the register blocking
        const char *force = std::getenv("PIPE_HIP_FIR_R");
        const int forced = force ? std::atoi(force) : 0;
        for (int R : Rs) {
            if (forced && R != forced)
                continue;
            Geometry g;
            int split = 1;
            bool ok = geometry(R, split, &g);
            while ((!ok || g.lds > 64 * 1024) && split < cfg.channels)
                ok = geometry(R, ++split, &g);
            if (!ok)
                continue;
            g.blocks = ((frames + g.TF - 1) / g.TF) * cfg.lines * ((cfg.channels + g.CG - 1) / g.CG);
            if (!have || g.blocks > best->blocks)
                *best = g;
            have = true;
            if (g.blocks >= 512)
                break;
        }
        return have;
    }

    template <int R>
    int launch_r(int in_dtype, int out_dtype, const Geometry &g, dim3 grid, const void *d_in,
                 void *d_out, const double *hist, const double *taps, const FirArgs &a, hipStream_t s)
    {
#define PH_FIR_LAUNCH(TI, TO, NAME)                                                                  \
    do {                                                                                             \
        auto kfn = fir_direct_kernel<R, TI, TO>;                                                     \
        if (g.lds > 64 * 1024)                                                                       \
            PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),                          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));     \
        hipLaunchKernelGGL(kfn, grid, dim3(kThreads), g.lds, s, static_cast<const TI *>(d_in),      \
                           static_cast<TO *>(d_out), hist, taps, a);                                 \
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

    int launch(const Geometry &g, int in_dtype, int out_dtype, dim3 grid, const void *d_in,
               void *d_out, const double *hist, const double *taps, const FirArgs &a, hipStream_t s)
    {
        switch (g.R) {
        case 16: return launch_r<16>(in_dtype, out_dtype, g, grid, d_in, d_out, hist, taps, a, s);
        case 8: return launch_r<8>(in_dtype, out_dtype, g, grid, d_in, d_out, hist, taps, a, s);
        case 4: return launch_r<4>(in_dtype, out_dtype, g, grid, d_in, d_out, hist, taps, a, s);
        case 2: return launch_r<2>(in_dtype, out_dtype, g, grid, d_in, d_out, hist, taps, a, s);
        default: return launch_r<1>(in_dtype, out_dtype, g, grid, d_in, d_out, hist, taps, a, s);
        }
    }

    int N_ = 0, H_ = 0;
    DevBuf taps_[2];
    DevBuf hist_[2];
    size_t hist_bytes_ = 0;
    int cur_taps_ = 0, cur_hist_ = 0;
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
