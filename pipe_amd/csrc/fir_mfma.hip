// The bit-exact direct-form FIR on the float64 matrix pipe (large calls; fir.hip keeps the VALU form
// for one-buffer calls and for filters this file does not take).
//
// v_mfma_f64_16x16x4_f64 adds its four products to the accumulator as a chain of IEEE fused
// multiply-adds in k order, k = 0 first (scripts/micro/mfma_f64_order.hip: 51 200 of 51 200 random
// results bit for bit).  The oracle's sample is the ordered chain
//     acc = 0;  for k = 0 .. N - 1:  acc = fma(h[k], x[n - k], acc)                (oracle/dsp_oracle.c)
// so a 16 x 16 tile of outputs D[i][j] = y[T0 + 16 j + i], accumulated over blocks of four inputs from
// the newest to the oldest, is that chain for all 256 outputs at once:
//     A_b[i][q] = h[i - 15 + 4 b + q]        (zero outside 0 .. N - 1)        lane (i = l % 16, q = l / 16)
//     B_b[q][j] = x[T0 + 16 j + 15 - 4 b - q]                                lane (j = l % 16, q = l / 16)
//     D += A_b B_b,  b = 0 .. ceil((N + 15) / 4) - 1
// Row i meets 15 - i zero taps before h[0] and some after h[N - 1]: fma(0, x, acc) == acc for every
// finite x, so the result is the oracle's bit for bit -- unless a zero tap meets an Inf or a NaN.  The
// staging pass therefore looks at every value it converts; a pass that saw a non-finite one computes its
// outputs with the plain ordered loop instead (same LDS window, N taps per output, nothing padded).
//
// Per MFMA (1024 fma): one ds_read_b64 of taps shared by the channel pair + one of inputs per channel.
// The window keeps one pad double per 16 (the 16 columns of a B operand are 16 frames apart: without
// it, one bank), shifted so that a group of four blocks never straddles a pad.
#include <cmath>
#include <cstdlib>

#include <hip/hip_ext.h>

#include "common.hpp"
#include "fir_mfma.hpp"

namespace pipehip {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kMfmaThreads = 256;
constexpr int kMfmaWaves = kMfmaThreads / 64;

struct MfmaArgs {
    int64_t frames;       // per Line, this call
    int64_t line_stride;  // elements between Lines (frames * C)
    int C, lines, N, H;
    int TF;               // frames per pass (a multiple of 256)
    int tiles_per_line;   // ceil(frames / TF)
    int ngroups;          // channel groups of two (the last one of one when C is odd)
    int64_t npass;        // lines * tiles_per_line * ngroups
    int groups16;         // groups of four blocks: ceil((N + 15) / 16)
    int delta;            // window shift: (N - 1 + delta) % 16 == 0
    int c16;              // (N - 1 + delta) / 16
    int wstride;          // doubles per channel window (padded)
    double *hist_new;
};

__device__ __forceinline__ int wpos(int w) { return w + (w >> 4); }

template <typename T>
__device__ __forceinline__ void store_pair(T *p, double a, double b);
template <>
__device__ __forceinline__ void store_pair<float>(float *p, double a, double b)
{
    p[0] = (float)a;
    p[1] = (float)b;
}
template <>
__device__ __forceinline__ void store_pair<double>(double *p, double a, double b)
{
    p[0] = a;
    p[1] = b;
}

// the tiles of one pass (CG channels of the group at once: the taps' operand is shared)
template <int CG, typename TOut>
__device__ __forceinline__ void mfma_tiles(const MfmaArgs &a, const double *hp, const double *win, TOut *__restrict__ out,
                                           int64_t f0, int64_t left, int wave, int q, int j)
{
    const double *ap = hp + j + q;  // A lane (i = lane % 16, q): hp[i + q + 4 b]
    for (int t = wave; t * 256 < a.TF; t += kMfmaWaves) {
        if ((int64_t)t * 256 >= left)
            break;
        // padded position of (tile column j, block 0, q): 17 (16 t + j + c16) + 15 - q; group g: - 17 g; block u: - 4 u
        const double *bp = win + 17 * (16 * t + j + a.c16) + 15 - q;
        const double *bq = bp + (CG == 2 ? a.wstride : 0);
        v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        // Groups of four blocks, two operand sets in flight: the reads of group g + 1 are requested before
        // group g's instructions.  (The set read past the last group is never used; its addresses stay
        // inside the allocation.)
        double ha[4], xa[4], ya[4], hb[4], xb[4], yb[4];
        auto fetch = [&](int g, double (&h)[4], double (&x)[4], double (&y)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                h[u] = ap[16 * g + 4 * u];
                x[u] = bp[-17 * g - 4 * u];
                if (CG == 2)
                    y[u] = bq[-17 * g - 4 * u];
            }
        };
        auto run4 = [&](const double (&h)[4], const double (&x)[4], const double (&y)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(h[u], x[u], acc0, 0, 0, 0);
                if (CG == 2)
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(h[u], y[u], acc1, 0, 0, 0);
            }
        };
        fetch(0, ha, xa, ya);
        int g = 0;
        for (; g + 1 < a.groups16; g += 2) {
            fetch(g + 1, hb, xb, yb);
            run4(ha, xa, ya);
            fetch(g + 2, ha, xa, ya);
            run4(hb, xb, yb);
        }
        if (g < a.groups16)
            run4(ha, xa, ya);
        // D lane (q, j), register r: row 4 r + q, column j -> output 256 t + 16 j + 4 r + q
        const int o0 = 256 * t + 16 * j + q;
        TOut *dst = out + (f0 + o0) * a.C;
        const int step = 4 * a.C;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (o0 + 4 * r < left) {
                if (CG == 2)
                    store_pair<TOut>(dst + r * step, acc0[r], acc1[r]);
                else
                    dst[r * step] = (TOut)acc0[r];
            }
        }
    }
}

template <typename TIn, typename TOut, int PF>
__global__ void __launch_bounds__(kMfmaThreads)
fir_mfma_kernel(const TIn *__restrict__ in_base, TOut *__restrict__ out_base, const double *__restrict__ hist_base,
                const double *__restrict__ taps, const MfmaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *hp = reinterpret_cast<double *>(smem_raw);           // hp[t + 15] = h[t], zeros around: 16 groups16 + 32
    double *win = hp + 16 * a.groups16 + 32;                     // [2 channels][wstride]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;

    for (int i = tid; i < 16 * a.groups16 + 32; i += kMfmaThreads) {
        const int t = i - 15;
        hp[i] = t >= 0 && t < a.N ? taps[t] : 0.0;
    }
    // the `delta` cells below a window's first element are only ever met by zero taps: finite, once
    for (int i = tid; i < 2 * a.delta; i += kMfmaThreads)
        win[(i & 1) * a.wstride + wpos(i >> 1)] = 0.0;
    const int nwin = a.TF + a.N - 1;  // window: frames f0 - (N - 1) .. f0 + TF - 1

    // Passes whose window lies inside the Line (no history, no end of stream) and fits kPF elements per
    // thread are requested a pass ahead: their loads fly under the previous pass's matrix instructions.
    // The others (a Line's first and last pass, filters with long windows) are staged when their turn comes.
    constexpr int kPF = PF;  // (20 covers 512 taps at 2048 frames a pass; 36, float32 input only, every filter this file takes)
    TIn pf[kPF];
    struct Pass {
        int line, c0, cg;
        int64_t f0;
    };
    auto decode = [&](int64_t pass) {
        Pass p;
        const int tile = (int)(pass % a.tiles_per_line);
        const int64_t rest = pass / a.tiles_per_line;
        p.line = (int)(rest % a.lines);
        p.c0 = 2 * (int)(rest / a.lines);
        p.cg = a.C - p.c0 >= 2 ? 2 : 1;
        p.f0 = (int64_t)tile * a.TF;
        return p;
    };
    auto interior = [&](const Pass &p) {
        return p.f0 - (a.N - 1) >= 0 && p.f0 + a.TF <= a.frames && nwin * p.cg <= kPF * kMfmaThreads;
    };
    auto issue = [&](const Pass &p) {
        const TIn *__restrict__ src = in_base + (int64_t)p.line * a.line_stride + (p.f0 - (a.N - 1)) * a.C + p.c0;
#pragma unroll
        for (int i = 0; i < kPF; ++i) {
            const int e = tid + i * kMfmaThreads;
            if (e < nwin * p.cg)
                pf[i] = p.cg == 2 ? src[(int64_t)(e >> 1) * a.C + (e & 1)] : src[(int64_t)e * a.C];
        }
    };
    bool fast = false;
    Pass cur = decode(blockIdx.x < a.npass ? blockIdx.x : 0);
    if ((int64_t)blockIdx.x < a.npass && interior(cur)) {
        issue(cur);
        fast = true;
    }

    for (int64_t pass = blockIdx.x; pass < a.npass; pass += gridDim.x) {
        const int line = cur.line, c0 = cur.c0, cg = cur.cg;
        const int64_t f0 = cur.f0;
        const TIn *__restrict__ in = in_base + (int64_t)line * a.line_stride;
        const double *__restrict__ hist = hist_base + (int64_t)line * a.H * a.C;

        __syncthreads();  // the previous pass is through with the window
        // ---- the window: element w of channel c = frame f0 - (N - 1) + w (history below 0, silence past the end)
        bool odd = false;
        if (fast) {
#pragma unroll
            for (int i = 0; i < kPF; ++i) {
                const int e = tid + i * kMfmaThreads;
                if (e < nwin * cg) {
                    const double v = (double)pf[i];
                    odd = odd || !(__builtin_fabs(v) <= 1.79769313486231570815e308);
                    win[(cg == 2 ? (e & 1) : 0) * a.wstride + wpos((cg == 2 ? e >> 1 : e) + a.delta)] = v;
                }
            }
        } else {
            for (int e = tid; e < nwin * cg; e += kMfmaThreads) {
                const int w = cg == 2 ? e >> 1 : e;
                const int c = cg == 2 ? e & 1 : 0;
                const int64_t g = f0 - (a.N - 1) + w;
                double v = 0.0;
                if (g >= 0) {
                    if (g < a.frames)
                        v = (double)in[g * a.C + c0 + c];
                } else if (g >= -(int64_t)a.H) {
                    v = hist[(g + a.H) * a.C + c0 + c];
                }
                odd = odd || !(__builtin_fabs(v) <= 1.79769313486231570815e308);  // Inf or NaN
                win[c * a.wstride + wpos(w + a.delta)] = v;
            }
        }
        const int nonfinite = __syncthreads_or(odd ? 1 : 0);  // (and the window is complete)
        // the next pass of this workgroup: requested now when it can be
        fast = false;
        if (pass + gridDim.x < a.npass) {
            cur = decode(pass + gridDim.x);
            if (interior(cur)) {
                issue(cur);
                fast = true;
            }
        }

        // ---- the history for the next call: the Line's last tile holds its frames
        if (a.H > 0 && f0 + a.TF >= a.frames) {
            double *__restrict__ hn = a.hist_new + (int64_t)line * a.H * a.C + c0;
            const int w0 = (int)(a.frames - f0);  // window element of history row 0
            for (int e = tid; e < a.H * cg; e += kMfmaThreads) {
                const int r = cg == 2 ? e >> 1 : e;
                const int c = cg == 2 ? e & 1 : 0;
                hn[(int64_t)r * a.C + c] = win[c * a.wstride + wpos(w0 + r + a.delta)];
            }
        }

        TOut *__restrict__ out = out_base + (int64_t)line * a.line_stride + c0;
        const int64_t left = a.frames - f0;  // outputs of this pass: min(TF, left)
        if (nonfinite) {
            // the plain ordered loop: exactly N taps per output, nothing multiplied by a padding zero
            const int nout = (int)(left < a.TF ? left : a.TF);
            for (int e = tid; e < nout * cg; e += kMfmaThreads) {
                const int o = cg == 2 ? e >> 1 : e;
                const int c = cg == 2 ? e & 1 : 0;
                const double *wc = win + c * a.wstride;
                double acc = 0.0;
                for (int k = 0; k < a.N; ++k)
                    acc = __builtin_fma(hp[k + 15], wc[wpos(o + a.N - 1 - k + a.delta)], acc);
                out[(f0 + o) * a.C + c] = (TOut)acc;
            }
            continue;
        }

        // ---- 16 x 16 tiles: tile t of the pass covers outputs [256 t, 256 t + 256), both channels at once
        if (cg == 2)
            mfma_tiles<2, TOut>(a, hp, win, out, f0, left, wave, q, j);
        else
            mfma_tiles<1, TOut>(a, hp, win, out, f0, left, wave, q, j);
    }
}

}  // namespace

bool fir_mfma_takes(int ntaps, int64_t frames, int channels, int lines, int cus, int64_t min_passes)
{
    if (PH_ENV_AB("PIPE_HIP_FIR_NO_MFMA"))
        return false;
    if (ntaps < 16 || ntaps > kFirMfmaMaxTaps || frames <= 0)
        return false;
    // from 32 passes of 1024 frames x 2 channels (8 pipe buffers of 4096 x 2); below that a call is bound
    // by its launch and staging latencies and fir.hip's small-call kernel is as fast (scripts/fir_exact_sweep.py:
    // equal up to 8 buffers, 9.3 - 10.2 us against 12 - 13 from 16 to 64)
    (void)cus;
    const int64_t groups = (channels + 1) / 2;
    const int64_t passes = ((frames + 1023) / 1024) * lines * groups;
    return passes >= min_passes;  // (the handle's PIPE_HIP_FIR_MFMA_MIN_PASSES, default 32; tests: 1 sends small calls here too)
}

int run_fir_mfma(const void *d_in, int in_dtype, void *d_out, int out_dtype, const double *hist, double *hist_new,
                 const double *taps, int ntaps, int64_t frames, int channels, int lines, int cus, hipStream_t s,
                 const char **kernel_name, KernelTimer *timer, hipEvent_t *completion)
{
    MfmaArgs a{};
    a.frames = frames;
    a.line_stride = frames * channels;
    a.C = channels;
    a.lines = lines;
    a.N = ntaps;
    a.H = ntaps - 1;
    a.ngroups = (channels + 1) / 2;
    a.groups16 = (ntaps + 15 + 15) / 16;
    a.delta = ((1 - ntaps) % 16 + 16) % 16;
    a.c16 = (ntaps - 1 + a.delta) / 16;
    a.hist_new = hist_new;
    // frames per pass: 2048 while that still gives every CU two passes, else 1024
    a.TF = 2048;
    if (((frames + 2047) / 2048) * lines * a.ngroups < 2 * (int64_t)cus)
        a.TF = 1024;
    if (const char *e = PH_ENV_AB("PIPE_HIP_FIR_MFMA_TF"))  // A/B: 1024 / 2048 / 4096
        if (std::atoi(e) >= 256 && std::atoi(e) % 256 == 0)
            a.TF = std::atoi(e);
    a.tiles_per_line = (int)((frames + a.TF - 1) / a.TF);
    a.npass = (int64_t)a.tiles_per_line * lines * a.ngroups;
    const int wraw = a.TF + ntaps - 1 + a.delta + 16;
    a.wstride = wraw + wraw / 16 + 1;
    a.wstride += (a.wstride & 1);
    const size_t lds = sizeof(double) * ((size_t)16 * a.groups16 + 32 + 2 * (size_t)a.wstride);
    if (lds > 160 * 1024 - 64)
        return PIPE_HIP_EINVAL;

    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    if (timer)
        PH_TRY(timer->pair(&ev_a, &ev_b));
    if (!ev_b && completion && *completion) {  // the launch signals the buffer's completion
        ev_b = *completion;
        *completion = nullptr;
    }
#define PH_MFMA_LAUNCH(TI, TO, NAME)                                                                               \
    do {                                                                                                           \
        auto kfn = fir_mfma_kernel<TI, TO, 20>;                                                                    \
        if (sizeof(TI) == 4 && (a.TF + ntaps - 1) * 2 > 20 * kMfmaThreads)                                         \
            kfn = fir_mfma_kernel<TI, TO, (sizeof(TI) == 4 ? 36 : 20)>;                                            \
        if (lds > 64 * 1024)                                                                                       \
            PH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)lds));                                                                 \
        int per_cu = 0;                                                                                            \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, kMfmaThreads, lds) != hipSuccess || per_cu < 1) { \
            (void)hipGetLastError();                                                                               \
            per_cu = 1;                                                                                            \
        }                                                                                                          \
        const int64_t resident = (int64_t)per_cu * cus;                                                            \
        const unsigned grid = (unsigned)(a.npass < resident ? a.npass : resident);                                 \
        hipExtLaunchKernelGGL(kfn, dim3(grid), dim3(kMfmaThreads), lds, s, ev_a, ev_b, 0, static_cast<const TI *>(d_in), \
                              static_cast<TO *>(d_out), hist, taps, a);                                            \
        *kernel_name = NAME;                                                                                       \
    } while (0)
    if (in_dtype == PIPE_HIP_F32 && out_dtype == PIPE_HIP_F32)
        PH_MFMA_LAUNCH(float, float, "fir_mfma_kernel<f32,f32>");
    else if (in_dtype == PIPE_HIP_F64 && out_dtype == PIPE_HIP_F64)
        PH_MFMA_LAUNCH(double, double, "fir_mfma_kernel<f64,f64>");
    else if (in_dtype == PIPE_HIP_F32)
        PH_MFMA_LAUNCH(float, double, "fir_mfma_kernel<f32,f64>");
    else
        PH_MFMA_LAUNCH(double, float, "fir_mfma_kernel<f64,f32>");
#undef PH_MFMA_LAUNCH
    PH_HIP(hipGetLastError());
    return PIPE_HIP_OK;
}

}  // namespace pipehip
