// The polyphase resampler by ROWS, for long streams of 2, 4, 8 ... channels (gfx950).
//
// Contract as in resampler.hip (oracle/dsp_oracle.c: odsp_resampler_process): output m reads input frame
// n = floor(m down / up) with phase p = (m down) mod up,
//     acc = +0.0;  for j = 0 .. T-1:  acc = fma(proto[p + j up], x[n - j], acc)      (binary64, j ascending)
//
// What bounds the other forms (profiles/r05_resampler_wave_profile.txt): their lanes are CONSECUTIVE outputs, so every
// lane has taps of its own (2 x T float64 registers per lane) and a window of its own that it reads from LDS for
// every output: 56 LDS reads and 212 vector instructions per wave and step around 98 fma.
//
// Here a wave's 64 lanes are the SAME output of different periods of the phase pattern: lane (row, channel) computes
// outputs i, i + 1, i + 2, ... of its row, a row being B periods (B up outputs from B down input frames, so that
// every row starts at phase 0).  All lanes are at the same phase at the same time:
//   - the taps of an output are wave-uniform: they sit in SCALAR registers (six s_load_dwordx8 per output for 24
//     taps, from a table laid out in output order) and enter v_fma_f64 as its scalar operand -- no vector register,
//     no LDS read, no per-lane select;
//   - a lane's window is its OWN row's, T samples in float64 REGISTERS; consecutive outputs slide it by 0, 1 or 2
//     frames.  The code is unrolled over one turn of the ring (slot r = frame index mod T), so "sliding" is which
//     registers an fma names: a sample is converted once, written into one slot, and read by T outputs' fma
//     without ever moving;
//   - which outputs fire after which frame is scalar control flow (the phase pattern is the same in every row).
// A lane holds a PAIR of channels (two independent fma chains; one alone waits out every fma's latency).  Per output
// and lane: 2 T fma, two conversions, one LDS store; per input frame: one LDS read and two conversions.
//
// Memory.  Rows are 1176 bytes apart (147 float32 stereo frames) and all lanes want frame t of their rows at the
// same time.  First version: every wave staged 64-byte pieces of its 64 rows through LDS by itself -- bit-exact, and
// 50 us a launch against the wave kernel's 29: a quarter of the fma, tap loads or not, made no difference; without
// the stores 38 us, without the loads 34 (profiles/r05_resampler_rows_ablation.txt).  Half-used cache lines, every
// frame fetched twice (a segment refills its window with the T - 1 frames ahead of it), and a launch that is ONE
// round of waves, all loading, then all computing.  Now a WORKGROUP owns 128 / C consecutive rows -- one contiguous
// stretch of the stream, 75 KB for 160 / 147 -- loads it once, fully coalesced, into LDS (rows padded apart so that
// the lanes' frame-t reads spread over the banks), and its waves each compute one segment of every row (the window's
// refill is LDS reads).  The workgroup walks blocks of rows; the next block's input flies (in registers) under the
// current block's tap loops and is deposited into a SECOND buffer, so a block costs one barrier; the results leave
// from the lanes as they are made (8 bytes per lane and output, 16 with two channels: a lane's consecutive outputs
// are adjacent).  (The version before parked the results in LDS -- 82 KB beside ONE buffer of input -- for a fully
// coalesced store behind a second barrier: load, deposit, tap loops, store followed each other.)
// Same operations in the same order per output as every other form: bit for bit the oracle's.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <utility>
#include <vector>

#include <hip/hip_ext.h>

#include "resampler_rows.hpp"

namespace pipehip {
namespace rows {
namespace {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

#ifndef PH_RR_ABLATE
#define PH_RR_ABLATE 0  // scripts/build_ablate_lib.sh resampler_rows PH_RR_ABLATE rr 1 2 3 4: one cost removed at a time (wrong results)
#endif

// scripts/build_ablate_lib.sh resampler_rows PH_RR_PROF rrprof 1: s_memtime ticks per phase and wave, printed by the
// 15th launch (profiles/r05_resampler_rows_profile.txt)
#ifdef PH_RR_PROF
#define PH_RR_STAMP(i)                                                \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        rrprof[i] += now_ - rrlast;                                   \
        rrlast = now_;                                                \
    } while (0)
#else
#define PH_RR_STAMP(i) \
    do {               \
    } while (0)
#endif

template <typename F, int... Is>
__device__ __forceinline__ void for_each_const(std::integer_sequence<int, Is...>, F &&f)
{
    (f(std::integral_constant<int, Is>{}), ...);
}

typedef double v4d __attribute__((ext_vector_type(4)));
// Four taps into scalar registers.  Written as an instruction: left to the compiler, the loads of the NEXT output's
// taps are hoisted above the fma that still read the current ones, into registers of their own (the kernel has none
// to spare: 2 T scalar registers of taps), and come back through v_writelane / v_readlane.  The value is valid
// behind taps_wait().
template <int OFF>
__device__ __forceinline__ v4d taps_load(const double *row)
{
    v4d r;
    asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(r) : "s"(row), "n"(OFF));
    return r;
}
template <int N>
__device__ __forceinline__ void taps_wait(v4d (&h)[N])
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int g = 0; g < N; ++g)
        asm volatile("" : "+s"(h[g]));  // uses of the taps stay behind the wait
}

template <typename TIn, typename TOut, int TT>
__global__ void __launch_bounds__(768) resample_rows_kernel(const Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int H = TT - 1;
    constexpr int NC = TT / 4;
    constexpr int PEI = 16 / (int)sizeof(TIn);  // samples per 16-byte piece
    const int C = a.C;      // channels (even); a lane holds one PAIR of them (two independent fma chains)
    const int rpb = a.rpb;  // rows of a workgroup: lane = (row, pair), 64 / (C / 2) of them (6 channels: 21 rows, one lane idle)
    // two buffers of a block's input (rows -1 .. rpb - 1, in_stride samples apart): block b's in buffer b & 1
    TIn *const inb0 = reinterpret_cast<TIn *>(smem);
    TIn *const inb1 = reinterpret_cast<TIn *>(smem + a.out_off);  // (out_off == 0: rows so long that only ONE buffer fits -- a second barrier instead)

    const int lane = (int)threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int row_in = a.row_in, row_out = a.row_out;
    const int NE = row_in * C;
    const int nblocks = a.lines * a.blocks_per_line;

    // ---- a block's stretch of the input, on its way into LDS: rows -1 .. rpb - 1 of NE samples in 16-byte pieces (a
    // row's last piece may reach into the next row's first samples: the pad takes them; of row -1 only the T - 1
    // frames ahead of row 0 are read later).  Piece idx = tid + u blockDim of the flat (row, piece) list is this
    // thread's u-th: ALL of a block's loads are issued before the first is waited for -- and one block ahead: the
    // workgroup walks blocks blockIdx, + gridDim, ..., and block b + 1's pieces fly while block b is computed.
    constexpr int UMAX = 8;  // (resampler.hip's launch_rows checks that a block's pieces fit)
    v4u pre[UMAX];
    const int npieces = (NE + PEI - 1) / PEI;
    const int total_pieces = (rpb + 1) * npieces;
    auto request = [&](int b) {
        const int line = b / a.blocks_per_line;
        const int blk = b - line * a.blocks_per_line;
        const int fbw = a.fb0 + blk * rpb * row_in;  // frame 0 of the block's row 0, relative to the call's input
        const TIn *__restrict__ lin = reinterpret_cast<const TIn *>(a.in) + (int64_t)line * a.in_frames * C;
        const double *__restrict__ lhist = a.hist + (int64_t)line * H * C;
        // the Line's input as one raw buffer: a piece past its end reads as zero (frames no emitted output reads)
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<TIn *>(lin), 0, (int)(a.in_frames * C * (int64_t)sizeof(TIn)), 0x00020000);
#pragma unroll
        for (int u = 0; u < UMAX; ++u) {
            const int idx = (int)threadIdx.x + u * (int)blockDim.x;
            if (idx >= total_pieces)
                break;
            const int r1 = (int)__umulhi((unsigned)idx, a.piece_magic);  // idx / npieces: the row, counted from row -1
            const int pc = idx - r1 * npieces;
            const int e0 = (fbw + (r1 - 1) * row_in) * C + pc * PEI;  // the piece's first sample, relative to the Line's input
            if (e0 >= 0) {
                pre[u] = (v4u)__builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)e0 * (unsigned)sizeof(TIn), 0, 0);
            } else {  // a stream's first row: history below frame 0, silence below the history
                struct alignas(16) Piece {
                    TIn s[PEI];
                } p;
#pragma unroll
                for (int k = 0; k < PEI; ++k) {
                    const int e = e0 + k;
                    const int g = e >= 0 ? e / C : -((-e + C - 1) / C);  // floor(e / C)
                    const int c = e - g * C;
                    p.s[k] = g >= 0 ? (g < a.in_frames ? lin[(int64_t)g * C + c] : (TIn)0)
                                    : (g >= -H ? (TIn)lhist[(g + H) * C + c] : (TIn)0);  // (history frames ARE values of the input's type)
                }
                pre[u] = __builtin_bit_cast(v4u, p);
            }
        }
    };
    auto deposit = [&](TIn *inb) {
#pragma unroll
        for (int u = 0; u < UMAX; ++u) {
            const int idx = (int)threadIdx.x + u * (int)blockDim.x;
            if (idx >= total_pieces)
                break;
            const int r1 = (int)__umulhi((unsigned)idx, a.piece_magic);
            const int pc = idx - r1 * npieces;
            v2u *d = reinterpret_cast<v2u *>(inb + r1 * a.in_stride + pc * PEI);
            d[0] = v2u{pre[u].x, pre[u].y};
            d[1] = v2u{pre[u].z, pre[u].w};
        }
    };

    // Order of a turn: [barrier: block b's input is complete in buffer b & 1] tap loops of block b, the results leaving
    // from the lanes as they are made; then block b + 1's input from registers into the OTHER buffer (nobody reads it:
    // its last readers were block b - 1's tap loops, and every wave was past those when it came to the barrier) and block
    // b + 2's requested.  One barrier a block, no phase in which every wave stores, no results parked in LDS: the version
    // before parked them (82 KB beside ONE buffer of input) for a fully coalesced store behind a second barrier --
    // load, deposit, tap loops, store followed each other, the tap loops 55 % of a block's time.
#ifdef PH_RR_PROF
    unsigned long long rrprof[8] = {}, rrlast = __builtin_amdgcn_s_memtime();
#endif
    int b = (int)blockIdx.x;
    int par = 0;
    if (PH_RR_ABLATE != 4 && b < nblocks) {
        request(b);
        deposit(inb0);
        if (b + (int)gridDim.x < nblocks)
            request(b + (int)gridDim.x);
    }
    PH_RR_STAMP(0);
    for (; b < nblocks; b += (int)gridDim.x, par ^= 1) {
    __syncthreads();
    PH_RR_STAMP(1);
    TIn *const inb = par ? inb1 : inb0;
    const int line = b / a.blocks_per_line;
    const int blk = b - line * a.blocks_per_line;
    const int obw = a.ob0 + blk * rpb * row_out;  // output 0 of the block's row 0, relative to the call's first output
    TOut *__restrict__ lout = reinterpret_cast<TOut *>(a.out) + (int64_t)line * a.out_cap * C;

    // ---- the wave's segment: outputs [i0, i1) of every row; everything here is wave-uniform
    const int up = a.up;
    // (resampler.hip's launch_rows cuts the row: equal segments)
    const int i0 = a.seg_b[wave], i1 = a.seg_b[wave + 1];
    if (i0 < i1) {
        int i = i0;
        int n_out = (int)(((int64_t)i0 * a.down) / up);              // newest frame output i reads (row-relative)
        int tm = (int)((int64_t)i0 * a.down - (int64_t)n_out * up);  // (i down) mod up
        const int dq = a.down / up, dr = a.down - dq * up;
        const int lrow = (int)(((unsigned)lane * a.pair_rcp) >> 16);  // lane / (C / 2)
        const bool lane_on = lrow < rpb;                              // (a channel count that does not divide 128 leaves lanes over)
        const int row = lane_on ? lrow : 0, ch = 2 * (lane - lrow * (C >> 1));
        // samples (frame n of the lane's row, its pair of channels): n >= 0 in its own row, n < 0 at the end of the row above
        const TIn *const myin = inb + (row + 1) * a.in_stride + ch;
        const int above = a.in_stride - NE;
        struct alignas(2 * sizeof(TIn)) InPair {
            TIn x, y;
        };
        // (results go to global memory: the caller's buffer is aligned to an element, no more)
        struct __attribute__((packed, aligned(sizeof(TOut)))) OutPair {
            TOut x, y;
        };

        // the taps of the output to come, in scalar registers
        const double *trow = a.rtaps + (int64_t)(i0 % up) * TT;  // the output's row of the tap table
        const double *const tend = a.rtaps + (int64_t)up * TT;
        v4d h[NC];
        for_each_const(std::make_integer_sequence<int, NC>{}, [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            h[g] = taps_load<32 * g>(trow);
        });

        // the window: slot r holds frame q with q mod TT == r (q counted from the segment's first frame, n_out - (T - 1))
        double X0[TT], X1[TT];
        const int ns = n_out - H;
        // (its refill, eight frames' reads in flight at a time: all T - 1 at once were 2 T registers beside the window's)
        for_each_const(std::make_integer_sequence<int, (H + 7) / 8>{}, [&](auto bc) {
            constexpr int q0 = decltype(bc)::value * 8;
            constexpr int nq = H - q0 < 8 ? H - q0 : 8;
            InPair fr[nq];
#pragma unroll
            for (int q = 0; q < nq; ++q) {
                const int n = ns + q0 + q;
                fr[q] = *reinterpret_cast<const InPair *>(myin + n * C - (n < 0 ? above : 0));
            }
#pragma unroll
            for (int q = 0; q < nq; ++q) {
                X0[q0 + q] = (double)fr[q].x;
                X1[q0 + q] = (double)fr[q].y;
            }
        });
        X0[H] = X1[H] = 0.0;
        PH_RR_STAMP(2);
        const TIn *src = myin + n_out * C;  // frame n_cur + 1
        InPair nx = *reinterpret_cast<const InPair *>(src);
        int n_cur = n_out - 1;  // the newest frame in the window (row-relative)
        // Results leave from the lane: (row, pair) lanes of one row write C / 2 x 8 adjacent bytes; with two channels a
        // lane's consecutive outputs are adjacent, and two of them leave as one 16-byte store.
        int go = obw + row * row_out + i0;  // the lane's output at hand, relative to the call's first
        TOut *gdst = lout + (int64_t)go * C + ch;
        const int out_frames = lane_on ? (int)a.out_frames : 0;  // (an idle lane's outputs are nobody's)
        OutPair pend{};
        bool have = false;  // (wave-uniform: stereo only)
        for (;;) {
            for_each_const(std::make_integer_sequence<int, TT>{}, [&](auto qc) {
                constexpr int r = (decltype(qc)::value + H) % TT;
                if (i >= i1)
                    return;
                X0[r] = (double)nx.x;
                X1[r] = (double)nx.y;
                src += C;
                nx = *reinterpret_cast<const InPair *>(src);  // (one frame ahead; past the last row's end it reads the pad or the parked outputs: never used)
                ++n_cur;
                while (n_out == n_cur && i < i1) {
                    // ---- one output of every row: x[n - j] sits in slot (r - j) mod TT
                    const double *tn = trow + TT;
                    tn = tn == tend ? a.rtaps : tn;
                    double acc0 = 0.0, acc1 = 0.0;
                    taps_wait(h);
                    for_each_const(std::make_integer_sequence<int, NC>{}, [&](auto gc) {
                        constexpr int g = decltype(gc)::value;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int j = g * 4 + jj;
                            const int sl = (r - j + 2 * TT) % TT;
                            if (PH_RR_ABLATE == 2 && jj > 0)
                                continue;
                            acc0 = __builtin_fma(h[g][jj], X0[sl], acc0);
                            acc1 = __builtin_fma(h[g][jj], X1[sl], acc1);
                        }
                        // these four taps are spent: the next output's take their registers.  (The sums pass through
                        // the ordered asm stream and nothing may cross the barrier: the load stays behind the fma.)
                        asm volatile("" : "+v"(acc0), "+v"(acc1));
                        __builtin_amdgcn_sched_barrier(0);
                        if (PH_RR_ABLATE != 1)
                            h[g] = taps_load<32 * g>(tn);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    OutPair op;
                    op.x = (TOut)acc0;
                    op.y = (TOut)acc1;
                    if (PH_RR_ABLATE == 3) {
                    } else if (C != 2) {
                        if ((unsigned)go < (unsigned)out_frames)
                            *reinterpret_cast<OutPair *>(gdst) = op;
                    } else if (!have) {
                        pend = op;
                        have = true;
                    } else {
                        struct __attribute__((packed, aligned(sizeof(TOut)))) Quad {
                            OutPair a, b;
                        };
                        if ((unsigned)(go - 1) < (unsigned)out_frames && (unsigned)go < (unsigned)out_frames)
                            *reinterpret_cast<Quad *>(gdst - C) = Quad{pend, op};
                        else if ((unsigned)(go - 1) < (unsigned)out_frames)
                            *reinterpret_cast<OutPair *>(gdst - C) = pend;
                        else if ((unsigned)go < (unsigned)out_frames)
                            *reinterpret_cast<OutPair *>(gdst) = op;
                        have = false;
                    }
                    ++go;
                    gdst += C;
                    ++i;
                    trow = tn;
                    tm += dr;
                    n_out += dq;
                    if (tm >= up) {
                        tm -= up;
                        ++n_out;
                    }
                }
            });
            if (i >= i1)
                break;
        }
        // The last output asked for the taps of an output that never comes: those loads are still in flight, and the
        // compiler (which does not know that the asm statements are loads) hands the taps' registers to the code
        // below the moment they are dead -- the late data then lands in an address (seen: writes to read-only
        // pages).  They are waited for while still the taps'.
        taps_wait(h);
        if (have && (unsigned)(go - 1) < (unsigned)out_frames && PH_RR_ABLATE != 3)
            *reinterpret_cast<OutPair *>(gdst - C) = pend;  // (a segment of an odd number of outputs)
        PH_RR_STAMP(3);
    }

    // ---- the next block's input into the other buffer, the one after requested
    if (a.out_off == 0)
        __syncthreads();  // (one buffer: every wave is past its tap loops before the next block takes its place)
    if (PH_RR_ABLATE != 4 && b + (int)gridDim.x < nblocks) {
        deposit(par ? inb0 : inb1);
        if (b + 2 * (int)gridDim.x < nblocks)
            request(b + 2 * (int)gridDim.x);  // flies under the next block's tap loops
    }
    PH_RR_STAMP(5);
    }
#ifdef PH_RR_PROF
    if (a.prof && lane == 0) {
        unsigned long long *dst = a.prof + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 8;
        for (int k = 0; k < 8; ++k)
            dst[k] = rrprof[k];
    }
#endif
}

template <typename TIn, typename TOut>
bool launch_t(const Args &a, hipStream_t s, hipEvent_t ev_a, hipEvent_t ev_b)
{
    const int64_t nblocks = (int64_t)a.lines * a.blocks_per_line;
    const dim3 grid((unsigned)(nblocks < a.max_groups ? nblocks : a.max_groups));
    const dim3 block((unsigned)(64 * a.segs));
    const size_t lds = a.lds_bytes;
#ifdef PH_RR_PROF
    static unsigned long long *prof = nullptr;
    if (!prof && hipMalloc(&prof, sizeof(unsigned long long) * 8 * 65536) != hipSuccess)
        return false;
    const_cast<Args &>(a).prof = prof;
#endif
    // (more than 64 KB of LDS is asked for once per device and instantiation, not once per launch)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return false;
#define PH_RR(TTV)                                                                                            \
    do {                                                                                                      \
        static std::atomic<int> granted[64];                                                                  \
        if ((int)lds > 64 * 1024 && granted[dev].load(std::memory_order_relaxed) < (int)lds) {                \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(resample_rows_kernel<TIn, TOut, TTV>),     \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)      \
                return false;                                                                                 \
            granted[dev].store((int)lds, std::memory_order_relaxed);                                          \
        }                                                                                                     \
        hipExtLaunchKernelGGL((resample_rows_kernel<TIn, TOut, TTV>), grid, block, lds, s, ev_a, ev_b, 0, a); \
    } while (0)
    switch (a.T) {
    case 8: PH_RR(8); break;
    case 12: PH_RR(12); break;
    case 16: PH_RR(16); break;
    case 24: PH_RR(24); break;  // (32 taps: the window alone is 128 registers -- those streams keep the wave kernel)
    default: return false;
    }
#undef PH_RR
#ifdef PH_RR_PROF
    {
        static int launches = 0;
        const size_t nw = (size_t)grid.x * a.segs;
        if (++launches == 15 && nw <= 65536) {
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> h(nw * 8);
            (void)hipMemcpy(h.data(), prof, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
            static const char *names[7] = {"first input (request, deposit)", "barrier ahead of the tap loops", "window refill", "tap loops",
                                           "barrier behind the tap loops", "deposit next, request", "output leaves"};
            double sum[8] = {}, tot = 0, mx = 0;
            for (size_t w = 0; w < nw; ++w) {
                double wt = 0;
                for (int k = 0; k < 7; ++k) {
                    sum[k] += (double)h[w * 8 + k];
                    wt += (double)h[w * 8 + k];
                }
                mx = wt > mx ? wt : mx;
            }
            for (int k = 0; k < 7; ++k)
                tot += sum[k];
            std::fprintf(stderr, "[resampler rows prof] s_memtime ticks per wave: %zu waves in %u workgroups, %lld blocks\n", nw, grid.x, (long long)nblocks);
            for (int k = 0; k < 7; ++k)
                std::fprintf(stderr, "[resampler rows prof]   %-32s %9.1f  %5.1f %%\n", names[k], sum[k] / (double)nw, 100.0 * sum[k] / tot);
            std::fprintf(stderr, "[resampler rows prof]   %-32s %9.1f (slowest wave %9.1f)\n", "total", tot / (double)nw, mx);
            // by the wave's place in its workgroup (= its segment of the rows): refill + tap loops, and the wait behind them
            for (int w = 0; w < a.segs; ++w) {
                double t3 = 0, t4 = 0, t2 = 0;
                for (unsigned g = 0; g < grid.x; ++g) {
                    t2 += (double)h[((size_t)g * a.segs + w) * 8 + 2];
                    t3 += (double)h[((size_t)g * a.segs + w) * 8 + 3];
                    t4 += (double)h[((size_t)g * a.segs + w) * 8 + 4];
                }
                std::fprintf(stderr, "[resampler rows prof]   wave %2d: refill %8.1f tap loops %8.1f barrier behind %8.1f\n", w, t2 / grid.x, t3 / grid.x, t4 / grid.x);
            }
        }
    }
#endif
    return hipGetLastError() == hipSuccess;
}

}  // namespace

bool launch(const Args &a, int in_f64, int out_f64, hipStream_t s, hipEvent_t ev_a, hipEvent_t ev_b)
{
    // (the ABI hands the resampler buffers of ONE sample type: the mixed instantiations would be code no call reaches)
    // (float64 streams: a block's stretch -- 64 rows of a period each, in and out -- is twice a CU's LDS; they keep the
    // wave kernel, and no instantiation is built that no call can reach)
    if (!in_f64 && !out_f64)
        return launch_t<float, float>(a, s, ev_a, ev_b);
    return false;
}

}  // namespace rows
}  // namespace pipehip
