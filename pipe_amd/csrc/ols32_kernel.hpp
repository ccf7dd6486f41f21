// The kernel of the 32 x 32 overlap-save form (see fir_ols32.hip), with an optional epilogue that
// runs the FIR's output tile through a biquad cascade and a gain before it is stored
// (chain_fused.hip).  Included by both files; each instantiates what it launches.
#pragma once

#include <hip/hip_runtime.h>

#include "common.hpp"
#include "fir_hist.hpp"
#include "ols32_core.hpp"

namespace pipehip {
namespace ols {

constexpr int kWaves32 = 8;             // waves per workgroup = per CU
constexpr unsigned kOut32 = 0x80000000u;  // a buffer offset beyond any num_records: loads 0, stores dropped

struct Args32 {
    int64_t frames;       // frames per Line in this call
    int64_t line_stride;  // elements between Lines
    int C, N, H;
    int L;                // valid outputs per tile
    int pairs, lines, tiles_per_line;
    int ipl, upl;         // items per Line (tiles x pairs); units (item pairs) per Line
    int64_t nunits;
    int d_slot, d_line;   // the wave stride of the launch as (slot, Line) digits
    double *hist_new;
};

// ---- the biquad + gain epilogue of the fused chain (chain_fused.hip) ---------------------------
// Arguments that are the same for every tile of a launch.  Matrices are zero-input state
// transitions of the S-section DF2T cascade (state order s1_0, s2_0, s1_1, s2_1, ...), computed on
// the host in long double.
template <int S>
struct FuseConst {
    double c[S][5];              // {b0, b1, b2, a1, a2} per section
    double gain;
    double A[5][2 * S][2 * S];   // M^(32 * 2^i), i = 0..4: the scan over a tile's 32 segments
    double ML[2 * S][2 * S];     // M^L: one whole tile
    double T32[2 * S][2 * S];    // (M^L)^32: one look-back window
    int has_gain;
    int D;                       // (M^L)^j is below 2^-90 from j = D on (2^30: never within a window)
};
struct FuseArgs {
    int k0, n00;                 // H / 32, H % 32: lane and step of a tile's first output
    unsigned epoch;              // tag of this launch's records (never 0)
    unsigned long long *rec;     // [series][tile][A | P][2 NV] 8-byte {tag, half a double} granules
    const double *state;         // [lines][C][S][2]: the biquad stage's own state, read at a Line's first tile
    double *state_out;           // same shape: the state after the call, written at a Line's last tile
                                 // (a second array: Lines of a few tiles have both in flight at once)
    const double *Tj;            // [33][2S][2S]: (M^L)^j
    const double *Pk;            // [32][2S][2S]: M^(32 k - H) for k > k0
    int *err;                    // set when a bounded spin gives up
};

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

// out = z + m * v  (2S x 2S, m wave-uniform)
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

__device__ __forceinline__ unsigned long long granule_load(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(unsigned long long *p, unsigned tag, unsigned v)
{
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One tile (FIR output in lo/hi, natural layout, re/im = the pair's two channels) through the
// cascade, in place.  Per half-wave = per item:
//   1. transpose to SEGMENT layout through the item's plane: lane k owns window positions
//      [32 k, 32 k + 32) (positions below H -- no output -- are written as zeros);
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
template <int S>
__device__ __forceinline__ void fused_epilogue(cd (&lo)[16], cd (&hi)[16], double *pa, double *pb, const Args32 &a,
                                               const FuseArgs &fa, const FuseConst<S> &fc, int line, int tile, int pair,
                                               bool valid, int l5, int half)
{
    constexpr int N2 = 2 * S;
    constexpr int NV = 2 * N2;  // doubles per record: two channels x 2S states
    const int64_t len64 = a.frames - (int64_t)tile * a.L;
    const int len = (int)(len64 < a.L ? len64 : a.L);  // output frames of this tile (<= 0: none)
    const bool last_tile = tile == a.tiles_per_line - 1;
    valid = valid && len > 0;

    // ---- 1. segment layout, 2. zero-state pass ----------------------------------------------
    // Channel 0's segment stays in registers; channel 1's stays in the plane (it is the last one
    // written there) and is read again in step 5: 64 registers less across the look-back.
    double xr[32];
    double zr[N2], zi[N2];
#pragma unroll
    for (int i = 0; i < N2; ++i)
        zr[i] = zi[i] = 0.0;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            double v = part == 0 ? PH_NAT(r).re : PH_NAT(r).im;
            if (r < fa.k0)
                v = 0.0;
            else if (r == fa.k0)
                v = l5 >= fa.n00 ? v : 0.0;
            PH_COL(r) = v;
        }
        wave_fence();
        if (part == 0) {
#pragma unroll
            for (int c = 0; c < 32; ++c)
                xr[c] = PH_ROW(c);
            wave_fence();
#pragma unroll
            for (int c = 0; c < 32; ++c)
                (void)biquad_step<S>(xr[c], zr, fc);
        } else {
            double xi[32];
#pragma unroll
            for (int c = 0; c < 32; ++c)
                xi[c] = PH_ROW(c);
#pragma unroll
            for (int c = 0; c < 32; ++c)
                (void)biquad_step<S>(xi[c], zi, fc);
        }
    }

    // ---- 3. scan over the half-wave's 32 segments ------------------------------------------------
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int d = 1 << i;
        double tr[N2], ti[N2];
#pragma unroll
        for (int j = 0; j < N2; ++j) {
            tr[j] = __shfl_up(zr[j], d, 32);
            ti[j] = __shfl_up(zi[j], d, 32);
            if (l5 < d) {
                tr[j] = 0.0;
                ti[j] = 0.0;
            }
        }
        affine<S>(zr, zr, fc.A[i], tr);
        affine<S>(zi, zi, fc.A[i], ti);
    }
    // zr / zi: end state of segment l5 for a zero tile start.  Exclusive form and the tile's aggregate:
    double er[N2], ei[N2], Zr[N2], Zi[N2];
#pragma unroll
    for (int j = 0; j < N2; ++j) {
        er[j] = __shfl_up(zr[j], 1, 32);
        ei[j] = __shfl_up(zi[j], 1, 32);
        if (l5 == 0) {
            er[j] = 0.0;
            ei[j] = 0.0;
        }
        Zr[j] = __shfl(zr[j], 31, 32);
        Zi[j] = __shfl(zi[j], 31, 32);
    }

    // ---- 4. the tile's true start state ---------------------------------------------------------
    const int64_t series = (int64_t)line * a.pairs + pair;
    unsigned long long *recs = fa.rec + (series * a.tiles_per_line) * (2 * 2 * NV);  // this series' records
    auto publish = [&](int kind, const double (&vr)[N2], const double (&vi)[N2]) {
        if (l5 == 31 && valid && !last_tile) {
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
    publish(0, Zr, Zi);  // A: the aggregate, before anything is waited for

    double sr[N2], si[N2];  // start state of the tile
#pragma unroll
    for (int j = 0; j < N2; ++j)
        sr[j] = si[j] = 0.0;
    {
        // R = (M^L)^(32 w) after w whole windows
        double R[N2][N2];
#pragma unroll
        for (int i = 0; i < N2; ++i)
#pragma unroll
            for (int j = 0; j < N2; ++j)
                R[i][j] = i == j ? 1.0 : 0.0;
        int base = tile - 1;     // newest predecessor of the current window
        int dist0 = 0;           // its distance from the tile, in tiles, minus one
        bool done = !valid;
        unsigned spins = 0;
        while (!__all(done)) {
            const int u = base - l5;   // the predecessor this lane looks at (-1: the stage's own state)
            int st = 2;                // 2: a P (or nothing to add), 1: an A, 0: not there yet
            if (!done && u >= 0 && dist0 + l5 < fc.D) {
                const unsigned long long *r = recs + (int64_t)u * (2 * 2 * NV);
                const unsigned tp = (unsigned)(granule_load(r + 2 * NV) >> 32);
                const unsigned ta = (unsigned)(granule_load(r) >> 32);
                st = tp == fa.epoch ? 2 : (ta == fa.epoch ? 1 : 0);
            }
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
                if (l5 <= jlim && (st == 1 || (st == 2 && l5 == jp))) {
                    if (u == -1) {
                        // the series starts here: the biquad stage's own state (from the last call)
                        const double *sp = fa.state + ((int64_t)line * a.C + 2 * pair) * N2;
#pragma unroll
                        for (int j = 0; j < N2; ++j) {
                            vr[j] = sp[j];
                            vi[j] = sp[N2 + j];
                        }
                    } else if (u >= 0 && dist0 + l5 < fc.D) {
                        const unsigned long long *r = recs + ((int64_t)u * 2 + (st == 2 ? 1 : 0)) * (2 * NV);
                        double pay[NV];
                        for (unsigned tries = 0;; ++tries) {
                            bool ok = true;
#pragma unroll
                            for (int j = 0; j < NV; ++j) {
                                const unsigned long long g0 = granule_load(r + 2 * j), g1 = granule_load(r + 2 * j + 1);
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
                            vr[j] = pay[j];
                            vi[j] = pay[N2 + j];
                        }
                    }
                    // (M^L)^l5 applied to this predecessor's contribution
                    const double *tj = fa.Tj + (size_t)l5 * N2 * N2;
                    double m[N2][N2];
#pragma unroll
                    for (int i = 0; i < N2; ++i)
#pragma unroll
                        for (int j = 0; j < N2; ++j)
                            m[i][j] = tj[i * N2 + j];
                    double zero[N2];
#pragma unroll
                    for (int j = 0; j < N2; ++j)
                        zero[j] = 0.0;
                    double wr[N2], wi[N2];
                    affine<S>(wr, zero, m, vr);
                    affine<S>(wi, zero, m, vi);
#pragma unroll
                    for (int j = 0; j < N2; ++j) {
                        vr[j] = wr[j];
                        vi[j] = wi[j];
                    }
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
                affine<S>(sr, sr, R, vr);
                affine<S>(si, si, R, vi);
                if (resolved) {
                    done = true;
                } else {
                    double Rn[N2][N2];
#pragma unroll
                    for (int i = 0; i < N2; ++i)
#pragma unroll
                        for (int j = 0; j < N2; ++j) {
                            double acc = 0.0;
#pragma unroll
                            for (int k = 0; k < N2; ++k)
                                acc = __builtin_fma(R[i][k], fc.T32[k][j], acc);
                            Rn[i][j] = acc;
                        }
#pragma unroll
                    for (int i = 0; i < N2; ++i)
#pragma unroll
                        for (int j = 0; j < N2; ++j)
                            R[i][j] = Rn[i][j];
                    base -= 32;
                    dist0 += 32;
                }
            } else if (!done) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1u << 22)) {  // seconds: something is wrong; give up loudly
                    *fa.err = 1;
                    done = true;
                }
            }
        }
    }
    {   // P: the tile's true end state, for the tiles after it
        double pr[N2], pi[N2];
        affine<S>(pr, Zr, fc.ML, sr);
        affine<S>(pi, Zi, fc.ML, si);
        publish(1, pr, pi);
    }

    // ---- 5. the ordered recurrence from the true start states ------------------------------------
    double pk[N2][N2];  // per-lane table entry M^(32 l5 - H)
    {
        const double *src = fa.Pk + (size_t)l5 * N2 * N2;
#pragma unroll
        for (int i = 0; i < N2; ++i)
#pragma unroll
            for (int j = 0; j < N2; ++j)
                pk[i][j] = src[i * N2 + j];
    }
    double str[N2], sti[N2];
    affine<S>(str, er, pk, sr);
    affine<S>(sti, ei, pk, si);
    if (l5 <= fa.k0) {  // lanes before the first output hold zeros; lane k0 gets the tile's start
                        // state injected at its first output (step n00), exactly
#pragma unroll
        for (int j = 0; j < N2; ++j)
            str[j] = sti[j] = 0.0;
    }
    // the Line's last tile leaves the state after its last frame in the biquad stage's own array
    const int plast = a.H + len - 1;
    const bool capture = valid && last_tile;
    const int kl = plast >> 5, jl = plast & 31;
    double cr[N2], ci[N2];
#pragma unroll
    for (int j = 0; j < N2; ++j)
        cr[j] = ci[j] = 0.0;
    const bool any_capture = __any(capture);
    // channel 0 out of registers, channel 1 out of the plane; each result goes back where its
    // input was
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        if (c == fa.n00 && l5 == fa.k0) {
#pragma unroll
            for (int j = 0; j < N2; ++j)
                str[j] = sr[j];
        }
        double yr = biquad_step<S>(xr[c], str, fc);
        if (fc.has_gain)
            yr = yr * fc.gain;
        xr[c] = yr;
        if (any_capture && capture && l5 == kl && c == jl) {
#pragma unroll
            for (int j = 0; j < N2; ++j)
                cr[j] = str[j];
        }
    }
    double xi[32];
#pragma unroll
    for (int c = 0; c < 32; ++c)
        xi[c] = PH_ROW(c);
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        if (c == fa.n00 && l5 == fa.k0) {
#pragma unroll
            for (int j = 0; j < N2; ++j)
                sti[j] = si[j];
        }
        double yi = biquad_step<S>(xi[c], sti, fc);
        if (fc.has_gain)
            yi = yi * fc.gain;
        xi[c] = yi;
        if (any_capture && capture && l5 == kl && c == jl) {
#pragma unroll
            for (int j = 0; j < N2; ++j)
                ci[j] = sti[j];
        }
    }
    if (capture && l5 == kl) {
        double *sp = fa.state_out + ((int64_t)line * a.C + 2 * pair) * N2;
#pragma unroll
        for (int j = 0; j < N2; ++j) {
            sp[j] = cr[j];
            sp[N2 + j] = ci[j];
        }
    }

    // ---- back to natural layout (channel 1 first: the plane is still its) ----------------------
#pragma unroll
    for (int part = 1; part >= 0; --part) {
        if (part == 0)
            wave_fence();
#pragma unroll
        for (int c = 0; c < 32; ++c)
            PH_ROW(c) = part == 0 ? xr[c] : xi[c];
        wave_fence();
        if (part == 0) {
#pragma unroll
            for (int r = 0; r < 32; ++r)
                PH_NAT(r).re = PH_COL(r);
        } else {
#pragma unroll
            for (int r = 0; r < 32; ++r)
                PH_NAT(r).im = PH_COL(r);
        }
    }
    wave_fence();
}

// S = 0: the FIR alone.  S = 1, 2: the FIR's tile goes through an S-section biquad cascade and a
// gain before it is stored (chain_fused.hip; fa / fc are then the epilogue's arguments).
template <typename TIn, typename TOut, int S = 0>
__global__ void __launch_bounds__(kWaves32 * 64)
fir_ols32_kernel(const TIn *__restrict__ in_base, TOut *__restrict__ out_base, const double *__restrict__ hist_base,
                 const double2 *__restrict__ tw_g, const double2 *__restrict__ hperm_g, const Args32 a,
                 const FuseArgs fa, const FuseConst<(S > 0 ? S : 1)> fc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double2 *hspec = reinterpret_cast<double2 *>(smem_raw);       // H[0..512] (+ pad)
    double2 *tws = hspec + kHalf32 + 1;                              // W1024^(k n), k = 1..31, n = 0..31
    double *planes = reinterpret_cast<double *>(tws + 31 * 32);   // [waves][2][kPlane32]

    fir_history_carry(in_base, hist_base, a.hist_new, a.frames, a.line_stride, a.H, a.C, a.lines);
    for (int i = threadIdx.x; i < kHalf32; i += kWaves32 * 64)
        hspec[i] = hperm_g[i];
    for (int i = threadIdx.x; i < 31 * 32; i += kWaves32 * 64)
        tws[i] = tw_g[32 + i];
    __syncthreads();

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, l5 = lane & 31;
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
    const int64_t wave_global = (int64_t)wave_u * nb + xb;
    const int64_t wave_stride = (int64_t)nb * kWaves32;
    int line = 0, slot = 0;
    if (wave_global < a.nunits) {
        line = __builtin_amdgcn_readfirstlane((int)(wave_global / a.upl));
        slot = __builtin_amdgcn_readfirstlane((int)(wave_global % a.upl));
    }
    const unsigned in_step = (unsigned)(32 * a.C * sizeof(TIn));    // 32 frames
    const unsigned out_step = (unsigned)(32 * a.C * sizeof(TOut));
    auto bytes31 = [](int64_t n) { return (int)(n < 0x7FFFFFFF ? n : 0x7FFFFFFF); };
    const int64_t last = a.frames - 1;

    for (int64_t unit = wave_global; unit < a.nunits; unit += wave_stride) {
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
        const int64_t fr00 = (int64_t)tile0 * a.L - a.H;  // first window frame of half 0's item

        cd lo[16], hi[16];
        // ---- the window: lane l5, register r -> window index l5 + 32 r ---------------------------
        if (tile0 > 0) {
            // both windows start inside the Line: one buffer resource based at half 0's window,
            // 32-bit lane offsets; frames past the end of the Line read as zero
            const TIn *base = in_base + (int64_t)line * a.line_stride + fr00 * a.C;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<TIn *>(base), 0, bytes31((a.frames - fr00) * a.C * (int64_t)sizeof(TIn)), 0x00020000);
            const unsigned v0 = valid ? (unsigned)((((tile - tile0) * a.L + l5) * a.C + c0) * (int)sizeof(TIn)) : kOut32;
            In2 pf[32];
#pragma unroll
            for (int r = 0; r < 32; ++r)
                pf[r] = buf_load_pair<TIn>(rs, v0 + (unsigned)r * in_step);
#pragma unroll
            for (int r = 0; r < 32; ++r)
                PH_NAT(r) = cd{(double)pf[r].x, (double)pf[r].y};
        } else {
            // a Line's first tile: its head is the history
            const TIn *__restrict__ in = in_base + (int64_t)line * a.line_stride;
            const double *__restrict__ hist = hist_base + (int64_t)line * a.H * a.C;
            const int64_t fr0 = (int64_t)tile * a.L - a.H;
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
                    } else {
                        re = hist[(g + a.H) * a.C + c0];
                        im = hist[(g + a.H) * a.C + c0 + 1];
                    }
                }
                PH_NAT(r) = cd{re, im};
            }
        }
        // the next unit's coordinates (uniform)
        const int cur_line = line;
        slot += a.d_slot;
        if (slot >= a.upl) {
            slot -= a.upl;
            ++line;
        }
        line += a.d_line;

        ols32_transform(lo, hi, pa, pb, twl, hlo, hhi);
        if constexpr (S > 0)
            fused_epilogue<S>(lo, hi, pa, pb, a, fa, fc, cur_line, tile, c0 >> 1, valid, l5, half);

        // ---- store the valid part: window index i >= H is frame t0 + i - H --------------------
        {
            const int64_t t00 = (int64_t)tile0 * a.L;
            TOut *base = out_base + (int64_t)cur_line * a.line_stride + t00 * a.C;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                base, 0, bytes31((a.frames - t00) * a.C * (int64_t)sizeof(TOut)), 0x00020000);
            const int o0 = (((tile - tile0) * a.L + l5 - a.H) * a.C + c0) * (int)sizeof(TOut);
            const int i0 = valid ? l5 - a.H : -2048;  // window index - H of register 0: outputs need >= 0
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int off = o0 + r * (int)out_step;
                buf_store_pair<TOut>(rs, i0 + 32 * r >= 0 ? (unsigned)off : kOut32, PH_NAT(r).re, PH_NAT(r).im);
            }
        }
    }
}


}  // namespace ols
}  // namespace pipehip
