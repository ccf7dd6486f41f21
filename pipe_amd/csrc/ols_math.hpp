// Device-side building blocks shared by the overlap-save kernels (fir_ols.hip: one 1024-point
// transform per wave as 16 x 16 x 4; fir_ols32.hip: one per half-wave as 32 x 32): complex
// float64 arithmetic, in-register 4- and 16-point DFTs, LDS twiddle application, channel-pair
// access through buffer resources.
#pragma once

#include <hip/hip_runtime.h>

namespace pipehip {
namespace ols {

constexpr double kPi = 3.14159265358979323846264338327950288;

struct cd {
    double re, im;
};

__device__ __forceinline__ cd cmul(cd a, cd b)
{
    cd r;
    r.re = __builtin_fma(a.re, b.re, -(a.im * b.im));
    r.im = __builtin_fma(a.re, b.im, a.im * b.re);
    return r;
}
__device__ __forceinline__ cd cmulc(cd a, cd b)  // a * conj(b)
{
    cd r;
    r.re = __builtin_fma(a.re, b.re, a.im * b.im);
    r.im = __builtin_fma(a.im, b.re, -(a.re * b.im));
    return r;
}

// 4-point DFT, SIGN = -1 forward (W4 = -i), +1 inverse (W4 = +i)
template <int SIGN>
__device__ __forceinline__ void dft4(cd &x0, cd &x1, cd &x2, cd &x3)
{
    const cd s02{x0.re + x2.re, x0.im + x2.im};
    const cd d02{x0.re - x2.re, x0.im - x2.im};
    const cd s13{x1.re + x3.re, x1.im + x3.im};
    const cd d13{x1.re - x3.re, x1.im - x3.im};
    // SIGN * i * d13
    const cd j13 = SIGN < 0 ? cd{d13.im, -d13.re} : cd{-d13.im, d13.re};
    x0 = cd{s02.re + s13.re, s02.im + s13.im};
    x2 = cd{s02.re - s13.re, s02.im - s13.im};
    x1 = cd{d02.re + j13.re, d02.im + j13.im};
    x3 = cd{d02.re - j13.re, d02.im - j13.im};
}

// multiply by W16^e (forward) or its conjugate (inverse), e compile-time
template <int SIGN, int E>
__device__ __forceinline__ cd tw16(cd v)
{
    constexpr int e = ((E % 16) + 16) % 16;
    if constexpr (e == 0) {
        return v;
    } else if constexpr (e == 4) {  // -i (fwd)
        return SIGN < 0 ? cd{v.im, -v.re} : cd{-v.im, v.re};
    } else if constexpr (e == 8) {
        return cd{-v.re, -v.im};
    } else if constexpr (e == 12) {
        return SIGN < 0 ? cd{-v.im, v.re} : cd{v.im, -v.re};
    } else {
        constexpr double c = e == 1   ? 0.92387953251128673848
                             : e == 2 ? 0.70710678118654752440
                             : e == 3 ? 0.38268343236508977173
                             : e == 6 ? -0.70710678118654752440
                             : e == 9 ? -0.92387953251128673848
                                      : 0.0;
        constexpr double s = e == 1   ? 0.38268343236508977173
                             : e == 2 ? 0.70710678118654752440
                             : e == 3 ? 0.92387953251128673848
                             : e == 6 ? 0.70710678118654752440
                             : e == 9 ? -0.38268343236508977173
                                      : 0.0;
        // W16^e = c - i*s (forward); conj for inverse
        const cd w{c, SIGN < 0 ? -s : s};
        return cmul(v, w);
    }
}

// 16-point DFT in place: input v[n], output v[k]   (n = j + 4i, k = m + 4p)
template <int SIGN>
__device__ __forceinline__ void dft16(cd (&v)[16])
{
    // stage 1: 4-point DFTs over i for each j  -> t[j][m] stored at v[j + 4m]
#pragma unroll
    for (int j = 0; j < 4; ++j)
        dft4<SIGN>(v[j], v[j + 4], v[j + 8], v[j + 12]);
    // twiddle t[j][m] *= W16^(j*m)
    v[1 + 4 * 1] = tw16<SIGN, 1>(v[1 + 4 * 1]);
    v[1 + 4 * 2] = tw16<SIGN, 2>(v[1 + 4 * 2]);
    v[1 + 4 * 3] = tw16<SIGN, 3>(v[1 + 4 * 3]);
    v[2 + 4 * 1] = tw16<SIGN, 2>(v[2 + 4 * 1]);
    v[2 + 4 * 2] = tw16<SIGN, 4>(v[2 + 4 * 2]);
    v[2 + 4 * 3] = tw16<SIGN, 6>(v[2 + 4 * 3]);
    v[3 + 4 * 1] = tw16<SIGN, 3>(v[3 + 4 * 1]);
    v[3 + 4 * 2] = tw16<SIGN, 6>(v[3 + 4 * 2]);
    v[3 + 4 * 3] = tw16<SIGN, 9>(v[3 + 4 * 3]);
    // stage 2: 4-point DFTs over j for each m: inputs v[j + 4m], outputs X[m + 4p]
    // in place the result p lands at v[p + 4m]; transpose to k = m + 4p below
#pragma unroll
    for (int m = 0; m < 4; ++m)
        dft4<SIGN>(v[0 + 4 * m], v[1 + 4 * m], v[2 + 4 * m], v[3 + 4 * m]);
    // v[p + 4m] holds X[m + 4p]: swap (p,m) <-> (m,p)
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int m = p + 1; m < 4; ++m) {
            const cd t = v[p + 4 * m];
            v[p + 4 * m] = v[m + 4 * p];
            v[m + 4 * p] = t;
        }
}

// v[k] *= t[k * stride] (CONJ: its conjugate) for k = 1..15, t a row set of an LDS twiddle
// table (exactly rounded entries: no power chains).  The reads run G entries ahead of their
// use and the scheduler may not move them further up: fifteen in flight would cost 60 VGPRs and
// with them the fourth wave per SIMD.
template <bool CONJ>
__device__ __forceinline__ void twiddle(cd (&v)[16], const double2 *__restrict__ t, int stride)
{
    constexpr int G = 5;
    double2 w[2][G];
    /* no leading barrier: the first reads may start under the preceding DFT */
#pragma unroll
    for (int j = 0; j < G; ++j)
        w[0][j] = t[(1 + j) * stride];
#pragma unroll
    for (int g = 0; g < 15 / G; ++g) {
        if (g + 1 < 15 / G) {
#pragma unroll
            for (int j = 0; j < G; ++j)
                w[(g + 1) & 1][j] = t[(1 + G * (g + 1) + j) * stride];
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int k = 1 + G * g + j;
            const cd ww{w[g & 1][j].x, w[g & 1][j].y};
            v[k] = CONJ ? cmulc(v[k], ww) : cmul(v[k], ww);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <typename T>
struct Pair;
template <>
struct Pair<float> {
    using type = float2;
};
template <>
struct Pair<double> {
    using type = double2;
};

// channel-pair access through a buffer resource (offset beyond num_records: loads give 0,
// stores are dropped)
template <typename T>
__device__ __forceinline__ typename Pair<T>::type buf_load_pair(__amdgpu_buffer_rsrc_t r, unsigned voff);
template <>
__device__ __forceinline__ float2 buf_load_pair<float>(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0));
}
template <>
__device__ __forceinline__ double2 buf_load_pair<double>(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
template <typename T>
__device__ __forceinline__ void buf_store_pair(__amdgpu_buffer_rsrc_t r, unsigned voff, double re, double im);
template <>
__device__ __forceinline__ void buf_store_pair<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, double re, double im)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const float2 o{(float)re, (float)im};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o), r, voff, 0, 0);
}
template <>
__device__ __forceinline__ void buf_store_pair<double>(__amdgpu_buffer_rsrc_t r, unsigned voff, double re, double im)
{
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const double2 o{re, im};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), r, voff, 0, 0);
}

// one element
template <typename T>
__device__ __forceinline__ T buf_load_one(__amdgpu_buffer_rsrc_t r, unsigned voff);
template <>
__device__ __forceinline__ float buf_load_one<float>(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
template <>
__device__ __forceinline__ double buf_load_one<double>(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0));
}
// one element (the lone channel of an odd channel count)
template <typename T>
__device__ __forceinline__ void buf_store_one(__amdgpu_buffer_rsrc_t r, unsigned voff, double v);
template <>
__device__ __forceinline__ void buf_store_one<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, double v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v), r, voff, 0, 0);
}
template <>
__device__ __forceinline__ void buf_store_one<double>(__amdgpu_buffer_rsrc_t r, unsigned voff, double v)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, voff, 0, 0);
}

// Lanes of one wave talk through the wave-private buffer.  The hardware keeps a
// wave's LDS operations in order, but the compiler reasons per thread and would
// happily move a read above a write to a "different" address: fence every phase.
__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace ols
}  // namespace pipehip
