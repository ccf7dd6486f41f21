// Microbenchmark: the register-only part of fir_ols (16-point DFTs + twiddle powers) in a loop,
// no memory traffic: what VALU issue rate does this instruction stream reach on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>

struct cd { double re, im; };
__device__ __forceinline__ cd cmul(cd a, cd b)
{
    cd r;
    r.re = __builtin_fma(a.re, b.re, -(a.im * b.im));
    r.im = __builtin_fma(a.re, b.im, a.im * b.re);
    return r;
}
template <int SIGN>
__device__ __forceinline__ void dft4(cd &x0, cd &x1, cd &x2, cd &x3)
{
    const cd s02{x0.re + x2.re, x0.im + x2.im};
    const cd d02{x0.re - x2.re, x0.im - x2.im};
    const cd s13{x1.re + x3.re, x1.im + x3.im};
    const cd d13{x1.re - x3.re, x1.im - x3.im};
    const cd j13 = SIGN < 0 ? cd{d13.im, -d13.re} : cd{-d13.im, d13.re};
    x0 = cd{s02.re + s13.re, s02.im + s13.im};
    x2 = cd{s02.re - s13.re, s02.im - s13.im};
    x1 = cd{d02.re + j13.re, d02.im + j13.im};
    x3 = cd{d02.re - j13.re, d02.im - j13.im};
}

__device__ __forceinline__ cd cmulc(cd a, cd b)
{
    cd r;
    r.re = __builtin_fma(a.re, b.re, a.im * b.im);
    r.im = __builtin_fma(a.im, b.re, -(a.re * b.im));
    return r;
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

// v[k] *= w^k for k = 1..15 from the one per-lane constant w (|w| = 1): no twiddle
// table and no LDS traffic.  Powers are built in groups of four off w, w^2, w^3 so
// that at most five of them are live at once (register pressure), and no power is
// more than five complex multiplications away from w (rounding error).
__device__ __forceinline__ void apply_powers(cd (&v)[16], const cd w)
{
    const cd w2 = cmul(w, w);
    const cd w3 = cmul(w2, w);
    v[1] = cmul(v[1], w);
    v[2] = cmul(v[2], w2);
    v[3] = cmul(v[3], w3);
    cd b = cmul(w2, w2);  // w^4
    v[4] = cmul(v[4], b);
    v[5] = cmul(v[5], cmul(b, w));
    v[6] = cmul(v[6], cmul(b, w2));
    v[7] = cmul(v[7], cmul(b, w3));
    b = cmul(b, b);  // w^8
    v[8] = cmul(v[8], b);
    v[9] = cmul(v[9], cmul(b, w));
    v[10] = cmul(v[10], cmul(b, w2));
    v[11] = cmul(v[11], cmul(b, w3));
    b = cmul(b, cmul(w2, w2));  // w^12
    v[12] = cmul(v[12], b);
    v[13] = cmul(v[13], cmul(b, w));
    v[14] = cmul(v[14], cmul(b, w2));
    v[15] = cmul(v[15], cmul(b, w3));
}


template <int MODE>
__global__ void __launch_bounds__(1024) k(double *out, int iters)
{
    cd v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        v[r] = cd{(double)(threadIdx.x + r), (double)(r * 3 + 1)};
    const double c1 = 0.92387953251128673848, s1 = 0.38268343236508977173;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // radix-4 butterflies only (pure v_add_f64)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                dft4<-1>(v[j], v[j + 4], v[j + 8], v[j + 12]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
                dft4<-1>(v[4 * m], v[4 * m + 1], v[4 * m + 2], v[4 * m + 3]);
        } else if (MODE == 2) {  // one dft16 + one apply_powers, as in fir_ols
            cd w{c1 + 1e-9 * threadIdx.x, s1};
            asm volatile("" : "+v"(w.re), "+v"(w.im));
            dft16<-1>(v);
            apply_powers(v, w);
        } else if (MODE == 3) {  // dft16 only
            dft16<-1>(v);
        } else {          // complex multiplies only
            const cd w{c1, s1};
#pragma unroll
            for (int r = 0; r < 16; ++r)
                v[r] = cmul(v[r], w);
        }
        asm volatile("" : "+v"(v[0].re), "+v"(v[5].im));
    }
    double acc = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        acc += v[r].re + v[r].im;
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int MODE>
void run(int waves_per_simd, int iters, int instr_per_iter, const char *name)
{
    double *d;
    hipMalloc(&d, sizeof(double) * 256 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int threads = 64 * 4 * waves_per_simd;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double per_simd = (double)waves_per_simd * iters * instr_per_iter;
    printf("%-22s waves/SIMD=%d  %.2f cycles per VALU instr (@2.1GHz)\n", name, waves_per_simd, ms * 1e-3 * 2.1e9 / per_simd);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>(w, 20000, 128, "radix-4 adds (128/iter)");
        run<1>(w, 20000, 64, "16 cmul (64/iter)");
        run<3>(w, 20000, 160, "dft16 (160/iter)");
        run<2>(w, 20000, 280, "dft16+powers (280/iter)");
    }
    return 0;
}
