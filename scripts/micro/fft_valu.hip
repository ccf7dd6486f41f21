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
    }
    return 0;
}
