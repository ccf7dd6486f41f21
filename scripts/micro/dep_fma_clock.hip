// Microbenchmark (GPU box): what a lone wave gets.  A dependent chain of v_fma_f64 timed with the
// shader clock (clock64) and the constant 100 MHz clock (wall_clock64): cycles per dependent fma and
// the shader clock a sporadic small launch actually runs at (idle -> launch -> idle, as one pipe
// buffer per ProcessFunc call does), against back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dep_fma scripts/micro/dep_fma_clock.hip && /tmp/dep_fma
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>

__global__ void chain(double *out, long long *t, int n, double a, double b)
{
    double x = out[threadIdx.x];
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i)
        x = __builtin_fma(x, a, b);
    const long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) {
        t[0] = c1 - c0;
        t[1] = w1 - w0;
    }
}

__global__ void chain2(double *out, long long *t, int n, double a, double b)
{
    // two dependent fma per step, the biquad's loop-carried path: s -> y -> s
    double s = out[threadIdx.x], y = 0;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        y = __builtin_fma(a, b, s);
        s = __builtin_fma(-a, y, b);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = s + y;
    if (threadIdx.x == 0) {
        t[0] = c1 - c0;
        t[1] = w1 - w0;
    }
}

// the biquad's step (5 float64 ops, 2 of them on the loop-carried path), 16 steps unrolled, inputs in registers
__global__ void bq(double *out, long long *t, int n, double b0, double b1, double b2, double a1, double a2)
{
    double s1 = out[threadIdx.x], s2 = 0, x[16], acc = 0;
    for (int u = 0; u < 16; ++u)
        x[u] = out[(threadIdx.x + u) & 63];
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const double y = __builtin_fma(b0, x[u], s1);
            const double tt = __builtin_fma(b1, x[u], s2);
            s1 = __builtin_fma(-a1, y, tt);
            const double uu = b2 * x[u];
            s2 = __builtin_fma(-a2, y, uu);
            x[u] = y;
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    for (int u = 0; u < 16; ++u)
        acc += x[u];
    out[threadIdx.x] = acc + s1 + s2;
    if (threadIdx.x == 0) {
        t[0] = c1 - c0;
        t[1] = w1 - w0;
    }
}

// the biquad step as the LDS-staged kernel runs it: 2 live lanes of a 256-thread workgroup (the rest
// waits at the barrier), inputs read from / results written to an LDS plane, 16 bytes per access
typedef double f64x2 __attribute__((ext_vector_type(2)));
template <bool LDS_IO, int LIVE>
__global__ void __launch_bounds__(256) bq_lds(double *out, long long *t, int n, double b0, double b1, double b2, double a1, double a2)
{
    __shared__ __attribute__((aligned(16))) double xs[2 * 4100];
    for (int i = threadIdx.x; i < 2 * 4100; i += 256)
        xs[i] = out[i & 63];
    __syncthreads();
    long long c0 = 0, w0 = 0, c1 = 0, w1 = 0;
    double s1 = 0, s2 = 0;
    if (threadIdx.x < LIVE) {
        double *col = xs + (threadIdx.x & 1) * 4100;
        f64x2 xa[8], xb[8];
        for (int u = 0; u < 8; ++u)
            xa[u] = *(const f64x2 *)(col + 2 * u);
        c0 = clock64(), w0 = wall_clock64();
        auto step = [&](double x) {
            const double y = __builtin_fma(b0, x, s1);
            const double tt = __builtin_fma(b1, x, s2);
            s1 = __builtin_fma(-a1, y, tt);
            const double uu = b2 * x;
            s2 = __builtin_fma(-a2, y, uu);
            return y;
        };
        auto run = [&](f64x2 (&x)[8], f64x2 (&nx)[8], int k) {
            double *cur = col + k * 16;
            if (LDS_IO) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    nx[u] = *(const f64x2 *)(cur + 16 + 2 * u);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                f64x2 y;
                y.x = step(x[u].x);
                y.y = step(x[u].y);
                if (LDS_IO) {
                    *(f64x2 *)(cur + 2 * u) = y;
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    x[u] = y;
                }
            }
        };
        for (int k = 0; k + 2 <= n / 16; k += 2) {
            run(xa, xb, k);
            if (LDS_IO)
                run(xb, xa, k + 1);
            else
                run(xa, xb, k + 1);
        }
        c1 = clock64(), w1 = wall_clock64();
        out[threadIdx.x] = s1 + s2 + xa[0].x + xb[0].x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        t[0] = c1 - c0;
        t[1] = w1 - w0;
    }
}

// the biquad step with its input arriving through SGPRs (s_load_dwordx16: 8 frames per scalar instruction, a
// channel per wave, coefficients in VGPRs) and its output leaving through LDS (OUT = 0), global memory
// (OUT = 1) or nowhere (OUT = 2)
typedef const __attribute__((address_space(4))) double *const_f64;
template <int OUT>
__global__ void __launch_bounds__(256) bq_sload(double *out, long long *t, int n, const double *xg, double *yg, double b0, double b1,
                                                double b2, double a1, double a2)
{
    __shared__ __attribute__((aligned(16))) double ys[4100];
    long long c0 = 0, w0 = 0, c1 = 0, w1 = 0;
    double s1 = 0, s2 = 0;
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(a1), "+v"(a2));
    if (threadIdx.x < 64) {
        const_f64 xp = (const_f64)xg;
        double xa[8], xb[8];
        for (int u = 0; u < 8; ++u)
            xa[u] = xp[u];
        c0 = clock64(), w0 = wall_clock64();
        auto step = [&](double x) {
            const double y = __builtin_fma(b0, x, s1);
            const double tt = __builtin_fma(b1, x, s2);
            s1 = __builtin_fma(-a1, y, tt);
            const double uu = b2 * x;
            s2 = __builtin_fma(-a2, y, uu);
            return y;
        };
        auto run = [&](double (&x)[8], double (&nx)[8], int k) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                nx[u] = xp[(k + 1) * 8 + u];
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                f64x2 y;
                y.x = step(x[u]);
                y.y = step(x[u + 1]);
                if (OUT == 0) {
                    if (threadIdx.x == 0)
                        *(f64x2 *)(ys + k * 8 + u) = y;
                } else if (OUT == 1) {
                    if (threadIdx.x == 0)
                        *(f64x2 *)(yg + k * 8 + u) = y;
                } else {
                    s2 += y.x * 1e-300;
                }
            }
        };
        for (int k = 0; k + 2 <= n / 8; k += 2) {
            run(xa, xb, k);
            run(xb, xa, k + 1);
        }
        c1 = clock64(), w1 = wall_clock64();
        out[threadIdx.x] = s1 + s2 + xa[0] + ys[threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        t[0] = c1 - c0;
        t[1] = w1 - w0;
    }
}

// 8 independent fma chains: the issue rate of float64 fma for a lone wave
__global__ void indep(double *out, long long *t, int n, double a, double b)
{
    double x[8];
    for (int u = 0; u < 8; ++u)
        x[u] = out[(threadIdx.x + u) & 63];
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                x[u] = __builtin_fma(x[u], a, b);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double acc = 0;
    for (int u = 0; u < 8; ++u)
        acc += x[u];
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) {
        t[0] = c1 - c0;
        t[1] = w1 - w0;
    }
}

int main()
{
    double *d;
    long long *t, h[2];
    hipMalloc(&d, 256 * 8);
    hipMemset(d, 0, 256 * 8);
    hipMalloc(&t, 16);
    const int n = 4096;
    double *xg, *yg;
    hipMalloc(&xg, 5000 * 8);
    hipMemset(xg, 0, 5000 * 8);
    hipMalloc(&yg, 5000 * 8);
    for (int mode = 0; mode < 12; ++mode) {
        double cyc = 0, wall = 0;
        const int reps = 50;
        for (int r = 0; r < reps; ++r) {
            if (mode == 0)
                std::this_thread::sleep_for(std::chrono::microseconds(300));
            if (mode == 9)
                hipLaunchKernelGGL((bq_sload<0>), dim3(1), dim3(256), 0, 0, d, t, n, xg, yg, 0.2, 0.4, 0.2, -0.5, 0.3);
            else if (mode == 10)
                hipLaunchKernelGGL((bq_sload<1>), dim3(1), dim3(256), 0, 0, d, t, n, xg, yg, 0.2, 0.4, 0.2, -0.5, 0.3);
            else if (mode == 11)
                hipLaunchKernelGGL((bq_sload<2>), dim3(1), dim3(256), 0, 0, d, t, n, xg, yg, 0.2, 0.4, 0.2, -0.5, 0.3);
            else if (mode == 5)
                hipLaunchKernelGGL((bq_lds<false, 64>), dim3(1), dim3(256), 0, 0, d, t, n, 0.2, 0.4, 0.2, -0.5, 0.3);
            else if (mode == 6)
                hipLaunchKernelGGL((bq_lds<false, 2>), dim3(1), dim3(256), 0, 0, d, t, n, 0.2, 0.4, 0.2, -0.5, 0.3);
            else if (mode == 7)
                hipLaunchKernelGGL((bq_lds<true, 64>), dim3(1), dim3(256), 0, 0, d, t, n, 0.2, 0.4, 0.2, -0.5, 0.3);
            else if (mode == 8)
                hipLaunchKernelGGL((bq_lds<true, 2>), dim3(1), dim3(256), 0, 0, d, t, n, 0.2, 0.4, 0.2, -0.5, 0.3);
            else if (mode == 3)
                hipLaunchKernelGGL(bq, dim3(1), dim3(64), 0, 0, d, t, n, 0.2, 0.4, 0.2, -0.5, 0.3);
            else if (mode == 4)
                hipLaunchKernelGGL(indep, dim3(1), dim3(64), 0, 0, d, t, n, 0.999, 1e-3);
            else if (mode == 2)
                hipLaunchKernelGGL(chain2, dim3(1), dim3(64), 0, 0, d, t, n, 0.999, 1e-3);
            else
                hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, d, t, n, 0.999, 1e-3);
            hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            cyc += h[0];
            wall += h[1];
        }
        const char *what = mode == 0   ? "sporadic, 1 fma/step"
                           : mode == 1 ? "back to back, 1 fma/step"
                           : mode == 2 ? "back to back, 2 dependent fma/step"
                           : mode == 3 ? "biquad step x16 unrolled, per step"
                           : mode == 4 ? "8 independent fma per step"
                           : mode == 5 ? "biquad step, wg 256, 64 live lanes, registers"
                           : mode == 6 ? "biquad step, wg 256, 2 live lanes, registers"
                           : mode == 7 ? "biquad step, wg 256, 64 live lanes, LDS in/out"
                           : mode == 8 ? "biquad step, wg 256, 2 live lanes, LDS in/out"
                           : mode == 9 ? "biquad step, x via s_load, y to LDS (lane 0)"
                           : mode == 10 ? "biquad step, x via s_load, y to global (lane 0)"
                                        : "biquad step, x via s_load, y dropped";
        std::printf("%-36s %7.2f shader cycles/step, %7.2f ns/step -> sclk %.0f MHz\n", what, cyc / reps / n,
                    wall / reps / n * 10.0, cyc / wall * 100.0);
    }
    return 0;
}
