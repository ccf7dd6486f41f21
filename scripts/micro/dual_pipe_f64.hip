// Do float64 VALU instructions and float64 MFMA instructions of different waves on one SIMD run side by
// side, or do they share the multipliers?  Workgroups of 512 threads: waves 0..3 loop over v_fma_f64 (16
// independent per trip), waves 4..7 over v_mfma_f64_16x16x4_f64 (two chains); MODE 1: only the VALU waves
// work, MODE 2: only the MFMA waves, MODE 3: both.  Time per launch with the same trip counts.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512) dual(double *out, int iters_v, int iters_m, double a, double b)
{
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave < 4) {
        if (MODE & 1) {
            double acc[16];
            for (int i = 0; i < 16; ++i)
                acc[i] = i;
            for (int it = 0; it < iters_v; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            }
            for (int i = 0; i < 16; ++i)
                s += acc[i];
        }
    } else if (MODE & 2) {
        v4d c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        for (int it = 0; it < iters_m; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            }
        }
        s = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[1] + c1[2] + c1[3];
    }
    if (s == 123.0)
        out[threadIdx.x] = s;
}

template <int MODE>
static float run(int blocks, int iv, int im)
{
    double *out;
    (void)hipMalloc(&out, 8192);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    dual<MODE><<<blocks, 512>>>(out, 10, 10, 1e-30, 1e-30);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    dual<MODE><<<blocks, 512>>>(out, iv, im, 1e-30, 1e-30);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipFree(out);
    return ms;
}

int main()
{
    for (int blocks : {256, 512}) {
        const int iv = 40000, im = 8000;  // 640 k fma instructions, 64 k mfma instructions per wave
        const float v = run<1>(blocks, iv, im), m = run<2>(blocks, iv, im), both = run<3>(blocks, iv, im);
        const double fv = 2.0 * 64 * 16 * iv * 4 * blocks, fm = 2048.0 * 8 * im * 4 * blocks;
        std::printf("%d workgroups per CU: VALU alone %.3f ms (%.1f TF/s), MFMA alone %.3f ms (%.1f TF/s), both %.3f ms (%.1f TF/s together)\n", blocks / 256, v,
                    fv / v / 1e9, m, fm / m / 1e9, both, (fv + fm) / both / 1e9);
    }
    return 0;
}
