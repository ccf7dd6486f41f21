// Throughput of v_mfma_f64_16x16x4_f64 (and 4x4x4) per SIMD by occupancy: cycles per instruction at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void __launch_bounds__(256) k16(double *out, int iters, double a, double b)
{
    v4d acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        acc[c] = v4d{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c)
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c)
        s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 123.0)
        out[threadIdx.x] = s;
}

template <int CHAINS>
__global__ void __launch_bounds__(256) k4(double *out, int iters, double a, double b)
{
    double acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        acc[c] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c)
                acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c)
        s += acc[c];
    if (s == 123.0)
        out[threadIdx.x] = s;
}

template <typename K>
static void timeit(const char *name, K kern, int per_trip, double flops_per_inst)
{
    double *out;
    (void)hipMalloc(&out, 4096);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int iters = 4000;
    std::printf("%-22s", name);
    for (int blocks : {256, 512, 1024, 2048}) {
        kern<<<blocks, 256>>>(out, 10, 1e-30, 1e-30);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        kern<<<blocks, 256>>>(out, iters, 1e-30, 1e-30);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        const double n = (double)iters * per_trip * (blocks / 256);  // instructions per SIMD
        std::printf("  %dw: %6.1f cyc, %5.1f TF/s", blocks / 256, ms * 1e-3 * 2.4e9 / n, n * 1024 * flops_per_inst / ms / 1e9);
    }
    std::printf("\n");
    (void)hipFree(out);
}

int main()
{
    timeit("16x16x4 1 chain", k16<1>, 8, 2048);
    timeit("16x16x4 2 chains", k16<2>, 16, 2048);
    timeit("16x16x4 4 chains", k16<4>, 32, 2048);
    timeit("4x4x4 (4 blocks) 1 ch", k4<1>, 8, 512);
    timeit("4x4x4 (4 blocks) 4 ch", k4<4>, 32, 512);
    return 0;
}
