// Microbenchmark: throughput of v_permlane32_swap_b32 / v_permlane16_swap_b32 (gfx950) and a
// correctness print of their lane semantics.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void __launch_bounds__(1024) k(unsigned *out, int iters)
{
    unsigned a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 16 + i;
        b[i] = threadIdx.x * 16 + 8 + i;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) {
                auto r = __builtin_amdgcn_permlane32_swap(a[i], b[i], false, false);
                a[i] = r[0];
                b[i] = r[1];
            } else if (OP == 1) {
                auto r = __builtin_amdgcn_permlane16_swap(a[i], b[i], false, false);
                a[i] = r[0];
                b[i] = r[1];
            } else {
                a[i] = a[i] * 3 + b[i];  // plain 32-bit VALU for reference (v_mad)
            }
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        acc += a[i] ^ b[i];
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

__global__ void show(unsigned *out)
{
    unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + threadIdx.x] = q[0];
    out[192 + threadIdx.x] = q[1];
}

template <int OP>
void run(int wps, const char *name)
{
    unsigned *d;
    hipMalloc(&d, 4 * 256 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 20000;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * wps), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * wps), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-22s waves/SIMD=%d  %.2f cycles per instr (@2.1GHz)\n", name, wps, ms * 1e-3 * 2.1e9 / ((double)wps * iters * 8));
    hipFree(d);
}

int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(show, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *nm[4] = {"permlane32_swap vdst", "permlane32_swap src ", "permlane16_swap vdst", "permlane16_swap src "};
    for (int j = 0; j < 4; ++j) {
        printf("%s:", nm[j]);
        for (int l = 0; l < 64; l += 8)
            printf(" [%d]=%u", l, h[64 * j + l]);
        printf("\n");
    }
    for (int w : {1, 4}) {
        run<0>(w, "v_permlane32_swap_b32");
        run<1>(w, "v_permlane16_swap_b32");
        run<2>(w, "v_mad_u32 (reference)");
    }
    return 0;
}
