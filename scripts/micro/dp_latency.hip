// Microbenchmark: dependent-chain latency of f64 VALU ops on gfx950 (1 wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS>
__global__ void __launch_bounds__(256) k(double *out, int iters)
{
    asm volatile("v_mov_b32 v130, 0\n v_mov_b32 v131, 0x3ff00000\n" ::: "v130", "v131");
    for (int it = 0; it < iters; ++it) {
        if (CHAINS == 1)
            asm volatile(".rept 128\n v_fma_f64 v[0:1], v[0:1], v[130:131], v[130:131]\n .endr\n" ::: "memory");
        else if (CHAINS == 2)
            asm volatile(".rept 64\n v_fma_f64 v[0:1], v[0:1], v[130:131], v[130:131]\n v_fma_f64 v[4:5], v[4:5], v[130:131], v[130:131]\n .endr\n" ::: "memory");
        else if (CHAINS == 3)
            asm volatile(".rept 42\n v_fma_f64 v[0:1], v[0:1], v[130:131], v[130:131]\n v_fma_f64 v[4:5], v[4:5], v[130:131], v[130:131]\n v_fma_f64 v[8:9], v[8:9], v[130:131], v[130:131]\n .endr\n v_fma_f64 v[0:1], v[0:1], v[130:131], v[130:131]\n v_fma_f64 v[4:5], v[4:5], v[130:131], v[130:131]\n" ::: "memory");
        else if (CHAINS == 4)
            asm volatile(".rept 32\n v_fma_f64 v[0:1], v[0:1], v[130:131], v[130:131]\n v_fma_f64 v[4:5], v[4:5], v[130:131], v[130:131]\n v_fma_f64 v[8:9], v[8:9], v[130:131], v[130:131]\n v_fma_f64 v[12:13], v[12:13], v[130:131], v[130:131]\n .endr\n" ::: "memory");
        else if (CHAINS == 10)  // add chain
            asm volatile(".rept 128\n v_add_f64 v[0:1], v[0:1], v[130:131]\n .endr\n" ::: "memory");
        else if (CHAINS == 11)  // ds_write_b128 issue cost (one wave)
            asm volatile("v_mov_b32 v20, 0\n .rept 128\n ds_write_b128 v20, v[0:3]\n .endr\n s_waitcnt lgkmcnt(0)\n" ::: "memory", "v20");
        else if (CHAINS == 12)
            asm volatile("v_mov_b32 v20, 0\n .rept 128\n ds_read_b128 v[4:7], v20\n .endr\n s_waitcnt lgkmcnt(0)\n" ::: "memory", "v20");
    }
    if (iters < 0)
        out[threadIdx.x] = 1.0;
}

template <int C>
double cyc(int wps, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    double *d;
    hipMalloc(&d, 4096);
    const int grid = 256 * wps;
    hipLaunchKernelGGL(k<C>, dim3(grid), dim3(256), C >= 11 ? 4096 : 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<C>, dim3(grid), dim3(256), C >= 11 ? 4096 : 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipFree(d);
    return (ms * 1e-3) * 2.2e9 / ((double)iters * 128.0) * 1.0;  // cycles per instruction per wave @2.2GHz
}

int main()
{
    const int it = 20000;
    printf("1 wave/SIMD, cycles per instruction (@2.2 GHz assumed):\n");
    printf("  fma f64, 1 dependent chain : %.2f\n", cyc<1>(1, it));
    printf("  fma f64, 2 chains          : %.2f\n", cyc<2>(1, it));
    printf("  fma f64, 3 chains          : %.2f\n", cyc<3>(1, it));
    printf("  fma f64, 4 chains          : %.2f\n", cyc<4>(1, it));
    printf("  add f64, 1 dependent chain : %.2f\n", cyc<10>(1, it));
    printf("  ds_write_b128 (1 wave/SIMD): %.2f   (2 waves/SIMD: %.2f per wave-instr)\n", cyc<11>(1, it), cyc<11>(2, it));
    printf("  ds_read_b128  (1 wave/SIMD): %.2f   (2 waves/SIMD: %.2f per wave-instr)\n", cyc<12>(1, it), cyc<12>(2, it));
    return 0;
}
