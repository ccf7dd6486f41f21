// Microbenchmark: issue rate of v_add_f64 / v_mul_f64 / v_fma_f64 / v_fmac_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void __launch_bounds__(256) k(double *out, int iters)
{
    asm volatile("v_mov_b32 v130, 0\n v_mov_b32 v131, 0x3ff00000\n" ::: "v130", "v131");
    for (int it = 0; it < iters; ++it) {
        if (OP == 0)
            asm volatile(".rept 8\n .irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n v_add_f64 v[4*\\i:4*\\i+1], v[4*\\i:4*\\i+1], v[66+4*\\i:66+4*\\i+1]\n .endr\n .endr\n" ::: "memory");
        else if (OP == 1)
            asm volatile(".rept 8\n .irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n v_mul_f64 v[4*\\i:4*\\i+1], v[4*\\i:4*\\i+1], v[66+4*\\i:66+4*\\i+1]\n .endr\n .endr\n" ::: "memory");
        else if (OP == 2)
            asm volatile(".rept 8\n .irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n v_fma_f64 v[4*\\i:4*\\i+1], v[130:131], v[66+4*\\i:66+4*\\i+1], v[4*\\i:4*\\i+1]\n .endr\n .endr\n" ::: "memory");
        else if (OP == 3)
            asm volatile(".rept 8\n .irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n v_fmac_f64 v[4*\\i:4*\\i+1], v[130:131], v[66+4*\\i:66+4*\\i+1]\n .endr\n .endr\n" ::: "memory");
        else if (OP == 4)  // add expressed as fma(x, 1.0, y) with an inline constant
            asm volatile(".rept 8\n .irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n v_fma_f64 v[4*\\i:4*\\i+1], v[66+4*\\i:66+4*\\i+1], 1.0, v[4*\\i:4*\\i+1]\n .endr\n .endr\n" ::: "memory");
        else if (OP == 5)  // 32-bit v_mov for reference
            asm volatile(".rept 8\n .irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n v_mov_b32 v[4*\\i], v[66+4*\\i]\n .endr\n .endr\n" ::: "memory");
        else if (OP == 6)
            asm volatile(".rept 8\n .irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n v_cvt_f64_f32 v[4*\\i:4*\\i+1], v[66+4*\\i]\n .endr\n .endr\n" ::: "memory");
        else if (OP == 7)
            asm volatile(".rept 8\n .irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n v_cvt_f32_f64 v[4*\\i], v[66+4*\\i:66+4*\\i+1]\n .endr\n .endr\n" ::: "memory");
    }
    if (iters < 0)
        out[threadIdx.x] = 1.0;
}

template <int OP>
double run(int wps, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    double *d;
    hipMalloc(&d, 4096);
    const int grid = 256 * wps;
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipFree(d);
    // wave-instructions per second per SIMD -> cycles per instruction at ~2.1-2.4 GHz
    const double winstr = (double)grid * 4 /*waves*/ * iters * 128.0;
    const double per_simd_per_s = winstr / (256.0 * 4) / (ms * 1e-3);
    return per_simd_per_s / 1e9;  // G wave-instr / s / SIMD
}

int main()
{
    const int iters = 20000;
    const char *names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_fmac_f64", "v_fma_f64(x,1.0,y)", "v_mov_b32", "v_cvt_f64_f32", "v_cvt_f32_f64"};
    for (int w = 1; w <= 4; w *= 2) {
        double r[8] = {run<0>(w, iters), run<1>(w, iters), run<2>(w, iters), run<3>(w, iters), run<4>(w, iters), run<5>(w, iters), run<6>(w, iters), run<7>(w, iters)};
        for (int i = 0; i < 8; ++i)
            printf("waves/SIMD=%d %-20s %.3f G wave-instr/s/SIMD  (%.2f cycles @2.1GHz)\n", w, names[i], r[i], 2.1 / r[i]);
    }
    return 0;
}
