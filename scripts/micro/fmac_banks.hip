// Microbenchmark: v_fmac_f64 issue rate vs VGPR bank alignment of (acc, x) operands
// and vs SGPR/VGPR tap operand.  hipcc --offload-arch=gfx950 -O3 fmac_banks.hip -o fmac_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
// 16 independent accumulators; ACC(i), X(i) give the first VGPR of tuple i
#define BODY(ACCBASE, ACCSTRIDE, XBASE, XSTRIDE, TAP)                                      \
    "v_fmac_f64 v[" #ACCBASE "+0*" #ACCSTRIDE ":" #ACCBASE "+0*" #ACCSTRIDE "+1], " TAP ", v[" #XBASE "+0*" #XSTRIDE ":" #XBASE "+0*" #XSTRIDE "+1]\n"

template <int VARIANT>
__global__ void __launch_bounds__(256) k(double *out, int iters)
{
    // all variants: 16 acc tuples + 16 x tuples, explicit registers
    // VARIANT 0: acc v[0:1],v[4:5],... (bank 0,1)  x v[64:65],v[68:69],... (bank 0,1)  -> same banks
    // VARIANT 1: acc v[0:1],v[4:5],... (bank 0,1)  x v[66:67],v[70:71],... (bank 2,3)  -> disjoint banks
    // VARIANT 2: like 0 but tap from VGPR v[130:131]
    // VARIANT 3: like 1 but tap from VGPR v[130:131]
    // VARIANT 4: acc packed v[0:1],v[2:3],v[4:5]...  x packed v[64:65],v[66:67]... (mixed: compiler-like)
    asm volatile(
        "s_mov_b32 s20, 0\n s_mov_b32 s21, 0x3ff00000\n"
        "v_mov_b32 v130, 0\n v_mov_b32 v131, 0x3ff00000\n"
        ::: "s20", "s21", "v130", "v131");
    for (int it = 0; it < iters; ++it) {
        if (VARIANT == 0)
            asm volatile(".rept 8\n"
                ".irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n v_fmac_f64 v[4*\\i:4*\\i+1], s[20:21], v[64+4*\\i:64+4*\\i+1]\n .endr\n"
                ".endr\n" ::: "memory");
        else if (VARIANT == 1)
            asm volatile(".rept 8\n"
                ".irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n v_fmac_f64 v[4*\\i:4*\\i+1], s[20:21], v[66+4*\\i:66+4*\\i+1]\n .endr\n"
                ".endr\n" ::: "memory");
        else if (VARIANT == 2)
            asm volatile(".rept 8\n"
                ".irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n v_fma_f64 v[4*\\i:4*\\i+1], v[130:131], v[64+4*\\i:64+4*\\i+1], v[4*\\i:4*\\i+1]\n .endr\n"
                ".endr\n" ::: "memory");
        else if (VARIANT == 3)
            asm volatile(".rept 8\n"
                ".irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n v_fma_f64 v[4*\\i:4*\\i+1], v[130:131], v[66+4*\\i:66+4*\\i+1], v[4*\\i:4*\\i+1]\n .endr\n"
                ".endr\n" ::: "memory");
        else if (VARIANT == 4)
            asm volatile(".rept 8\n"
                ".irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n v_fmac_f64 v[2*\\i:2*\\i+1], s[20:21], v[64+2*\\i:64+2*\\i+1]\n .endr\n"
                ".endr\n" ::: "memory");
        else if (VARIANT == 5)  // acc at 2 mod 4, x at 0 mod 4, window rotates: x index shifts by one tuple per tap (like the FIR)
            asm volatile(
                ".irp k,0,1,2,3,4,5,6,7\n"
                ".irp i,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n v_fmac_f64 v[2+4*\\i:2+4*\\i+1], s[20:21], v[64+4*((\\i+\\k)%%16):64+4*((\\i+\\k)%%16)+1]\n .endr\n"
                ".endr\n" ::: "memory");
    }
    // registers are asm-owned; nothing meaningful to store, but keep the kernel alive
    if (iters < 0)
        out[threadIdx.x] = 1.0;
}

template <int V>
double run(int waves_per_simd, int iters)
{
    // one workgroup of 256 threads per CU slot: grid = 256 CUs * waves_per_simd
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    double *d;
    hipMalloc(&d, 4096);
    const int grid = 256 * waves_per_simd;
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double fma = (double)grid * 256 * iters * 128.0;  // per thread: 8*16 fma per iter
    hipFree(d);
    return fma * 2 / (ms * 1e-3) / 1e12;
}

int main()
{
    const int iters = 20000;
    for (int w = 1; w <= 2; ++w) {
        printf("waves/SIMD=%d  same-bank sgpr-tap:      %.1f TF\n", w, run<0>(w, iters));
        printf("waves/SIMD=%d  disjoint-bank sgpr-tap:  %.1f TF\n", w, run<1>(w, iters));
        printf("waves/SIMD=%d  same-bank vgpr-tap:      %.1f TF\n", w, run<2>(w, iters));
        printf("waves/SIMD=%d  disjoint-bank vgpr-tap:  %.1f TF\n", w, run<3>(w, iters));
        printf("waves/SIMD=%d  packed (compiler-like):  %.1f TF\n", w, run<4>(w, iters));
        printf("waves/SIMD=%d  acc@2 x@0 rotating:      %.1f TF\n", w, run<5>(w, iters));
    }
    return 0;
}
