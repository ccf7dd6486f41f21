// Microbenchmark: LDS instruction throughput per CU on gfx950 (8 waves per CU, one workgroup
// per CU), lane-consecutive and strided address patterns.  Prints CU-cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>

// OP: 0 ds_write_b128, 1 ds_write_b64, 2 ds_write2_b64, 3 ds_write_b32,
//     4 ds_read_b128, 5 ds_read_b64, 6 ds_read2_b64, 7 ds_read_b32
template <int OP>
__global__ void __launch_bounds__(512) k(float *out, int iters, int lane_stride_bytes, int wave_bytes)
{
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned addr = (unsigned)(wave * wave_bytes + lane * lane_stride_bytes);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0)
            asm volatile(".rept 8\n ds_write_b128 %0, v[100:103]\n .endr\n s_waitcnt lgkmcnt(0)" ::"v"(addr) : "memory", "v100", "v101", "v102", "v103");
        else if (OP == 1)
            asm volatile(".rept 8\n ds_write_b64 %0, v[100:101]\n .endr\n s_waitcnt lgkmcnt(0)" ::"v"(addr) : "memory", "v100", "v101");
        else if (OP == 2)
            asm volatile(".rept 8\n ds_write2_b64 %0, v[100:101], v[102:103] offset0:0 offset1:65\n .endr\n s_waitcnt lgkmcnt(0)" ::"v"(addr) : "memory", "v100", "v101", "v102", "v103");
        else if (OP == 3)
            asm volatile(".rept 8\n ds_write_b32 %0, v100\n .endr\n s_waitcnt lgkmcnt(0)" ::"v"(addr) : "memory", "v100");
        else if (OP == 4)
            asm volatile(".rept 8\n ds_read_b128 v[100:103], %0\n .endr\n s_waitcnt lgkmcnt(0)" ::"v"(addr) : "memory", "v100", "v101", "v102", "v103");
        else if (OP == 5)
            asm volatile(".rept 8\n ds_read_b64 v[100:101], %0\n .endr\n s_waitcnt lgkmcnt(0)" ::"v"(addr) : "memory", "v100", "v101");
        else if (OP == 6)
            asm volatile(".rept 8\n ds_read2_b64 v[100:103], %0 offset0:0 offset1:65\n .endr\n s_waitcnt lgkmcnt(0)" ::"v"(addr) : "memory", "v100", "v101", "v102", "v103");
        else if (OP == 7)
            asm volatile(".rept 8\n ds_read_b32 v100, %0\n .endr\n s_waitcnt lgkmcnt(0)" ::"v"(addr) : "memory", "v100");
    }
    if (iters < 0)
        out[threadIdx.x] = acc + smem[0];
}

template <int OP>
double run(int iters, int lane_stride, int waves)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float *d;
    hipMalloc(&d, 4096);
    const int wave_bytes = 64 * lane_stride > 16384 ? 16384 : 64 * lane_stride;
    const size_t lds = 160 * 1024 - 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), lds, 0, d, 10, lane_stride, wave_bytes);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), lds, 0, d, iters, lane_stride, wave_bytes);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipFree(d);
    const double instr_per_cu = (double)waves * iters * 8.0;
    return ms * 1e-3 * 2.1e9 / instr_per_cu;  // CU-cycles per wave-instruction at 2.1 GHz
}

int main()
{
    const int iters = 20000;
    const char *names[] = {"ds_write_b128", "ds_write_b64", "ds_write2_b64", "ds_write_b32",
                           "ds_read_b128", "ds_read_b64", "ds_read2_b64", "ds_read_b32"};
    const int strides[] = {16, 8, 8, 4, 16, 8, 8, 4};
    for (int waves : {1, 8}) {
        double r[8] = {run<0>(iters, strides[0], waves), run<1>(iters, strides[1], waves), run<2>(iters, strides[2], waves),
                       run<3>(iters, strides[3], waves), run<4>(iters, strides[4], waves), run<5>(iters, strides[5], waves),
                       run<6>(iters, strides[6], waves), run<7>(iters, strides[7], waves)};
        for (int i = 0; i < 8; ++i)
            printf("waves/CU=%d %-14s lane-consecutive: %.2f CU-cycles/instr (@2.1GHz)\n", waves, names[i], r[i]);
    }
    // 16-byte accesses at the exchange strides of fir_ols (65 and 17 double2 = 1040 / 272 bytes)
    for (int st : {1040, 272, 528}) {
        printf("stride %4d B: ds_write_b128 %.2f  ds_read_b128 %.2f  ds_write_b64 %.2f ds_read_b64 %.2f\n", st,
               run<0>(iters, st, 8), run<4>(iters, st, 8), run<1>(iters, st, 8), run<5>(iters, st, 8));
    }
    return 0;
}
