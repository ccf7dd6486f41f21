// What a launch costs before it has done anything: the overlap-save FIR's launch is 15 us for ANY call of up to one
// unit a wave (profiles/r06_fir_small_calls.txt) -- with the 32 x 32 kernel and with the 16 x 16 x 4 kernel, whose
// unit is half as long.  Candidates, timed apart (hipEvents over 200 launches back to back on one stream, and the
// same with a 5 us pause between launches): the grid itself (256 workgroups of 512 lanes), its 159 KB of LDS per
// workgroup, 212 registers a lane, the 24 KB of tables every workgroup copies into LDS, a dependent chain of 2400
// float64 operations on one wave per SIMD (a lone unit).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <chrono>
#include <thread>

template <int MODE>
__global__ void __launch_bounds__(512) k(const double2 *__restrict__ tab, double *out, int ntab, int chain, int busy_blocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2 *t = reinterpret_cast<double2 *>(smem);
    if (MODE >= 1) {  // the tables into LDS, as the FIR kernels do
        for (int i = threadIdx.x; i < ntab; i += 512)
            t[i] = tab[i];
        __syncthreads();
    }
    if (MODE >= 2 && (int)blockIdx.x < busy_blocks) {  // a lone unit: a dependent float64 chain per lane
        double a = t[threadIdx.x & 255].x + 1.0, b = 0.5;
        for (int i = 0; i < chain; ++i) {
            a = __builtin_fma(a, 0.999999, b);
            b = __builtin_fma(b, 0.999999, a);
        }
        if (a == 123.456)
            out[threadIdx.x] = a + b;
    }
}

template <int MODE>
static void run(const char *what, int grid, size_t lds, const double2 *tab, double *out, int chain, int busy, int pause_us)
{
    hipStream_t s;
    (void)hipStreamCreate(&s);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i)
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), lds, s, tab, out, 1505, chain, busy);
    (void)hipStreamSynchronize(s);
    const int n = 200;
    double sum = 0;
    if (pause_us == 0) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < n; ++i)
            hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), lds, s, tab, out, 1505, chain, busy);
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        sum = ms * 1e3 / n;
    } else {
        for (int i = 0; i < n; ++i) {
            hipExtLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), lds, s, e0, e1, 0, tab, out, 1505, chain, busy);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            sum += ms * 1e3 / n;
            std::this_thread::sleep_for(std::chrono::microseconds(pause_us));
        }
    }
    std::printf("%-78s grid %4d lds %6zu: %7.2f us a launch%s\n", what, grid, lds, sum, pause_us ? " (own events, paused between)" : " (back to back)");
    (void)hipStreamDestroy(s);
}

int main()
{
    double2 *tab;
    double *out;
    (void)hipMalloc(&tab, 1505 * sizeof(double2));
    (void)hipMemset(tab, 0, 1505 * sizeof(double2));
    (void)hipMalloc(&out, 4096);
    const size_t big = 159 * 1024, small = 32 * 1024;
    for (int pause : {0, 5}) {
        run<0>("empty kernel", 256, 1024, tab, out, 0, 0, pause);
        run<0>("empty kernel, 159 KB of LDS a workgroup", 256, big, tab, out, 0, 0, pause);
        run<0>("empty kernel, 32 workgroups", 32, big, tab, out, 0, 0, pause);
        run<1>("+ 24 KB of tables into LDS and a barrier", 256, big, tab, out, 0, 0, pause);
        run<1>("+ 24 KB of tables into LDS and a barrier, 32 KB of LDS", 256, small, tab, out, 0, 0, pause);
        run<2>("+ a chain of 2 x 1200 dependent float64 fma, every workgroup", 256, big, tab, out, 1200, 256, pause);
        run<2>("+ a chain of 2 x 1200 dependent float64 fma, 22 workgroups (171 units)", 256, big, tab, out, 1200, 22, pause);
        run<2>("+ the same on a grid of 22 workgroups", 22, big, tab, out, 1200, 22, pause);
        run<2>("+ a chain of 2 x 600 (a 16 x 16 x 4 item), every workgroup", 256, big, tab, out, 600, 256, pause);
    }
    return 0;
}
