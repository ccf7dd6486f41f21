// What the host link of this box delivers: pinned host memory <-> device memory, one direction alone and both at once,
// by the DMA engines (hipMemcpyAsync on two streams) and by kernels that read / write the pinned memory themselves.
// hipcc --offload-arch=gfx950 -O2 pcie_rates.hip -o pcie_rates
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ void copy16(const v4u *__restrict__ src, v4u *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}
__global__ void copy8(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t bytes = (size_t)64 << 20;
    void *h_a, *h_b, *d_a, *d_b;
    CK(hipHostMalloc(&h_a, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&h_b, bytes, hipHostMallocDefault));
    CK(hipMalloc(&d_a, bytes));
    CK(hipMalloc(&d_b, bytes));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto rate = [&](const char *what, auto fn) {
        for (int i = 0; i < 3; ++i)
            fn();
        CK(hipDeviceSynchronize());
        const int reps = 20;
        const double t0 = now_s();
        for (int i = 0; i < reps; ++i)
            fn();
        CK(hipDeviceSynchronize());
        const double dt = (now_s() - t0) / reps;
        std::printf("%-70s %7.3f ms  %6.1f GB/s each way\n", what, dt * 1e3, bytes / dt / 1e9);
    };
    rate("DMA H2D alone", [&] { CK(hipMemcpyAsync(d_a, h_a, bytes, hipMemcpyHostToDevice, s1)); });
    rate("DMA D2H alone", [&] { CK(hipMemcpyAsync(h_b, d_b, bytes, hipMemcpyDeviceToHost, s2)); });
    rate("DMA H2D + D2H at once", [&] {
        CK(hipMemcpyAsync(d_a, h_a, bytes, hipMemcpyHostToDevice, s1));
        CK(hipMemcpyAsync(h_b, d_b, bytes, hipMemcpyDeviceToHost, s2));
    });
    for (int grid : {64, 256, 1024, 4096}) {
        char name[128];
        std::snprintf(name, sizeof name, "kernel reads pinned (16 B / lane, %d x 256 threads)", grid);
        rate(name, [&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, s1, (const v4u *)h_a, (v4u *)d_a, bytes / 16); });
        std::snprintf(name, sizeof name, "kernel writes pinned (16 B / lane, %d x 256 threads)", grid);
        rate(name, [&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, s2, (const v4u *)d_b, (v4u *)h_b, bytes / 16); });
    }
    rate("kernel reads pinned (8 B / lane, 1024 x 256)", [&] {
        hipLaunchKernelGGL(copy8, dim3(1024), dim3(256), 0, s1, (const unsigned long long *)h_a, (unsigned long long *)d_a, bytes / 8);
    });
    rate("kernel reads + kernel writes pinned at once (16 B, 1024 x 256 each)", [&] {
        hipLaunchKernelGGL(copy16, dim3(1024), dim3(256), 0, s1, (const v4u *)h_a, (v4u *)d_a, bytes / 16);
        hipLaunchKernelGGL(copy16, dim3(1024), dim3(256), 0, s2, (const v4u *)d_b, (v4u *)h_b, bytes / 16);
    });
    rate("DMA H2D + kernel writes pinned at once", [&] {
        CK(hipMemcpyAsync(d_a, h_a, bytes, hipMemcpyHostToDevice, s1));
        hipLaunchKernelGGL(copy16, dim3(1024), dim3(256), 0, s2, (const v4u *)d_b, (v4u *)h_b, bytes / 16);
    });
    rate("kernel reads pinned + DMA D2H at once", [&] {
        hipLaunchKernelGGL(copy16, dim3(1024), dim3(256), 0, s1, (const v4u *)h_a, (v4u *)d_a, bytes / 16);
        CK(hipMemcpyAsync(h_b, d_b, bytes, hipMemcpyDeviceToHost, s2));
    });
    return 0;
}
