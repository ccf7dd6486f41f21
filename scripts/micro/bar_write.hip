// Can the CPU store straight into device memory (large BAR)?  hipMalloc / fine-grained device memory,
// a CPU store, a kernel that reads it back.  A platform without the mapping dies with SIGSEGV here.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void readback(const int *p, int *out) { out[0] = p[0] + p[1023]; }

int main()
{
    int *d = nullptr, *o = nullptr;
    for (int mode = 0; mode < 2; ++mode) {
        hipError_t e = mode == 0 ? hipMalloc(&d, 4096) : hipExtMallocWithFlags((void **)&d, 4096, hipDeviceMallocFinegrained);
        std::printf("mode %d alloc: %s\n", mode, hipGetErrorString(e));
        hipHostMalloc(&o, 64);
        hipPointerAttribute_t at;
        hipPointerGetAttributes(&at, d);
        std::printf("  device ptr %p host ptr %p\n", at.devicePointer, at.hostPointer);
        std::fflush(stdout);
        volatile int *h = (volatile int *)d;
        h[0] = 41;
        h[1023] = 1;
        __builtin_ia32_sfence();
        hipLaunchKernelGGL(readback, dim3(1), dim3(1), 0, 0, d, o);
        hipDeviceSynchronize();
        std::printf("  kernel read %d (want 42)\n", o[0]);
        // time 64 KB of CPU stores into it
        static char src[65536];
        char *big = nullptr;
        mode == 0 ? hipMalloc(&big, 65536) : hipExtMallocWithFlags((void **)&big, 65536, hipDeviceMallocFinegrained);
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 100; ++r) {
            std::memcpy(big, src, 65536);
            __builtin_ia32_sfence();
        }
        std::printf("  64 KiB CPU memcpy into it: %.2f us\n",
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 100);
        std::fflush(stdout);
    }
    // 64 MiB from pageable memory: straight into device memory through the BAR, against the usual
    // pageable -> pinned memcpy + hipMemcpy to the device
    {
        hipDeviceProp_t prop;
        hipGetDeviceProperties(&prop, 0);
        std::printf("isLargeBar %d\n", prop.isLargeBar);
        const size_t n = 64u << 20;
        char *src = (char *)malloc(n), *pin = nullptr, *dev = nullptr;
        std::memset(src, 1, n);
        hipHostMalloc(&pin, n);
        hipMalloc(&dev, n);
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            std::memcpy(dev, src, n);
            __builtin_ia32_sfence();
            auto t1 = std::chrono::steady_clock::now();
            std::memcpy(pin, src, n);
            auto t2 = std::chrono::steady_clock::now();
            hipMemcpy(dev, pin, n, hipMemcpyHostToDevice);
            auto t3 = std::chrono::steady_clock::now();
            auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            std::printf("64 MiB: CPU memcpy into device memory %.0f us (%.1f GB/s); memcpy to pinned %.0f us + hipMemcpy H2D %.0f us\n",
                        us(t0, t1), n / us(t0, t1) / 1e3, us(t1, t2), us(t2, t3));
        }
    }
    return 0;
}
