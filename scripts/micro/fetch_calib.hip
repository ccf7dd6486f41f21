// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns this library
// uses, on KNOWN byte counts (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports 1/2 of the bytes of a
// wide coalesced streaming read (16 B/lane) ... other access widths and WRITE_SIZE are
// uncalibrated: calibrate on a known byte count in your own access pattern").
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- /tmp/fetch_calib      (and a second pass: WRITE_SIZE)
// Every kernel reads (or writes) exactly `bytes` once; the program prints the byte count per kernel,
// scripts/gpu_fetch_calib.sh divides the counters by it.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace pipehip_calib {

// 16 B per lane, one vector per lane, grid as long as the buffer: the gain kernel's pattern
__global__ void read_b128(const float4 *__restrict__ in, float *__restrict__ sink, int64_t n4)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 v = i < n4 ? in[i] : float4{0, 0, 0, 0};
    if (v.x + v.y + v.z + v.w == 12345.678f)
        sink[0] = v.x;
}
// 8 B per lane: the overlap-save FIR's window loads (one channel pair of float32 per lane)
__global__ void read_b64(const float2 *__restrict__ in, float *__restrict__ sink, int64_t n2)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float2 v = i < n2 ? in[i] : float2{0, 0};
    if (v.x + v.y == 12345.678f)
        sink[0] = v.x;
}
// 4 B per lane, a wave's lanes on consecutive words (256 B per wave-load), eight loads in flight a
// workgroup-stride apart: the resampler's (and the direct FIR's) window staging of 2-channel float32
__global__ void read_b32_staging(const float *__restrict__ in, float *__restrict__ sink, int64_t n, int win_words,
                                 int step_words)
{
    // workgroup b stages the window [b * step_words, b * step_words + win_words): consecutive
    // windows overlap by win_words - step_words (the resampler: T + 1 frames of history per tile)
    const int64_t base = (int64_t)blockIdx.x * step_words;
    float acc = 0.f;
    for (int f0 = threadIdx.x; f0 < win_words; f0 += 8 * blockDim.x) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int64_t g = base + f0 + (int64_t)u * blockDim.x;
            g = g < n ? g : n - 1;
            v[u] = f0 + u * (int)blockDim.x < win_words ? in[g] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc += v[u];
    }
    if (acc == 12345.678f)
        sink[0] = acc;
}
__global__ void write_b128(float4 *__restrict__ out, int64_t n4)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4)
        out[i] = float4{1.f, 2.f, 3.f, (float)i};
}
__global__ void write_b64(float2 *__restrict__ out, int64_t n2)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n2)
        out[i] = float2{1.f, (float)i};
}
// the resampler's result stores: a lane writes the two channels of its output frame as two 4-byte
// stores (8 bytes apart per lane)
__global__ void write_b32_pairs(float *__restrict__ out, int64_t n2)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n2) {
        out[2 * i] = 1.f;
        out[2 * i + 1] = (float)i;
    }
}

}  // namespace pipehip_calib

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));            \
            return 1;                                                              \
        }                                                                          \
    } while (0)

int main(int argc, char **argv)
{
    using namespace pipehip_calib;
    const int64_t bytes = (argc > 1 ? std::atoll(argv[1]) : 512) << 20;  // MiB
    const int reps = 4;
    void *buf = nullptr, *sink = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 256));
    CK(hipMemset(buf, 0, bytes));
    const int64_t n = bytes / 4;
    // the resampler at 160/147, 2 channels, 960-output tiles: 882 input frames advance, 908-frame window
    const int step = 882 * 2, win = (882 + 26) * 2;
    const int64_t tiles = n / step - 1;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(read_b128, dim3((unsigned)(n / 4 / 256)), dim3(256), 0, 0, (const float4 *)buf, (float *)sink, n / 4);
        hipLaunchKernelGGL(read_b64, dim3((unsigned)(n / 2 / 256)), dim3(256), 0, 0, (const float2 *)buf, (float *)sink, n / 2);
        hipLaunchKernelGGL(read_b32_staging, dim3((unsigned)tiles), dim3(192), 0, 0, (const float *)buf, (float *)sink, n, win, step);
        // the same staging without overlap: the counter's factor for plain 4-byte-per-lane loads
        hipLaunchKernelGGL(read_b32_staging, dim3((unsigned)(n / step)), dim3(192), 0, 0, (const float *)buf, (float *)sink, n, step, step);
        hipLaunchKernelGGL(write_b128, dim3((unsigned)(n / 4 / 256)), dim3(256), 0, 0, (float4 *)buf, n / 4);
        hipLaunchKernelGGL(write_b64, dim3((unsigned)(n / 2 / 256)), dim3(256), 0, 0, (float2 *)buf, n / 2);
        hipLaunchKernelGGL(write_b32_pairs, dim3((unsigned)(n / 2 / 256)), dim3(256), 0, 0, (float *)buf, n / 2);
        CK(hipDeviceSynchronize());
    }
    // dispatch order within a repetition, with the bytes each one moves
    std::printf("{\"order\": [\"read_b128\", \"read_b64\", \"read_b32_staging(overlap)\", \"read_b32_staging(plain)\", "
                "\"write_b128\", \"write_b64\", \"write_b32_pairs\"], \"bytes\": [%lld, %lld, %lld, %lld, %lld, %lld, %lld], "
                "\"unique_bytes_overlap\": %lld, \"reps\": %d}\n",
                (long long)bytes, (long long)bytes, (long long)(tiles * win * 4), (long long)((n / step) * step * 4),
                (long long)bytes, (long long)bytes, (long long)bytes, (long long)((tiles - 1) * step * 4 + win * 4), reps);
    return 0;
}
