// Prototype: the bit-exact direct-form FIR on the float64 matrix pipe.  v_mfma_f64_16x16x4_f64 adds its
// four products to the accumulator as an fma chain in k order (scripts/micro/mfma_f64_order.hip), so a
// 16 x 16 tile of outputs D[i][j] = y[T0 + 16 j + i] accumulated over blocks of four inputs, newest
// first, IS the oracle's ordered sum acc = fma(h[k], x[n - k], acc), k = 0 .. N - 1:
//   A_b[i][q] = h[i - 15 + 4 b + q]   (zero outside 0 .. N - 1)      lane (i = l % 16, q = l / 16)
//   B_b[q][j] = x[T0 + 16 j + 15 - 4 b - q]                          lane (j = l % 16, q = l / 16)
// One channel, planar float32 in / out, window and padded taps in LDS.  Checks against the host chain,
// then times a large launch.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/fir_mfma_proto.hip -o /tmp/fir_mfma && /tmp/fir_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kN = 256;                    // taps
constexpr int kBlocks = (kN + 15 + 3) / 4;  // 68
constexpr int kTile = 256;                 // outputs per 16 x 16 tile
constexpr int kTilesPerWave = 2;           // two accumulators in flight
constexpr int kWaves = 4;
constexpr int kWgOut = kTile * kTilesPerWave * kWaves;  // 2048 outputs per workgroup pass
constexpr int kDelta = ((3 - (14 + kN)) % 4 + 4) % 4;  // window shift: every block's four inputs sit inside one group of 16
constexpr int kWinRaw = kWgOut + kN - 1 + 16 + kDelta;
constexpr int kWin = kWinRaw + kWinRaw / 16 + 1;  // one pad double per 16: the 16 columns of a B operand on 16 different banks
__device__ __forceinline__ constexpr int wpos(int w) { return (w + kDelta) + ((w + kDelta) >> 4); }

__global__ void __launch_bounds__(kWaves * 64) fir_mfma(const float *__restrict__ x, float *__restrict__ y, const double *__restrict__ taps,
                                                        long n_out)
{
    // x holds n_out + N - 1 samples: x[N - 1 + n] is the input of output n (the history in front)
    __shared__ double win[kWin];
    __shared__ double hp[kBlocks * 4 + 32];  // hp[t + 15] = h[t], zeros around
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kBlocks * 4 + 32; i += kWaves * 64) {
        const int t = i - 15;
        hp[i] = t >= 0 && t < kN ? taps[t] : 0.0;
    }
    const int q = lane >> 4, j = lane & 15;
    for (long base = (long)blockIdx.x * kWgOut; base < n_out; base += (long)gridDim.x * kWgOut) {
        __syncthreads();
        // window: win[w] = x[base + w] for w in [0, kWgOut + N - 1): output base + o reads win[o .. o + N - 1]
        for (int w = tid; w < kWgOut + kN - 1; w += kWaves * 64)
            win[wpos(w)] = base + w < n_out + kN - 1 ? (double)x[base + w] : 0.0;
        __syncthreads();
        v4d acc[kTilesPerWave];
        const double *bp[kTilesPerWave];
#pragma unroll
        for (int t = 0; t < kTilesPerWave; ++t) {
            acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
            const int o0 = (wave * kTilesPerWave + t) * kTile;  // first output of the tile, relative to base
            // input of (tile, col j, block b, q): output o0 + 16 j + 15 is the newest row; its x is win[o + N - 1]
            // padded position of w = o0 + 16 j + 15 + (N - 1) - q - 4 b: (w + delta) = 16 (o0 / 16 + j) + d, d = 14 + N + delta - q - 4 b,
            // and the four q of a block share floor(d / 16): 17 j - q per lane, the rest a constant per block
            bp[t] = win + 17 * (o0 / 16 + j) - q;
        }
        const double *ap = hp + j + q;  // A lane (i = lane % 16, q): h[i - 15 + 4 b + q] = hp[i + 4 b + q]
#pragma unroll
        for (int b = 0; b < kBlocks; ++b) {
            const double a = ap[4 * b];
#pragma unroll
            for (int t = 0; t < kTilesPerWave; ++t)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bp[t][(14 + kN + kDelta - 4 * b) + ((14 + kN + kDelta - 4 * b) >> 4)], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < kTilesPerWave; ++t) {
            const long o0 = base + (wave * kTilesPerWave + t) * kTile;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long n = o0 + 16 * j + 4 * r + q;
                if (n < n_out)
                    y[n] = (float)acc[t][r];
            }
        }
    }
}

int main()
{
    std::mt19937_64 rng(7);
    std::vector<double> h(kN);
    for (int k = 0; k < kN; ++k)
        h[k] = (float)(std::sin(0.1 * (k + 1)) / (k + 1.0));
    double *dh;
    (void)hipMalloc(&dh, 8 * kN);
    (void)hipMemcpy(dh, h.data(), 8 * kN, hipMemcpyHostToDevice);
    // ---- parity on a small stream
    {
        const long n = 5000;
        std::vector<float> x(n + kN - 1), y(n);
        for (auto &v : x)
            v = (float)((double)(rng() >> 40) * 0x1p-23 - 1.0);
        float *dx, *dy;
        (void)hipMalloc(&dx, 4 * x.size());
        (void)hipMalloc(&dy, 4 * n);
        (void)hipMemcpy(dx, x.data(), 4 * x.size(), hipMemcpyHostToDevice);
        fir_mfma<<<3, kWaves * 64>>>(dx, dy, dh, n);
        (void)hipMemcpy(y.data(), dy, 4 * n, hipMemcpyDeviceToHost);
        long bad = 0;
        for (long i = 0; i < n; ++i) {
            double acc = 0.0;
            for (int k = 0; k < kN; ++k)
                acc = std::fma(h[k], (double)x[kN - 1 + i - k], acc);
            bad += (float)acc != y[i];
        }
        std::printf("parity: %ld of %ld outputs differ from the ordered fma chain\n", bad, n);
        (void)hipFree(dx);
        (void)hipFree(dy);
    }
    // ---- speed
    {
        const long n = 1L << 28;
        float *dx, *dy;
        (void)hipMalloc(&dx, 4 * (n + kN - 1));
        (void)hipMalloc(&dy, 4 * n);
        (void)hipMemset(dx, 0, 4 * (n + kN - 1));
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        for (int grid : {256 * 2, 256 * 4, 256 * 8}) {
            fir_mfma<<<grid, kWaves * 64>>>(dx, dy, dh, n);
            (void)hipEventRecord(a);
            fir_mfma<<<grid, kWaves * 64>>>(dx, dy, dh, n);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, a, b);
            std::printf("grid %5d: %.3f ms for %ld outputs = %.1f Gsamples/s = %.1f TFLOP/s as the direct form counts (2 N per output)\n", grid, ms, n,
                        n / ms / 1e6, 2.0 * kN * n / ms / 1e9);
        }
    }
    return 0;
}
