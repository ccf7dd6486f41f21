/*
 * A Processor driven the way the reference drives ProcessFunc (pipe.go:423-451), from plain
 * C through the C ABI only: allocator -> StartFunc -> ProcessFunc per buffer (with a short
 * last buffer) -> FlushFunc.  No Python, no torch; links libpipe_hip.so.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/fir_stream.c -Lpipe_amd/lib -lpipe_hip \
 *       -Wl,-rpath,$PWD/pipe_amd/lib -o fir_stream
 *   ./fir_stream out.f64        # writes the float64 output stream, prints a summary
 *
 * Input: the repo's synthetic stream (SplitMix64, seed 0x5EED0000 + Line index), 2 channels,
 * 10 buffers of 512 frames and one of 200.  Taps: 64 x 1/64 (exact in binary).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "pipe_hip.h"

static uint64_t splitmix64_at(uint64_t seed, uint64_t i)
{
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

#define CHECK(call)                                                                     \
    do {                                                                                \
        int st_ = (call);                                                               \
        if (st_ != PIPE_HIP_OK) {                                                       \
            fprintf(stderr, "%s: %s (hipError %d)\n", #call, pipe_hip_strerror(st_),    \
                    pipe_hip_last_hip_error());                                         \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

int main(int argc, char **argv)
{
    enum { F = 512, C = 2, N = 64, FULL = 10, TAIL = 200 };
    double taps[N], in[F * C], out[F * C];
    pipe_hip_config cfg = {0, F, C, PIPE_HIP_F64, 1, 1};
    pipe_hip_processor *fir = NULL;
    int32_t ch = 0, up = 0, down = 0, n = 0;
    uint64_t pos = 0;
    double sum = 0.0;
    FILE *f = argc > 1 ? fopen(argv[1], "wb") : NULL;
    int k, i;

    for (i = 0; i < N; ++i)
        taps[i] = 1.0 / N;
    CHECK(pipe_hip_fir_create(&cfg, taps, N, &fir));          /* ProcessorAllocatorFunc  line.go:26-30 */
    CHECK(pipe_hip_output_properties(fir, &ch, &up, &down));  /* SignalProperties        line.go:38-41 */
    CHECK(pipe_hip_start(fir));                               /* StartFunc               run.go:64-74  */
    for (k = 0; k <= FULL; ++k) {
        const int frames = k < FULL ? F : TAIL;               /* short last buffer       pipe.go:441   */
        for (i = 0; i < frames * C; ++i, ++pos)
            in[i] = (double)(splitmix64_at(0x5EED0000ull, pos) >> 40) * (1.0 / 8388608.0) - 1.0;
        CHECK(pipe_hip_process(fir, in, frames, out, F, &n)); /* ProcessFunc             pipe.go:438   */
        if (n != frames)
            return 2;
        for (i = 0; i < n * ch; ++i)
            sum += out[i];
        if (f)
            fwrite(out, sizeof(double), (size_t)(n * ch), f);
    }
    CHECK(pipe_hip_flush(fir));                               /* FlushFunc               run.go:54-62  */
    pipe_hip_destroy(fir);
    if (f)
        fclose(f);
    printf("channels=%d rate=%d/%d frames=%d sum=%a\n", ch, up, down, FULL * F + TAIL, sum);
    return 0;
}
