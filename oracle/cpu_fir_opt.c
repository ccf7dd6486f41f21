/*
 * oracle/cpu_fir_opt.c -- TEST INFRASTRUCTURE: an OPTIMISED CPU FIR for bench.py's
 * `cpu_optimized` object ("an optimised CPU variant (AVX, f32) is reported
 * separately and labelled as such").  It is NOT the reference's algorithm and NOT bit-compatible
 * with the oracle: float32 accumulation, channels de-interleaved per block, the frame loop
 * vectorised by the compiler (-O3 -march=native: AVX2 / AVX-512 FMA on the GPU box's host), one
 * thread per slice of the Lines.  It answers "what would a tuned CPU implementation of the same
 * stage do on this host", nothing else.
 *
 *   cpu_fir_opt <lines> <channels> <buffer_frames> <buffers_per_line> <ntaps> <threads>
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int lines, channels, frames, buffers, ntaps, first_line;
    const float *taps;
    double checksum;
} job;

static inline uint64_t splitmix64(uint64_t seed, uint64_t i)
{
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* y[n] = sum_k h[k] x[n - k] for one de-interleaved channel plane; x points at frame 0 of the
 * buffer and has ntaps-1 history frames in front of it */
static void fir_plane(const float *restrict x, float *restrict y, int frames, const float *restrict h, int ntaps)
{
    enum { B = 64 };
    for (int n0 = 0; n0 < frames; n0 += B) {
        float acc[B];
        const int nb = frames - n0 < B ? frames - n0 : B;
        for (int i = 0; i < B; i++)
            acc[i] = 0.f;
        if (nb == B) {
            for (int k = 0; k < ntaps; k++) {
                const float hk = h[k];
                const float *xs = x + n0 - k;
                for (int i = 0; i < B; i++)
                    acc[i] += hk * xs[i];
            }
        } else {
            for (int k = 0; k < ntaps; k++)
                for (int i = 0; i < nb; i++)
                    acc[i] += h[k] * x[n0 - k + i];
        }
        memcpy(y + n0, acc, sizeof(float) * (size_t)nb);
    }
}

static void *run_job(void *arg)
{
    job *j = arg;
    const int C = j->channels, F = j->frames, H = j->ntaps - 1;
    float *in = malloc(sizeof(float) * (size_t)F * C);
    float *out = malloc(sizeof(float) * (size_t)F * C);
    float *plane = malloc(sizeof(float) * (size_t)(F + H) * C * (size_t)j->lines);  /* history + buffer per Line and channel */
    float *yp = malloc(sizeof(float) * (size_t)F);
    memset(plane, 0, sizeof(float) * (size_t)(F + H) * C * (size_t)j->lines);
    double sum = 0;
    for (int b = 0; b < j->buffers; b++) {
        for (int l = 0; l < j->lines; l++) {  /* round robin over the Lines, one buffer each */
            const uint64_t seed = 0x5EED0000ull + (uint64_t)(j->first_line + l);
            const uint64_t base = (uint64_t)b * F * C;
            for (int i = 0; i < F * C; i++)   /* the Source: interleaved synthetic samples */
                in[i] = (float)((double)(splitmix64(seed, base + i) >> 40) * 0x1p-23 - 1.0);
            for (int c = 0; c < C; c++) {
                float *p = plane + ((size_t)l * C + c) * (size_t)(F + H);
                for (int n = 0; n < F; n++)
                    p[H + n] = in[n * C + c];
                fir_plane(p + H, yp, F, j->taps, j->ntaps);
                for (int n = 0; n < F; n++)
                    out[n * C + c] = yp[n];
                memmove(p, p + F, sizeof(float) * (size_t)H);  /* carry the history */
            }
            sum += out[(F - 1) * C];  /* the Sink: keep the result alive */
        }
    }
    j->checksum = sum;
    free(in);
    free(out);
    free(plane);
    free(yp);
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: %s lines channels frames buffers ntaps threads\n", argv[0]);
        return 2;
    }
    int lines = atoi(argv[1]), channels = atoi(argv[2]), frames = atoi(argv[3]);
    int buffers = atoi(argv[4]), ntaps = atoi(argv[5]), threads = atoi(argv[6]);
    if (threads < 1)
        threads = 1;
    if (threads > lines)
        threads = lines;
    float *taps = malloc(sizeof(float) * (size_t)ntaps);
    double sum = 0, *t64 = malloc(sizeof(double) * (size_t)ntaps);
    for (int k = 0; k < ntaps; k++) {
        double m = k - 0.5 * (ntaps - 1);
        double s = fabs(m) < 1e-12 ? 0.5 : sin(M_PI * 0.5 * m) / (M_PI * m);
        double w = ntaps > 1 ? 0.54 - 0.46 * cos(2.0 * M_PI * k / (ntaps - 1)) : 1.0;
        t64[k] = s * w;
        sum += t64[k];
    }
    for (int k = 0; k < ntaps; k++)
        taps[k] = (float)(t64[k] / sum);
    job *jobs = calloc((size_t)threads, sizeof *jobs);
    pthread_t *tid = calloc((size_t)threads, sizeof *tid);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int first = 0;
    for (int t = 0; t < threads; t++) {
        int n = lines / threads + (t < lines % threads ? 1 : 0);
        jobs[t] = (job){n, channels, frames, buffers, ntaps, first, taps, 0};
        first += n;
        if (threads == 1)
            run_job(&jobs[t]);
        else
            pthread_create(&tid[t], NULL, run_job, &jobs[t]);
    }
    double chk = 0;
    for (int t = 0; t < threads; t++) {
        if (threads > 1)
            pthread_join(tid[t], NULL);
        chk += jobs[t].checksum;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    double samples = (double)lines * frames * buffers * channels;
    printf("{\"seconds\": %.6f, \"scalar_samples\": %.0f, \"msamples_per_s\": %.4f, \"threads\": %d, "
           "\"lines\": %d, \"channels\": %d, \"buffer_frames\": %d, \"buffers_per_line\": %d, \"ntaps\": %d, "
           "\"checksum\": %.6f}\n",
           sec, samples, samples / sec / 1e6, threads, lines, channels, frames, buffers, ntaps, chk);
    return 0;
}
