/*
 * What one ProcessFunc call costs through the C ABI (the cost a cgo shim adds is ~0.1 us on top): one
 * 4096 x 2 pipe buffer per pipe_hip_process, float32 and float64 buffers, for a gain, a 256-tap FIR and a
 * FIR -> gain chain, a biquad and FIR -> biquad -> gain (float64 buffers also with PIPE_HIP_PARAM_RELAXED_F64) -- on the
 * plain path (launch + completion word per call; PIPE_HIP_COMPLETION_EVENT=1 in the environment: a completion event,
 * round 4's plain path) and with PIPE_HIP_PARAM_RESIDENT (the next buffer's work queued on the device behind the
 * device's doorbell).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/percall_latency.c -Lpipe_amd/lib -lpipe_hip -lm \
 *       -Wl,-rpath,$PWD/pipe_amd/lib -o percall_latency && ./percall_latency [calls]
 *
 * Prints one JSON object per (stage, dtype): median / mean / p99 microseconds per call for both paths, and
 * whether the two paths' outputs were identical over the run.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "pipe_hip.h"

enum { F = 4096, C = 2, N = 256 };

static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static int cmp(const void *a, const void *b)
{
    const double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : x > y;
}
#define CHECK(call)                                                                                        \
    do {                                                                                                   \
        int st_ = (call);                                                                                  \
        if (st_ != PIPE_HIP_OK) {                                                                          \
            fprintf(stderr, "%s: %s (hipError %d)\n", #call, pipe_hip_strerror(st_), pipe_hip_last_hip_error()); \
            exit(1);                                                                                       \
        }                                                                                                  \
    } while (0)

static pipe_hip_processor *make(int kind, const pipe_hip_config *cfg, const double *taps)
{
    pipe_hip_processor *p = NULL, *st[2];
    if (kind >= 5) { /* 5, 6: kinds 3, 4 with PIPE_HIP_PARAM_RELAXED_F64 (float64 buffers through the tile biquad) */
        const double one = 1.0;
        p = make(kind - 2, cfg, taps);
        CHECK(pipe_hip_set_param(p, PIPE_HIP_PARAM_RELAXED_F64, &one, 1));
        return p;
    }
    if (kind == 0) {
        CHECK(pipe_hip_gain_create(cfg, 0.5, &p));
    } else if (kind == 1) {
        CHECK(pipe_hip_fir_create(cfg, taps, N, &p));
    } else if (kind == 2) {
        CHECK(pipe_hip_fir_create(cfg, taps, N, &st[0]));
        CHECK(pipe_hip_gain_create(cfg, 0.5, &st[1]));
        CHECK(pipe_hip_chain_create(st, 2, &p));
    } else {
        /* RBJ low-pass, 1 kHz at 48 kHz, Q = 1/sqrt(2): {b0, b1, b2, a1, a2} */
        const double w0 = 2 * 3.14159265358979323846 * 1000.0 / 48000.0, al = sin(w0) / (2 * 0.7071067811865476), a0 = 1 + al;
        const double q[5] = {(1 - cos(w0)) / 2 / a0, (1 - cos(w0)) / a0, (1 - cos(w0)) / 2 / a0, -2 * cos(w0) / a0, (1 - al) / a0};
        if (kind == 3) {
            CHECK(pipe_hip_biquad_create(cfg, q, 1, &p));
        } else {
            pipe_hip_processor *s3[3];
            CHECK(pipe_hip_fir_create(cfg, taps, N, &s3[0]));
            CHECK(pipe_hip_biquad_create(cfg, q, 1, &s3[1]));
            CHECK(pipe_hip_gain_create(cfg, 0.5, &s3[2]));
            CHECK(pipe_hip_chain_create(s3, 3, &p));
        }
    }
    CHECK(pipe_hip_start(p));
    return p;
}

int main(int argc, char **argv)
{
    const int calls = argc > 1 ? atoi(argv[1]) : 3000, warm = 200;
    static const char *names[7] = {"gain", "fir256", "chain fir256+gain", "biquad", "chain fir256+biquad+gain",
                                   "biquad relaxed_f64", "chain fir256+biquad+gain relaxed_f64"};
    double taps[N], sum = 0;
    for (int k = 0; k < N; ++k) {  /* a windowed sinc, normalised */
        const double t = k - (N - 1) / 2.0, w = 0.5 - 0.5 * cos(2 * 3.14159265358979323846 * k / (N - 1));
        taps[k] = (t == 0 ? 1.0 : sin(0.4 * t) / (0.4 * t)) * w;
        sum += taps[k];
    }
    for (int k = 0; k < N; ++k)
        taps[k] /= sum;
    double *lat = malloc(sizeof(double) * (size_t)calls);
    for (int dt = 0; dt < 2; ++dt) {
        const size_t es = dt == 0 ? 4 : 8;
        pipe_hip_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.device = 0;
        cfg.buffer_size = F;
        cfg.channels = C;
        cfg.dtype = dt == 0 ? PIPE_HIP_F32 : PIPE_HIP_F64;
        cfg.lines = 1;
        cfg.max_batch = 1;
        void *in = malloc(es * F * C), *out_a = malloc(es * F * C), *out_b = malloc(es * F * C);
        for (int kind = 0; kind < (dt == 0 ? 5 : 7); ++kind) {
            pipe_hip_processor *plain = make(kind, &cfg, taps), *res = make(kind, &cfg, taps);
            const double one = 1.0;
            CHECK(pipe_hip_set_param(res, PIPE_HIP_PARAM_RESIDENT, &one, 1)); /* (the only handle that asks: never EBUSY) */
            double stats[2][3];
            int same = 1;
            for (int path = 0; path < 2; ++path) {
                pipe_hip_processor *p = path == 0 ? plain : res;
                for (int i = 0; i < calls + warm; ++i) {
                    for (int j = 0; j < F * C; ++j) {  /* a new buffer every call */
                        const double v = sin(0.001 * (double)((i * 131 + j) % 100003));
                        if (dt == 0)
                            ((float *)in)[j] = (float)v;
                        else
                            ((double *)in)[j] = v;
                    }
                    int32_t m = 0;
                    const double t0 = now_us();
                    CHECK(pipe_hip_process(p, in, F, path == 0 ? out_a : out_b, F, &m));
                    const double t1 = now_us();
                    if (i >= warm)
                        lat[i - warm] = t1 - t0;
                    if (m != F)
                        same = 0;
                }
                double mean = 0;
                for (int i = 0; i < calls; ++i)
                    mean += lat[i];
                qsort(lat, (size_t)calls, sizeof(double), cmp);
                stats[path][0] = lat[calls / 2];
                stats[path][1] = mean / calls;
                stats[path][2] = lat[(int)(calls * 0.99)];
            }
            /* both handles have seen the same stream: their last outputs must agree bit for bit */
            if (memcmp(out_a, out_b, es * F * C) != 0)
                same = 0;
            printf("{\"stage\": \"%s\", \"io\": \"%s\", \"frames\": %d, \"channels\": %d, \"calls\": %d, "
                   "\"plain_us\": {\"median\": %.2f, \"mean\": %.2f, \"p99\": %.2f}, "
                   "\"resident_us\": {\"median\": %.2f, \"mean\": %.2f, \"p99\": %.2f}, \"outputs_identical\": %s}\n",
                   names[kind], dt == 0 ? "f32" : "f64", F, C, calls, stats[0][0], stats[0][1], stats[0][2], stats[1][0],
                   stats[1][1], stats[1][2], same ? "true" : "false");
            fflush(stdout);
            CHECK(pipe_hip_destroy(plain));
            CHECK(pipe_hip_destroy(res));
        }
        free(in);
        free(out_a);
        free(out_b);
    }
    free(lat);
    return 0;
}
