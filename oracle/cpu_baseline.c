/*
 * oracle/cpu_baseline.c -- TEST INFRASTRUCTURE: times the oracle's restatement
 * of the reference loop on the host CPU for bench.py's `cpu_baseline` leg
 * (kind "port": the Go reference cannot be built here).
 *
 *   cpu_baseline <lines> <channels> <buffer_frames> <buffers_per_line> <ntaps> <threads>
 *
 * threads == 1 : pipe.Run semantics, all Lines round-robin in one thread
 *                (run.go:112-132).
 * threads  > 1 : one pipe.Run per thread over a slice of the Lines (the
 *                reference's goroutine-per-Line async mode, merger.go:25-30).
 * Source = SplitMix64 synthetic stream, Processor = 256-tap style FIR via the
 * per-sample accessor path, Sink = discard.  Prints one JSON line.
 */
#define _GNU_SOURCE
#include "dsp_oracle.h"
#include "pipe_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int lines, channels, frames, buffers, ntaps;
    const double *taps;
    int first_line;
    int64_t sink_frames;
} job;

static void *run_job(void *arg)
{
    job *j = arg;
    opipe_line_desc *d = calloc((size_t)j->lines, sizeof *d);
    for (int i = 0; i < j->lines; i++) {
        d[i].src_kind = OPIPE_SRC_SYNTH;
        d[i].src_seed = 0x5EED0000ull + (uint64_t)(j->first_line + i);
        d[i].src_limit = (int64_t)j->frames * j->buffers;
        d[i].src_channels = j->channels;
        d[i].n_procs = 1;
        d[i].procs[0].kind = OPIPE_PROC_FIR;
        d[i].procs[0].params = j->taps;
        d[i].procs[0].n_params = j->ntaps;
        d[i].sink_discard = 1;
    }
    opipe_line_result *r = calloc((size_t)j->lines, sizeof *r);
    opipe_run_error e;
    opipe_run_lines(j->frames, j->lines, d, r, &e);
    j->sink_frames = 0;
    for (int i = 0; i < j->lines; i++)
        j->sink_frames += r[i].sink.samples;
    free(d);
    free(r);
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
    /* Hamming-windowed sinc, fc = 0.25 fs, unit DC gain */
    double *taps = malloc(sizeof(double) * (size_t)ntaps);
    double sum = 0;
    for (int k = 0; k < ntaps; k++) {
        double m = k - 0.5 * (ntaps - 1);
        double s = fabs(m) < 1e-12 ? 0.5 : sin(M_PI * 0.5 * m) / (M_PI * m);
        double w = ntaps > 1 ? 0.54 - 0.46 * cos(2.0 * M_PI * k / (ntaps - 1)) : 1.0;
        taps[k] = s * w;
        sum += taps[k];
    }
    for (int k = 0; k < ntaps; k++)
        taps[k] /= sum;

    job *jobs = calloc((size_t)threads, sizeof *jobs);
    pthread_t *tid = calloc((size_t)threads, sizeof *tid);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int first = 0;
    for (int t = 0; t < threads; t++) {
        int n = lines / threads + (t < lines % threads ? 1 : 0);
        jobs[t] = (job){n, channels, frames, buffers, ntaps, taps, first, 0};
        first += n;
        if (threads == 1)
            run_job(&jobs[t]);
        else
            pthread_create(&tid[t], NULL, run_job, &jobs[t]);
    }
    int64_t total_frames = 0;
    for (int t = 0; t < threads; t++) {
        if (threads > 1)
            pthread_join(tid[t], NULL);
        total_frames += jobs[t].sink_frames;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    double samples = (double)total_frames * channels;
    printf("{\"seconds\": %.6f, \"frames\": %lld, \"scalar_samples\": %.0f, "
           "\"msamples_per_s\": %.4f, \"threads\": %d, \"lines\": %d, \"channels\": %d, "
           "\"buffer_frames\": %d, \"buffers_per_line\": %d, \"ntaps\": %d}\n",
           sec, (long long)total_frames, samples, samples / sec / 1e6, threads, lines, channels,
           frames, buffers, ntaps);
    free(taps);
    free(jobs);
    free(tid);
    return 0;
}
