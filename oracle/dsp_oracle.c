/*
 * oracle/dsp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * See dsp_oracle.h for the arithmetic contract.  PARITY UNPINNED for every
 * function here: no FIR/biquad/gain/resampler/mix exists in /root/reference
 * (SURVEY.md F1/F2); scipy cross-checks live in tests/test_oracle_dsp.py.
 *
 * Built with -ffp-contract=off so that the ONLY fused operations are the
 * explicit fma() calls.
 */
#include "dsp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- gain ---------------------------------------------------------------- */
void odsp_gain(const double *in, double *out, int64_t samples, double gain)
{
    for (int64_t i = 0; i < samples; i++)
        out[i] = in[i] * gain;
}

/* ---- FIR ----------------------------------------------------------------- */
struct odsp_fir {
    int ntaps, channels;
    double *taps;
    double *hist; /* (ntaps-1) frames, oldest first, interleaved */
};

odsp_fir *odsp_fir_new(const double *taps, int ntaps, int channels)
{
    if (ntaps < 1 || channels < 1)
        return NULL;
    odsp_fir *f = calloc(1, sizeof *f);
    f->ntaps = ntaps;
    f->channels = channels;
    f->taps = malloc(sizeof(double) * (size_t)ntaps);
    memcpy(f->taps, taps, sizeof(double) * (size_t)ntaps);
    f->hist = calloc((size_t)(ntaps - 1) * (size_t)channels + 1, sizeof(double));
    return f;
}

void odsp_fir_reset(odsp_fir *f)
{
    memset(f->hist, 0, sizeof(double) * (size_t)(f->ntaps - 1) * (size_t)f->channels);
}

void odsp_fir_set_taps(odsp_fir *f, const double *taps)
{
    memcpy(f->taps, taps, sizeof(double) * (size_t)f->ntaps);
}

/* x[m][c] for m in [-(ntaps-1), frames): negative m reads the history */
static inline double fir_x(const odsp_fir *f, const double *in, int64_t m, int c)
{
    if (m >= 0)
        return in[m * f->channels + c];
    return f->hist[(m + (f->ntaps - 1)) * f->channels + c];
}

void odsp_fir_process(odsp_fir *f, const double *in, double *out, int64_t frames)
{
    const int C = f->channels, N = f->ntaps;
    for (int64_t n = 0; n < frames; n++) {
        for (int c = 0; c < C; c++) {
            double acc = 0.0;
            for (int k = 0; k < N; k++)
                acc = fma(f->taps[k], fir_x(f, in, n - k, c), acc);
            out[n * C + c] = acc;
        }
    }
    /* carry the last ntaps-1 frames of (history ++ in) */
    const int64_t H = N - 1;
    if (H > 0) {
        if (frames >= H) {
            memcpy(f->hist, in + (frames - H) * C, sizeof(double) * (size_t)(H * C));
        } else {
            memmove(f->hist, f->hist + frames * C, sizeof(double) * (size_t)((H - frames) * C));
            memcpy(f->hist + (H - frames) * C, in, sizeof(double) * (size_t)(frames * C));
        }
    }
}

void odsp_fir_free(odsp_fir *f)
{
    if (!f)
        return;
    free(f->taps);
    free(f->hist);
    free(f);
}

/* ---- biquad cascade (DF2T) ------------------------------------------------ */
struct odsp_biquad {
    int nsections, channels;
    double *coeffs; /* nsections x 5 */
    double *state;  /* channels x nsections x 2 */
};

odsp_biquad *odsp_biquad_new(const double *coeffs, int nsections, int channels)
{
    if (nsections < 1 || channels < 1)
        return NULL;
    odsp_biquad *b = calloc(1, sizeof *b);
    b->nsections = nsections;
    b->channels = channels;
    b->coeffs = malloc(sizeof(double) * 5u * (size_t)nsections);
    memcpy(b->coeffs, coeffs, sizeof(double) * 5u * (size_t)nsections);
    b->state = calloc((size_t)channels * (size_t)nsections * 2u, sizeof(double));
    return b;
}

void odsp_biquad_reset(odsp_biquad *b)
{
    memset(b->state, 0, sizeof(double) * (size_t)b->channels * (size_t)b->nsections * 2u);
}

void odsp_biquad_set_coeffs(odsp_biquad *b, const double *coeffs)
{
    memcpy(b->coeffs, coeffs, sizeof(double) * 5u * (size_t)b->nsections);
}

void odsp_biquad_process(odsp_biquad *b, const double *in, double *out, int64_t frames)
{
    const int C = b->channels, S = b->nsections;
    for (int64_t n = 0; n < frames; n++) {
        for (int c = 0; c < C; c++) {
            double x = in[n * C + c];
            for (int s = 0; s < S; s++) {
                const double *q = b->coeffs + 5 * s;
                double *st = b->state + ((size_t)c * (size_t)S + (size_t)s) * 2u;
                double y = fma(q[0], x, st[0]);
                double t = fma(q[1], x, st[1]);
                st[0] = fma(-q[3], y, t);
                double u = q[2] * x;
                st[1] = fma(-q[4], y, u);
                x = y;
            }
            out[n * C + c] = x;
        }
    }
}

void odsp_biquad_free(odsp_biquad *b)
{
    if (!b)
        return;
    free(b->coeffs);
    free(b->state);
    free(b);
}

/* ---- rational polyphase resampler ----------------------------------------- */
struct odsp_resampler {
    int T, up, down, channels;
    double *proto;   /* up*T */
    double *hist;    /* (T-1) frames, oldest first */
    int64_t in_total;  /* input frames consumed since reset */
    int64_t out_total; /* output frames produced since reset */
};

odsp_resampler *odsp_resampler_new(const double *proto, int taps_per_phase,
                                   int up, int down, int channels)
{
    if (taps_per_phase < 1 || up < 1 || down < 1 || channels < 1)
        return NULL;
    odsp_resampler *r = calloc(1, sizeof *r);
    r->T = taps_per_phase;
    r->up = up;
    r->down = down;
    r->channels = channels;
    size_t n = (size_t)up * (size_t)taps_per_phase;
    r->proto = malloc(sizeof(double) * n);
    memcpy(r->proto, proto, sizeof(double) * n);
    r->hist = calloc((size_t)(taps_per_phase - 1) * (size_t)channels + 1, sizeof(double));
    return r;
}

void odsp_resampler_reset(odsp_resampler *r)
{
    memset(r->hist, 0, sizeof(double) * (size_t)(r->T - 1) * (size_t)r->channels);
    r->in_total = 0;
    r->out_total = 0;
}

/* output m reads input frame floor(m*down/up); it is emitted once that frame
 * has been consumed: m < ceil(in_total*up/down) */
static int64_t resampler_out_end(const odsp_resampler *r, int64_t in_total)
{
    return (in_total * r->up + r->down - 1) / r->down;
}

int64_t odsp_resampler_out_frames(const odsp_resampler *r, int64_t in_frames)
{
    return resampler_out_end(r, r->in_total + in_frames) - r->out_total;
}

int64_t odsp_resampler_process(odsp_resampler *r, const double *in, int64_t in_frames,
                               double *out, int64_t out_cap_frames)
{
    const int C = r->channels, T = r->T;
    const int64_t n_out = odsp_resampler_out_frames(r, in_frames);
    if (n_out > out_cap_frames)
        return -1;
    for (int64_t i = 0; i < n_out; i++) {
        const int64_t m = r->out_total + i;
        const int64_t t = m * r->down;
        const int64_t n = t / r->up - r->in_total; /* index relative to `in` */
        const int p = (int)(t % r->up);
        for (int c = 0; c < C; c++) {
            double acc = 0.0;
            for (int j = 0; j < T; j++) {
                const int64_t idx = n - j;
                double x = idx >= 0 ? in[idx * C + c]
                                    : r->hist[(idx + (T - 1)) * C + c];
                acc = fma(r->proto[p + (int64_t)j * r->up], x, acc);
            }
            out[i * C + c] = acc;
        }
    }
    const int64_t H = T - 1;
    if (H > 0) {
        if (in_frames >= H) {
            memcpy(r->hist, in + (in_frames - H) * C, sizeof(double) * (size_t)(H * C));
        } else {
            memmove(r->hist, r->hist + in_frames * C,
                    sizeof(double) * (size_t)((H - in_frames) * C));
            memcpy(r->hist + (H - in_frames) * C, in, sizeof(double) * (size_t)(in_frames * C));
        }
    }
    r->in_total += in_frames;
    r->out_total += n_out;
    return n_out;
}

void odsp_resampler_free(odsp_resampler *r)
{
    if (!r)
        return;
    free(r->proto);
    free(r->hist);
    free(r);
}

/* ---- mix ------------------------------------------------------------------ */
void odsp_mix(const double *const *ins, int n_inputs, double *out, int64_t samples)
{
    for (int64_t i = 0; i < samples; i++) {
        double acc = ins[0][i];
        for (int k = 1; k < n_inputs; k++)
            acc = acc + ins[k][i];
        out[i] = acc;
    }
}

/* ---- synthetic input ------------------------------------------------------- */
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1u) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void odsp_synth_fill(uint64_t seed, int64_t first_index, double *out, int64_t samples)
{
    for (int64_t i = 0; i < samples; i++) {
        uint64_t u = splitmix64_at(seed, (uint64_t)(first_index + i));
        out[i] = (double)(u >> 40) * 0x1p-23 - 1.0;
    }
}
