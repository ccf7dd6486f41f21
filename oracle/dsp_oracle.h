/*
 * oracle/dsp_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU (plain C, float64) definition of the DSP stage bodies that the HIP
 * Processors implement: gain, direct-form FIR, DF2T biquad cascade, rational
 * polyphase resampler and n-input mix.
 *
 * PARITY UNPINNED: the reference (/root/reference, pipelined.dev/pipe) ships no
 * FIR / biquad / gain / resampler / mixer -- its only ProcessFunc is the
 * pass-through copy in mock/mock.go:139-157.  The operation
 * order written here therefore *is* the specification; it is cross-checked
 * against scipy (lfilter / sosfilt / upfirdn) in tests/test_oracle_dsp.py and
 * frozen as fixtures under tests/golden/.
 *
 * Arithmetic contract (both this file and the HIP kernels follow it exactly, so
 * float64 results are bit-identical and float32 results are the correctly
 * rounded float64 results):
 *   - all state and accumulation in IEEE-754 binary64, round-to-nearest-even;
 *   - every multiply-add written fma(a,b,c) is ONE fused operation;
 *   - FIR:      acc = +0.0; for k = 0..N-1: acc = fma(h[k], x[n-k], acc)
 *   - biquad:   y = fma(b0,x,s1); s1 = fma(-a1,y,fma(b1,x,s2)); s2 = fma(-a2,y,b2*x)
 *   - gain:     y = x*g
 *   - resample: acc = +0.0; for j = 0..T-1: acc = fma(h[p + j*L], x[n-j], acc)
 *   - mix:      y = ((in0 + in1) + in2) + ...
 *   - float32 I/O: inputs widened exactly, output = (float)y (one extra RNE
 *     rounding of the float64 result).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.
 */
#ifndef PIPE_DSP_ORACLE_H
#define PIPE_DSP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- gain ---------------------------------------------------------------- */
void odsp_gain(const double *in, double *out, int64_t samples, double gain);

/* ---- FIR (same taps for every channel, interleaved frames) --------------- */
typedef struct odsp_fir odsp_fir;
odsp_fir *odsp_fir_new(const double *taps, int ntaps, int channels);
void odsp_fir_reset(odsp_fir *f);                 /* StartFunc: zero history */
void odsp_fir_set_taps(odsp_fir *f, const double *taps); /* mutation, same ntaps */
/* process `frames` interleaved frames; in and out may not alias */
void odsp_fir_process(odsp_fir *f, const double *in, double *out, int64_t frames);
void odsp_fir_free(odsp_fir *f);

/* ---- biquad cascade, DF2T; coeffs = nsections x {b0,b1,b2,a1,a2} --------- */
typedef struct odsp_biquad odsp_biquad;
odsp_biquad *odsp_biquad_new(const double *coeffs, int nsections, int channels);
void odsp_biquad_reset(odsp_biquad *b);
void odsp_biquad_set_coeffs(odsp_biquad *b, const double *coeffs);
void odsp_biquad_process(odsp_biquad *b, const double *in, double *out, int64_t frames);
void odsp_biquad_free(odsp_biquad *b);

/* ---- rational polyphase resampler up/down -------------------------------- */
/* proto has up*taps_per_phase entries; phase p uses proto[p + j*up].         */
typedef struct odsp_resampler odsp_resampler;
odsp_resampler *odsp_resampler_new(const double *proto, int taps_per_phase,
                                   int up, int down, int channels);
void odsp_resampler_reset(odsp_resampler *r);
/* frames the next call would emit for in_frames new input frames */
int64_t odsp_resampler_out_frames(const odsp_resampler *r, int64_t in_frames);
/* returns frames written, or -1 if out_cap_frames is too small (nothing is
 * consumed in that case) */
int64_t odsp_resampler_process(odsp_resampler *r, const double *in, int64_t in_frames,
                               double *out, int64_t out_cap_frames);
void odsp_resampler_free(odsp_resampler *r);

/* ---- mix: out = ((in[0] + in[1]) + in[2]) ... ----------------------------- */
void odsp_mix(const double *const *ins, int n_inputs, double *out, int64_t samples);

/* ---- synthetic input: SplitMix64(seed) -> [-1,1) ---------- */
/* sample i of the stream with the given seed; exactly representable in f32 */
void odsp_synth_fill(uint64_t seed, int64_t first_index, double *out, int64_t samples);

#ifdef __cplusplus
}
#endif
#endif
