// fir_mfma.hip: the bit-exact direct-form FIR on the float64 matrix pipe (large calls).
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

namespace pipehip {

class KernelTimer;

constexpr int kFirMfmaMaxTaps = 4096;

// whether a call of this size goes to the matrix-pipe kernel (else fir.hip's VALU kernels)
bool fir_mfma_takes(int ntaps, int64_t frames, int channels, int lines, int cus, int64_t min_passes);

// one launch: `lines` Lines of `frames` frames x `channels` interleaved, history [lines][ntaps - 1][channels]
// float64 in `hist`, the next call's in `hist_new`.  *completion (may be null): handed to the launch as its
// stop event when the timer does not claim it, and cleared.
int run_fir_mfma(const void *d_in, int in_dtype, void *d_out, int out_dtype, const double *hist, double *hist_new,
                 const double *taps, int ntaps, int64_t frames, int channels, int lines, int cus, hipStream_t s,
                 const char **kernel_name, KernelTimer *timer, hipEvent_t *completion);

}  // namespace pipehip
