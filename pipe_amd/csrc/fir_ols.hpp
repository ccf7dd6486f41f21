// Overlap-save (1024-point float64 FFT) form of the FIR Processor: see fir_ols.hip.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace pipehip {
namespace ols {

class Plan {
public:
    struct Impl;
    Plan();
    ~Plan();
    Plan(const Plan &) = delete;
    Plan &operator=(const Plan &) = delete;

    // taps the FFT path is built for (otherwise the direct form is used)
    static bool supports(int ntaps, int channels);
    int init(int device, const double *taps, int ntaps, int channels = 2);
    // double-buffered like the direct form's taps: queued launches keep the old spectrum; the
    // upload is asynchronous on `s` (the stream the handle's launches go to)
    int set_taps(const double *taps, hipStream_t s);
    // work items (wave-sized 1024-point transforms) a call of this size launches
    int64_t items(int64_t frames, int channels, int lines) const;
    // advance every Line by `frames` frames; `hist` = the (N-1) frames before the call
    int run(const void *d_in, int in_dtype, void *d_out, int out_dtype, const double *hist, double *hist_new,
            int64_t frames,
            int channels, int lines, hipStream_t s, const char **kernel_name, KernelTimer *timer = nullptr);

    // filters above 512 taps run partitioned, on the 32 x 32 kernel only: 16-byte aligned buffers
    bool partitioned() const;
    int partitions() const;  // 1, or ceil(taps / 512)

    const Impl &impl() const { return *impl_; }

private:
    Impl *impl_;
};

}  // namespace ols
}  // namespace pipehip
