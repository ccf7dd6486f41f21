// The fused FIR -> biquad -> gain kernel of a chain (chain_fused.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "common.hpp"

namespace pipehip {
namespace fused {

// One to four sections.  A cascade run as ONE 2S x 2S recurrence does not fit a wave's registers (the
// 4 x 4 matrices of two sections: 200+ spilled, and hipcc places spill stores inside exec-masked
// regions of this kernel -- lanes that were masked off then reload garbage); two sections run as two
// passes of the one-section machinery over the tile in segment layout (ols32_kernel.hpp,
// fused_epilogue_sections), spill-free.  Only spill-free instantiations are launched
// (scripts/check_spills.sh); longer cascades take the staged chain.
constexpr int kMaxFusedSections = 4;  // (three and four: the global look-back only, round 6)

class Plan {
public:
    struct Impl;
    Plan();
    ~Plan();
    Plan(const Plan &) = delete;
    Plan &operator=(const Plan &) = delete;

    static bool enabled();  // PIPE_HIP_CHAIN_FUSED=0 switches the fused form off
    // One launch: every Line advances by `frames` frames through FIR -> biquad (-> gain).
    // float32 buffers -- or float64 ones (`f64`: a handle switched with PIPE_HIP_PARAM_RELAXED_F64) --, an even channel
    // count, pointers aligned to a channel pair (the caller checks).
    int run(const pipe_hip_processor::FirFuseView &fir, const pipe_hip_processor::BiquadFuseView &bq, bool has_gain,
            double gain, const void *d_in, void *d_out, bool f64, int64_t frames, int channels, int lines, hipStream_t s,
            KernelTimer *timer, const char **kernel_name);
    // the fused kernel's workgroup (512 threads + its LDS) fits a CU of the current device
    static bool launchable();
    // this cascade on a call of `frames` frames: one section always; two to four sections when every section
    // forgets within a look-back window
    bool accepts(const double *coeffs, int S, int ntaps, int64_t frames, hipStream_t s);
    // EHIP if a launch since the last poll gave up waiting for a predecessor tile.  The caller has
    // synchronised the launch stream; this is a read of a pinned flag.
    int poll_error();
    // While a chain stays fused the cascade's state lives in the plan (two tagged slots per channel
    // pair: the last tile of a launch writes it, the first tiles of the next read it, no kernel in
    // between).  export_state() moves it back into the biquad stage's own array -- before the
    // staged chain, a windowed run or a partial restart touches that -- and drop_state() forgets it
    // (the stage's array was just reset).
    int export_state(hipStream_t s);
    void drop_state();
    // After a launch whose look-back gave up (poll_error() said so): the cascade's state of BEFORE that launch back
    // in the biquad stage's array (the launch wrote the slot with the older tag; the other still holds what it
    // read).  The caller points the FIR stage back at its old history and runs the call again on the staged chain.
    int rollback(hipStream_t s);
    // debug (PIPE_HIP_PARAM_DEBUG): the next launch's tiles of index `tile` publish nothing, and a wait gives up
    // after `limit_us` microseconds -- the failure a preempted predecessor would cause, on demand
    void debug_fail_next(int tile, double limit_us);

private:
    int prepare(const double *coeffs, int S, int ntaps, hipStream_t s);
    Impl *impl_;
};

}  // namespace fused
}  // namespace pipehip
