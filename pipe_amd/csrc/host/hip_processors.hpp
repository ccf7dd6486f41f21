// ProcessorAllocatorFuncs backed by the HIP Processors: what the Go cgo shim of
// INTEGRATION.md does, written against the same C ABI (include/pipe_hip.h) so the
// drop-in seam is exercised end to end from compiled host code.
//
// Each allocator (line.go:26-30) creates the device handle for (bufferSize,
// input.Channels), echoes/derives the output SignalProperties, and returns a
// Processor whose hooks forward to the handle:
//     StartFunc   -> pipe_hip_start      (zero history; pipe may be restarted)
//     ProcessFunc -> pipe_hip_process    (host float64 buffers, synchronous)
//     FlushFunc   -> pipe_hip_flush
// Parameter changes travel as mutable.Mutations and end in pipe_hip_set_param.
#pragma once

#include <memory>
#include <vector>

#include "pipe.hpp"

struct pipe_hip_processor;

namespace pipe {
namespace hip {

struct Options {
    int device = 0;
};

// Owns one pipe_hip_processor; shared by the hooks of the Processor it backs.
class Handle {
public:
    explicit Handle(pipe_hip_processor *p) : p_(p) {}
    ~Handle();
    pipe_hip_processor *get() const { return p_; }
    // mutation bodies (applied in the processing thread right before ProcessFunc, pipe.go:433)
    mut::MutatorFunc SetGain(double g);
    mut::MutatorFunc SetTaps(std::vector<double> taps);
    mut::MutatorFunc SetCoeffs(std::vector<double> coeffs);

private:
    pipe_hip_processor *p_;
};

// Each function returns the allocator; *handle (optional) receives the created
// Handle at bind time so that tests / callers can queue mutations on it.
ProcessorAllocatorFunc Copy(Options o = {}, std::shared_ptr<Handle> *handle = nullptr);
ProcessorAllocatorFunc Gain(double gain, Options o = {}, std::shared_ptr<Handle> *handle = nullptr);
ProcessorAllocatorFunc Fir(std::vector<double> taps, Options o = {}, std::shared_ptr<Handle> *handle = nullptr);
ProcessorAllocatorFunc Biquad(std::vector<double> coeffs, Options o = {}, std::shared_ptr<Handle> *handle = nullptr);

// A fused run of stages (FIR -> biquad -> gain ...) as ONE Processor: the
// intermediates stay on the device (pipe_hip_chain_create).
struct StageSpec {
    enum Kind { kGain, kFir, kBiquad } kind;
    std::vector<double> params;  // gain: {g}; fir: taps; biquad: nsections x 5
};
ProcessorAllocatorFunc Chain(std::vector<StageSpec> stages, Options o = {},
                             std::shared_ptr<Handle> *handle = nullptr);

// The same chain for `lines` Lines behind ONE device handle (cfg.lines = lines): element i of
// the result is the allocator for Line i.  The Processors it makes only run under
// pipe::RunBatched, where all of them advance with one pipe_hip_process_lines call per pass;
// their parameters are shared (a mutation queued on *handle changes every Line of the group).
std::vector<ProcessorAllocatorFunc> BatchedChain(std::vector<StageSpec> stages, int lines, Options o = {},
                                                 std::shared_ptr<Handle> *handle = nullptr);

// route pool buffers through pinned memory (called once when a device exists)
void UsePinnedPools();

error StatusError(int status, const char *what);

}  // namespace hip
}  // namespace pipe
