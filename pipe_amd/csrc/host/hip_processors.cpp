#include <cstdlib>
#include "hip_processors.hpp"

#include <string>

#include "pipe_hip.h"

namespace pipe {
namespace hip {

error StatusError(int status, const char *what)
{
    if (status == PIPE_HIP_OK)
        return nullptr;
    std::string m = std::string("pipe_hip ") + what + ": " + pipe_hip_strerror(status);
    if (status == PIPE_HIP_EHIP)
        m += " (hipError " + std::to_string(pipe_hip_last_hip_error()) + ")";
    return NewError(m);
}

Handle::~Handle()
{
    if (p_)
        pipe_hip_destroy(p_);
}

mut::MutatorFunc Handle::SetGain(double g)
{
    pipe_hip_processor *p = p_;
    return [p, g]() -> error { return StatusError(pipe_hip_set_param(p, PIPE_HIP_PARAM_GAIN, &g, 1), "set gain"); };
}

mut::MutatorFunc Handle::SetTaps(std::vector<double> taps)
{
    pipe_hip_processor *p = p_;
    return [p, taps]() -> error {
        return StatusError(pipe_hip_set_param(p, PIPE_HIP_PARAM_TAPS, taps.data(), (int32_t)taps.size()), "set taps");
    };
}

mut::MutatorFunc Handle::SetCoeffs(std::vector<double> coeffs)
{
    pipe_hip_processor *p = p_;
    return [p, coeffs]() -> error {
        return StatusError(pipe_hip_set_param(p, PIPE_HIP_PARAM_COEFFS, coeffs.data(), (int32_t)coeffs.size()),
                           "set coeffs");
    };
}

void UsePinnedPools() { signal::SetPinnedAllocator(pipe_hip_host_alloc, pipe_hip_host_free); }

namespace {

pipe_hip_config make_cfg(const Options &o, int bufferSize, const SignalProperties &in)
{
    pipe_hip_config c{};
    c.device = o.device;
    c.buffer_size = bufferSize;
    c.channels = in.Channels;
    c.dtype = PIPE_HIP_F64;  // the reference pipe carries float64 (pipe.go:394,437)
    c.lines = 1;
    c.max_batch = 1;
    return c;
}

// fill the Processor's hooks from a created handle
error finish(pipe_hip_processor *raw, const SignalProperties &input, Processor *out,
             std::shared_ptr<Handle> *handle)
{
    auto h = std::make_shared<Handle>(raw);
    if (handle)
        *handle = h;
    // PIPE_HOST_RESIDENT (this harness's switch, what hip.Stage.SetResident is in the Go shim): every stage that can
    // take a queued launch back keeps its next buffer's work queued behind a doorbell (PIPE_HIP_PARAM_RESIDENT);
    // the others answer EINVAL and stay on the plain path.  The whole host-loop test suite then runs through it.
    if (const char *e = std::getenv("PIPE_HOST_RESIDENT")) {
        const double on = std::atof(e);
        if (on != 0.0)
            (void)pipe_hip_set_param(raw, PIPE_HIP_PARAM_RESIDENT, &on, 1);
    }
    // PIPE_HOST_RESIDENT_SHARED: every such stage joins its device's SHARED doorbell queue instead
    // (PIPE_HIP_PARAM_RESIDENT_SHARED: what a synchronous pipe.Run would ask for -- all Lines in one goroutine, round-robin)
    if (const char *e = std::getenv("PIPE_HOST_RESIDENT_SHARED")) {
        const double on = std::atof(e);
        if (on != 0.0)
            (void)pipe_hip_set_param(raw, PIPE_HIP_PARAM_RESIDENT_SHARED, &on, 1);
    }
    int32_t ch = 0, up = 1, down = 1;
    if (error e = StatusError(pipe_hip_output_properties(raw, &ch, &up, &down), "output_properties"))
        return e;
    out->SignalProperties = SignalProperties{input.SampleRate * up / down, ch};
    out->StartFunc = [h](const Context &) -> error { return StatusError(pipe_hip_start(h->get()), "start"); };
    out->FlushFunc = [h](const Context &) -> error { return StatusError(pipe_hip_flush(h->get()), "flush"); };
    out->ProcessFunc = [h](const signal::Floating &in, signal::Floating &o, int *n) -> error {
        int32_t written = 0;
        const int st = pipe_hip_process(h->get(), in.data(), in.Length(), o.data(), o.Length(), &written);
        if (st != PIPE_HIP_OK)
            return StatusError(st, "process");
        *n = written;
        return nullptr;
    };
    return nullptr;
}

}  // namespace

ProcessorAllocatorFunc Gain(double gain, Options o, std::shared_ptr<Handle> *handle)
{
    return [=](mut::Context, int bufferSize, SignalProperties input, Processor *out) -> error {
        const pipe_hip_config c = make_cfg(o, bufferSize, input);
        pipe_hip_processor *raw = nullptr;
        if (error e = StatusError(pipe_hip_gain_create(&c, gain, &raw), "gain_create"))
            return e;
        return finish(raw, input, out, handle);
    };
}

ProcessorAllocatorFunc Copy(Options o, std::shared_ptr<Handle> *handle) { return Gain(1.0, o, handle); }

ProcessorAllocatorFunc Fir(std::vector<double> taps, Options o, std::shared_ptr<Handle> *handle)
{
    return [=](mut::Context, int bufferSize, SignalProperties input, Processor *out) -> error {
        const pipe_hip_config c = make_cfg(o, bufferSize, input);
        pipe_hip_processor *raw = nullptr;
        if (error e = StatusError(pipe_hip_fir_create(&c, taps.data(), (int32_t)taps.size(), &raw), "fir_create"))
            return e;
        return finish(raw, input, out, handle);
    };
}

ProcessorAllocatorFunc Biquad(std::vector<double> coeffs, Options o, std::shared_ptr<Handle> *handle)
{
    return [=](mut::Context, int bufferSize, SignalProperties input, Processor *out) -> error {
        const pipe_hip_config c = make_cfg(o, bufferSize, input);
        pipe_hip_processor *raw = nullptr;
        if (error e = StatusError(pipe_hip_biquad_create(&c, coeffs.data(), (int32_t)(coeffs.size() / 5), &raw),
                                  "biquad_create"))
            return e;
        return finish(raw, input, out, handle);
    };
}

namespace {

error build_chain(const std::vector<StageSpec> &stages, const pipe_hip_config &c, pipe_hip_processor **chain_out)
{
    {
        std::vector<pipe_hip_processor *> raws;
        auto cleanup = [&raws]() {
            for (auto *r : raws)
                pipe_hip_destroy(r);
        };
        for (const StageSpec &s : stages) {
            pipe_hip_processor *raw = nullptr;
            int st = PIPE_HIP_EINVAL;
            if (s.kind == StageSpec::kGain && s.params.size() == 1)
                st = pipe_hip_gain_create(&c, s.params[0], &raw);
            else if (s.kind == StageSpec::kFir)
                st = pipe_hip_fir_create(&c, s.params.data(), (int32_t)s.params.size(), &raw);
            else if (s.kind == StageSpec::kBiquad)
                st = pipe_hip_biquad_create(&c, s.params.data(), (int32_t)(s.params.size() / 5), &raw);
            if (st != PIPE_HIP_OK) {
                cleanup();
                return StatusError(st, "chain stage create");
            }
            raws.push_back(raw);
        }
        pipe_hip_processor *chain = nullptr;
        const int st = pipe_hip_chain_create(raws.data(), (int32_t)raws.size(), &chain);
        if (st != PIPE_HIP_OK) {
            cleanup();
            return StatusError(st, "chain_create");
        }
        *chain_out = chain;
        return nullptr;
    }
}

// one handle shared by the Processors of `lines` Lines
class HipBatch final : public BatchGroup {
public:
    HipBatch(std::vector<StageSpec> stages, int lines, Options o) : stages_(std::move(stages)), lines_(lines), o_(o) {}
    int Slots() const override { return lines_; }

    // first allocator call creates the handle; the others must agree on the geometry
    error bind(int bufferSize, const SignalProperties &input, std::shared_ptr<Handle> *handle)
    {
        if (h_) {
            if (bufferSize != bufferSize_ || input.Channels != channels_)
                return NewError("batched chain: every Line must have the same buffer size and channels");
            return nullptr;
        }
        pipe_hip_config c = make_cfg(o_, bufferSize, input);
        c.lines = lines_;
        pipe_hip_processor *raw = nullptr;
        if (error e = build_chain(stages_, c, &raw))
            return e;
        h_ = std::make_shared<Handle>(raw);
        if (handle)
            *handle = h_;
        bufferSize_ = bufferSize;
        channels_ = input.Channels;
        return nullptr;
    }
    const std::shared_ptr<Handle> &handle() const { return h_; }

    // A slot's StartFunc.  The Lines of a group start together before the first pass
    // (run.go:76-85): the first of them starts the WHOLE handle once (one drain and one memset per
    // stage, not `lines` of them) and the others find it started.  A Line that joins a group that
    // has run a pass since (Pipe.AddLine) must not reset the others: its slot alone starts from
    // silence (pipe_hip_start_lines).
    error start(int slot)
    {
        int st = PIPE_HIP_OK;
        if (live_ == 0) {  // no Line of the group is live
            st = pipe_hip_start(h_->get());
            dirty_ = false;
        } else if (dirty_) {
            st = pipe_hip_start_lines(h_->get(), slot, 1);
        }
        if (st == PIPE_HIP_OK)
            ++live_;  // (a failed StartFunc gets no FlushFunc: run.go:54-62)
        return StatusError(st, "start");
    }
    // a Line's FlushFunc (its stream ended, or the pipe stops): the other Lines keep running
    error flush()
    {
        if (live_ > 0)
            --live_;
        return StatusError(pipe_hip_flush(h_->get()), "flush");
    }

    error ProcessLines(const std::vector<const signal::Floating *> &ins, const std::vector<signal::Floating *> &outs,
                       std::vector<int> *processed) override
    {
        const size_t n = (size_t)lines_;
        dirty_ = true;
        in_ptr_.assign(n, nullptr);
        out_ptr_.assign(n, nullptr);
        frames_.assign(n, 0);
        written_.assign(n, 0);
        bool pinned = true;
        for (size_t i = 0; i < n; ++i) {
            if (!ins[i])
                continue;
            if (outs[i]->Length() < ins[i]->Length())
                return NewError("batched chain: output buffer shorter than input");
            pinned = pinned && ins[i]->Pinned() && outs[i]->Pinned();
            in_ptr_[i] = ins[i]->data();
            out_ptr_[i] = outs[i]->data();
            frames_[i] = ins[i]->Length();
        }
        // pool buffers on pinned memory: the device gathers / scatters them itself
        const int st = pinned ? pipe_hip_process_lines_pinned(h_->get(), in_ptr_.data(), frames_.data(),
                                                              out_ptr_.data(), written_.data())
                              : pipe_hip_process_lines(h_->get(), in_ptr_.data(), frames_.data(), out_ptr_.data(),
                                                       written_.data());
        if (st != PIPE_HIP_OK)
            return StatusError(st, "process_lines");
        for (size_t i = 0; i < n; ++i)
            (*processed)[i] = written_[i];
        return nullptr;
    }

private:
    std::vector<StageSpec> stages_;
    int lines_;
    Options o_;
    std::shared_ptr<Handle> h_;
    int bufferSize_ = 0, channels_ = 0;
    int live_ = 0;        // slots started and not flushed
    bool dirty_ = false;  // a pass has run since the whole handle was last started
    std::vector<const void *> in_ptr_;
    std::vector<void *> out_ptr_;
    std::vector<int32_t> frames_, written_;
};

}  // namespace

ProcessorAllocatorFunc Chain(std::vector<StageSpec> stages, Options o, std::shared_ptr<Handle> *handle)
{
    return [=](mut::Context, int bufferSize, SignalProperties input, Processor *out) -> error {
        const pipe_hip_config c = make_cfg(o, bufferSize, input);
        pipe_hip_processor *chain = nullptr;
        if (error e = build_chain(stages, c, &chain))
            return e;
        return finish(chain, input, out, handle);
    };
}

std::vector<ProcessorAllocatorFunc> BatchedChain(std::vector<StageSpec> stages, int lines, Options o,
                                                 std::shared_ptr<Handle> *handle)
{
    auto group = std::make_shared<HipBatch>(std::move(stages), lines, o);
    std::vector<ProcessorAllocatorFunc> allocs;
    for (int slot = 0; slot < lines; ++slot) {
        allocs.push_back([group, slot, handle](mut::Context, int bufferSize, SignalProperties input, Processor *out) -> error {
            if (error e = group->bind(bufferSize, input, handle))
                return e;
            out->SignalProperties = input;  // a chain keeps rate and channels
            out->Batch = group;
            out->BatchSlot = slot;
            out->StartFunc = [group, slot](const Context &) -> error { return group->start(slot); };
            out->FlushFunc = [group](const Context &) -> error { return group->flush(); };
            out->ProcessFunc = [](const signal::Floating &, signal::Floating &, int *) -> error {
                return NewError("batched processor: run the Lines with pipe::RunBatched");
            };
            return nullptr;
        });
    }
    return allocs;
}

}  // namespace hip
}  // namespace pipe
