#include "mock.hpp"

namespace pipe {
namespace mock {

mut::Mutation Mutator::MockMutation()
{
    return mut::Mutate(Mutability, [this]() -> error {
        Mutated = true;
        return nullptr;
    });
}

SourceAllocatorFunc Source::Allocator()
{
    return [this](mut::Context mctx, int, ::pipe::Source *out) -> error {
        Mutability = mctx;
        out->SignalProperties = SignalProperties{SampleRate, Channels};
        out->StartFunc = [this](const Context &c) { return Start(c); };
        out->FlushFunc = [this](const Context &c) { return Flush(c); };
        out->SourceFunc = [this](signal::Floating &s, int *n) -> error {
            if (ErrorOnCall)
                return ErrorOnCall;
            if (Samples == Limit)
                return io::EOF_();
            int read = s.Length();
            const int left = Limit - Samples;  // ensure that we have enough samples
            if (left < read)
                read = left;
            const int64_t base = (int64_t)Samples * Channels;
            for (int i = 0; i < read * Channels; ++i)
                s.SetSample(i, Generator ? Generator(base + i) : Value);
            advance(read);
            *n = read;
            return nullptr;
        };
        return ErrorOnMake;
    };
}

mut::Mutation Source::Reset()
{
    return mut::Mutate(Mutability, [this]() -> error {
        static_cast<Counter &>(*this) = Counter{};
        return nullptr;
    });
}

ProcessorAllocatorFunc Processor::Allocator()
{
    return [this](mut::Context mctx, int, SignalProperties props, ::pipe::Processor *out) -> error {
        Mutability = mctx;
        out->SignalProperties = props;
        out->StartFunc = [this](const Context &c) { return Start(c); };
        out->FlushFunc = [this](const Context &c) { return Flush(c); };
        out->ProcessFunc = [this](const signal::Floating &in, signal::Floating &o, int *n) -> error {
            if (ErrorOnCall)
                return ErrorOnCall;
            *n = signal::FloatingAsFloating(in, o);
            advance(*n);
            return nullptr;
        };
        return ErrorOnMake;
    };
}

SinkAllocatorFunc Sink::Allocator()
{
    return [this](mut::Context mctx, int bufferSize, SignalProperties props, ::pipe::Sink *out) -> error {
        Mutability = mctx;
        if (!Discard)
            Values = signal::Allocator{props.Channels, 0, bufferSize}.Float64();
        out->StartFunc = [this](const Context &c) { return Start(c); };
        out->FlushFunc = [this](const Context &c) { return Flush(c); };
        out->SinkFunc = [this](const signal::Floating &in) -> error {
            if (ErrorOnCall)
                return ErrorOnCall;
            if (!Discard)
                Values.Append(in);
            advance(in.Length());
            return nullptr;
        };
        return ErrorOnMake;
    };
}

}  // namespace mock
}  // namespace pipe
