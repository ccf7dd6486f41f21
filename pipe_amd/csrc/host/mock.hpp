// Test doubles mirroring /root/reference/mock/mock.go: constant-value Source with
// Limit, pass-through Processor, counting/appending Sink, with the same
// fault-injection fields.  Host-only, exactly like the reference's mocks (they are
// the reference's own CPU components, not a fallback for the HIP Processors).
#pragma once

#include "pipe.hpp"

namespace pipe {
namespace mock {

struct Counter {  // mock.go:17-21,43-46
    int Messages = 0;
    int Samples = 0;  // frames
    signal::Floating Values;
    void advance(int size)
    {
        ++Messages;
        Samples += size;
    }
};
struct Flusher {  // mock.go:24-27,48-52
    bool Flushed = false;
    error ErrorOnFlush;
    error Flush(const Context &)
    {
        Flushed = true;
        return ErrorOnFlush;
    }
};
struct Starter {  // mock.go:30-33,54-58
    bool Started = false;
    error ErrorOnStart;
    error Start(const Context &)
    {
        Started = true;
        return ErrorOnStart;
    }
};
struct Mutator {  // mock.go:36-39,120-127
    mut::Context Mutability;
    bool Mutated = false;
    mut::Mutation MockMutation();
};

struct Source : Mutator, Counter, Starter, Flusher {  // mock.go:61-72
    int Limit = 0;
    double Value = 0;
    int Channels = 0;
    signal::Frequency SampleRate = 0;
    error ErrorOnCall;
    error ErrorOnMake;
    // optional generator replacing the constant fill: value of flat sample i
    std::function<double(int64_t)> Generator;
    SourceAllocatorFunc Allocator();  // mock.go:75-109  (Go: m.Source())
    mut::Mutation Reset();            // mock.go:111-118
};

struct Processor : Mutator, Counter, Starter, Flusher {  // mock.go:130-137
    error ErrorOnCall;
    error ErrorOnMake;
    ProcessorAllocatorFunc Allocator();  // mock.go:139-157
};

struct Sink : Mutator, Counter, Starter, Flusher {  // mock.go:160-168
    bool Discard = false;
    error ErrorOnCall;
    error ErrorOnMake;
    SinkAllocatorFunc Allocator();  // mock.go:170-192
};

}  // namespace mock
}  // namespace pipe
