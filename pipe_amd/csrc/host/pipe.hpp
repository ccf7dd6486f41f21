// Host-side mirror of the reference's operator API for the Processor-stage hot
// path (Go toolchain absent here, so the host side above the C ABI is C++):
//
//   pipe::Line{Source, Processors, Sink}  + *AllocatorFunc      line.go:14-35
//   pipe::SignalProperties                                      line.go:38-41
//   pipe::Source / Processor / Sink + *Func hook types          pipe.go:32-87
//   pipe::Run(ctx, bufferSize, lines...)   (sync, one thread)   pipe.go:89-103
//   pipe::New(bufferSize, lines...) -> Pipe::Start / Wait /Push pipe.go:105-126,197-257
//   executors, sync/async fittings, mutations                   run.go, fitting.go, mutable.go
//
// Same names, argument meaning and error behaviour as the reference, so that the
// tests read like pipe_test.go.  Live graph edits (AddLine / InsertProcessor,
// pipe.go:260-365) are out of scope.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "signal.hpp"

namespace pipe {

// ---- errors (Go `error` values) ---------------------------------------------
struct ErrorValue {
    std::string msg;
    std::shared_ptr<const ErrorValue> cause;  // %w
};
using error = std::shared_ptr<const ErrorValue>;  // nullptr == nil
error NewError(const std::string &msg);
error Wrap(const std::string &prefix, const error &cause);  // fmt.Errorf("prefix: %w", cause)
bool Is(const error &err, const error &target);             // errors.Is
std::string ErrorString(const error &err);
namespace io {
const error &EOF_();  // io.EOF sentinel (graceful end of stream)
}

// ErrorRun{ErrExec, ErrFlush}                                    error.go:11-39
struct ErrorRun {
    error ErrExec;
    error ErrFlush;
};

// ---- context.Context (cancellation only) ----------------------------------------
class Context {
public:
    Context() : done_(std::make_shared<std::atomic<bool>>(false)) {}
    static Context Background() { return Context(); }
    bool Done() const { return done_->load(std::memory_order_acquire); }
    void Cancel() const { done_->store(true, std::memory_order_release); }

private:
    std::shared_ptr<std::atomic<bool>> done_;
};

// ---- mutable (in-band parameter changes)                    mutable/mutable.go ----
namespace mut {
using MutatorFunc = std::function<error()>;
struct Context {  // 16 random bytes; all-zero == immutable     mutable.go:12,28-37
    uint64_t hi = 0, lo = 0;
    bool IsMutable() const { return hi != 0 || lo != 0; }
    bool operator<(const Context &o) const { return hi != o.hi ? hi < o.hi : lo < o.lo; }
    bool operator==(const Context &o) const { return hi == o.hi && lo == o.lo; }
};
Context Mutable();
inline Context Immutable() { return Context{}; }
struct Mutation {  // mutable.go:15-18,40-58
    Context ctx;
    MutatorFunc mutator;
    void Apply() const
    {
        if (mutator)
            mutator();
    }
};
Mutation Mutate(const Context &c, MutatorFunc m);  // throws std::logic_error on immutable
// Mutations: Context -> queued mutators; nullptr == nil map      mutable.go:21,61-122
class Mutations {
public:
    bool nil() const { return !map_; }
    Mutations &Put(const Mutation &m);
    error ApplyTo(const Context &id);  // runs and deletes this component's mutators
    Mutations &Append(const Mutations &src);
    Mutations Detach(const Context &id);
    size_t size() const { return map_ ? map_->size() : 0; }

private:
    std::shared_ptr<std::map<Context, std::vector<MutatorFunc>>> map_;
};
}  // namespace mut

// ---- fitting (stage-to-stage transport)          internal/fitting/fitting.go ----
namespace fitting {
struct Message {  // fitting.go:11-15
    signal::Floating Signal;
    mut::Mutations Mutations;
};
class Fitting {  // fitting.go:17-36
public:
    virtual ~Fitting() = default;
    virtual bool Send(const Context &ctx, Message m) = 0;
    virtual Message Receive(const Context &ctx, bool *ok) = 0;
    virtual void Close() = 0;
};
using New = std::function<std::shared_ptr<Fitting>()>;
std::shared_ptr<Fitting> Sync();   // one slot + closed flag      fitting.go:62-79
std::shared_ptr<Fitting> Async();  // channel of capacity 1       fitting.go:81-104
}  // namespace fitting

// ---- components ------------------------------------------------------------------
struct SignalProperties {  // line.go:38-41
    signal::Frequency SampleRate = 0;
    int Channels = 0;
};

using SourceFunc = std::function<error(signal::Floating &out, int *read)>;                       // pipe.go:47
using ProcessFunc = std::function<error(const signal::Floating &in, signal::Floating &out, int *n)>;  // pipe.go:64
using SinkFunc = std::function<error(const signal::Floating &in)>;                               // pipe.go:80
using StartFunc = std::function<error(const Context &)>;                                         // pipe.go:83
using FlushFunc = std::function<error(const Context &)>;                                         // pipe.go:86

struct out_link {  // line.go:51-54
    std::shared_ptr<fitting::Fitting> sender;
    std::shared_ptr<signal::PoolAllocator> allocator;
};
struct in_link {  // line.go:56-59
    std::shared_ptr<fitting::Fitting> receiver;
    std::shared_ptr<signal::PoolAllocator> allocator;
    void insert(const out_link &o)  // line.go:155-158
    {
        receiver = o.sender;
        allocator = o.allocator;
    }
};

// executor interface                                               run.go:13-18
class executor {
public:
    virtual ~executor() = default;
    virtual error execute(const Context &ctx) = 0;
    virtual error startHook(const Context &ctx) = 0;
    virtual error flushHook(const Context &ctx) = 0;
};

using Destination = std::shared_ptr<struct MutationChan>;  // mutable.Destination (chan cap 1)

struct Source : executor {  // pipe.go:35-43
    Destination dest;
    mut::Context Context;
    ::pipe::SourceFunc SourceFunc;
    ::pipe::StartFunc StartFunc;
    ::pipe::FlushFunc FlushFunc;
    ::pipe::SignalProperties SignalProperties;
    out_link out;
    void connect(int bufferSize, const fitting::New &fn);  // pipe.go:372-377
    error execute(const ::pipe::Context &ctx) override;     // pipe.go:379-413
    error startHook(const ::pipe::Context &ctx) override;
    error flushHook(const ::pipe::Context &ctx) override;
};

// A set of identical Processors -- one per Line -- that advance together: ONE device launch
// per pass for all of them instead of one ProcessFunc call per Line (RunBatched below).
// Slot i belongs to the Processor that carries BatchSlot == i.
struct BatchGroup {
    virtual ~BatchGroup() = default;
    virtual int Slots() const = 0;
    // ins[i] / outs[i]: this pass's buffers of slot i, nullptr for a slot that takes no part
    // (its Line has ended).  processed[i] receives the frames written to outs[i].
    virtual error ProcessLines(const std::vector<const signal::Floating *> &ins,
                               const std::vector<signal::Floating *> &outs, std::vector<int> *processed) = 0;
};

struct Processor : executor {  // pipe.go:52-60
    mut::Context Context;
    ::pipe::ProcessFunc ProcessFunc;
    ::pipe::StartFunc StartFunc;
    ::pipe::FlushFunc FlushFunc;
    ::pipe::SignalProperties SignalProperties;
    in_link in;
    out_link out;
    // set by a batched allocator: the group this Processor advances with (RunBatched only)
    std::shared_ptr<BatchGroup> Batch;
    int BatchSlot = -1;
    void connect(int bufferSize, const fitting::New &fn, const out_link &prev);  // pipe.go:415-421
    error execute(const ::pipe::Context &ctx) override;                          // pipe.go:423-451
    // execute() split around ProcessFunc for the batched pass: everything before the call
    // (receive, mutations, output allocation: pipe.go:424-437) / everything after it
    // (slice, send, free the input: pipe.go:438-450)
    struct Pending {
        fitting::Message m;
        signal::Floating output;
    };
    error batchBegin(const ::pipe::Context &ctx, Pending *pd);
    error batchEnd(const ::pipe::Context &ctx, Pending &pd, int processed, const error &procErr);
    error startHook(const ::pipe::Context &ctx) override;
    error flushHook(const ::pipe::Context &ctx) override;
};

struct Sink : executor {  // pipe.go:69-76
    mut::Context Context;
    ::pipe::SinkFunc SinkFunc;
    ::pipe::StartFunc StartFunc;
    ::pipe::FlushFunc FlushFunc;
    ::pipe::SignalProperties SignalProperties;
    in_link in;
    void connect(int bufferSize, const out_link &prev);  // pipe.go:453-455
    error execute(const ::pipe::Context &ctx) override;  // pipe.go:457-471
    error startHook(const ::pipe::Context &ctx) override;
    error flushHook(const ::pipe::Context &ctx) override;
};

// allocators                                                       line.go:21-35
using SourceAllocatorFunc = std::function<error(mut::Context mctx, int bufferSize, Source *out)>;
using ProcessorAllocatorFunc =
    std::function<error(mut::Context mctx, int bufferSize, SignalProperties input, Processor *out)>;
using SinkAllocatorFunc = std::function<error(mut::Context mctx, int bufferSize, SignalProperties input, Sink *out)>;

struct Line {  // line.go:14-19
    mut::Context Context;
    SourceAllocatorFunc Source;
    std::vector<ProcessorAllocatorFunc> Processors;
    SinkAllocatorFunc Sink;
};

inline std::vector<ProcessorAllocatorFunc> Processors(std::vector<ProcessorAllocatorFunc> p) { return p; }

// pipe.Run: every Line in ONE thread, round robin                  pipe.go:89-103
// Returns nil, a plain error ("error starting ..."), or an ErrorRun rendered as
// an error whose cause chain keeps the original for Is().
error Run(const Context &ctx, int bufferSize, std::vector<Line> lines, ErrorRun *detail = nullptr);

// pipe.Run with a STAGE-MAJOR pass: per pass every live Line's Source runs, then stage p of
// every Line, then every Sink.  Each Line sees exactly the data flow of Run (same buffers, same
// order within the Line, same EOF/flush/error rules); what changes is the interleaving across
// Lines, which lets Processors that share a BatchGroup advance with one device launch.
// Processors without a group execute one by one as in Run.  
// Live edits of a running batched pipe (second half).  In the reference they are
// mutations the executor applies between two passes: Pipe.AddLine -> multiLineExecutor.addRoute
// (pipe.go:260-300, run.go:134-145), Pipe.InsertProcessor -> runtime.insertProcessor +
// multiLineExecutor.startSyncProcessor (pipe.go:302-365, run.go:147-169).  Here an edit names the
// pass it arrives before, which is what a mutation pushed at that moment amounts to.
struct BatchedEdit {
    enum Kind { kAddLine, kInsertProcessor } kind = kAddLine;
    int before_pass = 0;            // 0: after the start hooks, before the first pass
    ::pipe::Line line;              // kAddLine
    int route = 0;                  // kInsertProcessor: index of the Line (initial Lines 0..n-1 in the order
                                    // given, added Lines continue the count) ...
    int pos = 0;                    // ... and the position of the new Processor in its chain
    ProcessorAllocatorFunc alloc;   // kInsertProcessor
};
error RunBatched(const Context &ctx, int bufferSize, std::vector<Line> lines, ErrorRun *detail = nullptr,
                 std::vector<BatchedEdit> edits = {});

// pipe.New + Start + Wait: immutable Line context => one thread per component
// connected by capacity-1 channels; mutable context => sync executor per context.
class Pipe {
public:
    ~Pipe();
    // Start returns a handle to Wait on                             pipe.go:197-214
    std::shared_ptr<struct ErrChan> Start(const Context &ctx, std::vector<mut::Mutation> initializers = {});
    void Push(std::vector<mut::Mutation> mutations);  // pipe.go:243-247 (before / between Starts)

private:
    friend error New(int, std::vector<Line>, std::unique_ptr<Pipe> *);
    struct Impl;
    std::unique_ptr<Impl> impl_;
};
error New(int bufferSize, std::vector<Line> lines, std::unique_ptr<Pipe> *out);  // pipe.go:105-126
error Wait(const std::shared_ptr<struct ErrChan> &errc);                          // pipe.go:249-257

}  // namespace pipe
