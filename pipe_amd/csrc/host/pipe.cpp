// Host-side execution of bound Lines: the C++ mirror of pipe.go / line.go /
// run.go / merger.go / internal/fitting / mutable for the per-buffer hot path.
// Citations are into /root/reference.
#include "pipe.hpp"

#include <algorithm>
#include <chrono>
#include <random>
#include <stdexcept>

namespace pipe {

// ---- errors ---------------------------------------------------------------------
error NewError(const std::string &msg)
{
    auto e = std::make_shared<ErrorValue>();
    e->msg = msg;
    return e;
}

error Wrap(const std::string &prefix, const error &cause)
{
    auto e = std::make_shared<ErrorValue>();
    e->msg = prefix + ": " + (cause ? cause->msg : std::string("<nil>"));
    e->cause = cause;
    return e;
}

bool Is(const error &err, const error &target)
{
    for (const ErrorValue *e = err.get(); e; e = e->cause.get())
        if (e == target.get())
            return true;
    return false;
}

std::string ErrorString(const error &err) { return err ? err->msg : std::string("<nil>"); }

namespace io {
const error &EOF_()
{
    static const error eof = NewError("EOF");
    return eof;
}
}  // namespace io

// ---- a Go channel of capacity 1 ---------------------------------------------------
// send/recv give up when the context is cancelled, like the select statements in
// fitting.go:81-99 and pusher.go.
template <typename T>
class Chan1 {
public:
    bool send(const Context &ctx, T v)
    {
        std::unique_lock<std::mutex> lk(mu_);
        while (full_ && !closed_) {
            if (ctx.Done())
                return false;
            cv_.wait_for(lk, std::chrono::microseconds(200));
        }
        if (closed_ || ctx.Done())
            return false;
        slot_ = std::move(v);
        full_ = true;
        cv_.notify_all();
        return true;
    }
    // ok == false: closed-and-drained, or cancelled
    T recv(const Context &ctx, bool *ok)
    {
        std::unique_lock<std::mutex> lk(mu_);
        while (!full_ && !closed_) {
            if (ctx.Done()) {
                *ok = false;
                return T();
            }
            cv_.wait_for(lk, std::chrono::microseconds(200));
        }
        if (!full_) {
            *ok = false;
            return T();
        }
        T v = std::move(slot_);
        slot_ = T();
        full_ = false;
        *ok = true;
        cv_.notify_all();
        return v;
    }
    bool try_recv(T *v)
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (!full_)
            return false;
        *v = std::move(slot_);
        slot_ = T();
        full_ = false;
        cv_.notify_all();
        return true;
    }
    void close()
    {
        std::lock_guard<std::mutex> lk(mu_);
        closed_ = true;
        cv_.notify_all();
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    T slot_{};
    bool full_ = false, closed_ = false;
};

struct MutationChan : Chan1<mut::Mutations> {};  // mutable.Destination  pusher.go:29-31

// ---- mutable ------------------------------------------------------------------------
namespace mut {

Context Mutable()
{
    static std::mutex mu;
    static std::mt19937_64 rng{std::random_device{}()};
    std::lock_guard<std::mutex> lk(mu);
    Context c;
    do {
        c.hi = rng();
        c.lo = rng();
    } while (!c.IsMutable());
    return c;
}

Mutation Mutate(const Context &c, MutatorFunc m)
{
    if (!c.IsMutable())
        throw std::logic_error("mutate immutable");  // mutable.go:41-43 panics
    return Mutation{c, std::move(m)};
}

Mutations &Mutations::Put(const Mutation &m)  // mutable.go:61-76
{
    if (!m.ctx.IsMutable())
        return *this;
    if (!map_)
        map_ = std::make_shared<std::map<Context, std::vector<MutatorFunc>>>();
    (*map_)[m.ctx].push_back(m.mutator);
    return *this;
}

error Mutations::ApplyTo(const Context &id)  // mutable.go:78-94
{
    if (!map_ || !id.IsMutable())
        return nullptr;
    auto it = map_->find(id);
    if (it == map_->end())
        return nullptr;
    for (auto &fn : it->second) {
        if (error e = fn())
            return e;
    }
    map_->erase(it);
    return nullptr;
}

Mutations &Mutations::Append(const Mutations &src)  // mutable.go:96-109
{
    if (!map_)
        map_ = std::make_shared<std::map<Context, std::vector<MutatorFunc>>>();
    if (src.map_)
        for (auto &kv : *src.map_) {
            auto &dst = (*map_)[kv.first];
            dst.insert(dst.end(), kv.second.begin(), kv.second.end());
        }
    return *this;
}

Mutations Mutations::Detach(const Context &id)  // mutable.go:111-122
{
    Mutations d;
    if (!map_)
        return d;
    auto it = map_->find(id);
    if (it == map_->end())
        return d;
    d.map_ = std::make_shared<std::map<Context, std::vector<MutatorFunc>>>();
    (*d.map_)[id] = std::move(it->second);
    map_->erase(it);
    return d;
}

}  // namespace mut

// ---- fittings -------------------------------------------------------------------------
namespace fitting {
namespace {

class SyncFitting final : public Fitting {  // fitting.go:38-41,62-79
public:
    bool Send(const Context &, Message m) override
    {
        if (closed_)
            return false;
        message_ = std::move(m);
        return true;
    }
    Message Receive(const Context &, bool *ok) override
    {
        *ok = !closed_;  // a closed fitting hands back the stale message with ok=false
        return message_;
    }
    void Close() override { closed_ = true; }

private:
    bool closed_ = false;
    Message message_;
};

class AsyncFitting final : public Fitting {  // fitting.go:43-45,81-104
public:
    bool Send(const Context &ctx, Message m) override { return ch_.send(ctx, std::move(m)); }
    Message Receive(const Context &ctx, bool *ok) override { return ch_.recv(ctx, ok); }
    void Close() override { ch_.close(); }

private:
    Chan1<Message> ch_;
};

}  // namespace

std::shared_ptr<Fitting> Sync() { return std::make_shared<SyncFitting>(); }
std::shared_ptr<Fitting> Async() { return std::make_shared<AsyncFitting>(); }

}  // namespace fitting

// ---- components: one buffer through one stage ------------------------------------------
static error callHook(const std::function<error(const Context &)> &hook, const Context &ctx)
{
    return hook ? hook(ctx) : nullptr;  // nil hooks are allowed  pipe.go:483-488
}

static std::shared_ptr<signal::PoolAllocator> poolAllocator(const SignalProperties &sp, int bufferSize)
{
    return signal::GetPoolAllocator(sp.Channels, bufferSize, bufferSize);  // pipe.go:490-492
}

void Source::connect(int bufferSize, const fitting::New &fn)
{
    out = out_link{fn(), poolAllocator(SignalProperties, bufferSize)};
}

error Source::execute(const ::pipe::Context &ctx)
{
    mut::Mutations ms;
    // non-blocking poll of the mutation channel, then of cancellation  pipe.go:383-392
    if (dest && dest->try_recv(&ms)) {
        if (error e = ms.ApplyTo(Context))
            return e;
    } else if (ctx.Done()) {
        out.sender->Close();
        return io::EOF_();
    }
    signal::Floating output = out.allocator->Float64();
    int read = 0;
    if (error e = SourceFunc(output, &read)) {
        out.sender->Close();
        output.Free(out.allocator.get());
        return e;
    }
    if (read != output.Length())
        output = output.Slice(0, read);
    if (!out.sender->Send(ctx, fitting::Message{output, ms})) {
        out.sender->Close();
        return io::EOF_();
    }
    return nullptr;
}

error Source::startHook(const ::pipe::Context &ctx) { return callHook(StartFunc, ctx); }
error Source::flushHook(const ::pipe::Context &ctx) { return callHook(FlushFunc, ctx); }

void Processor::connect(int bufferSize, const fitting::New &fn, const out_link &prev)
{
    in.insert(prev);
    out = out_link{fn(), poolAllocator(SignalProperties, bufferSize)};
}

error Processor::execute(const ::pipe::Context &ctx)
{
    bool ok = false;
    fitting::Message m = in.receiver->Receive(ctx, &ok);
    if (!ok) {
        out.sender->Close();
        return io::EOF_();
    }
    // `defer m.Signal.Free(p.in.allocator)`: the input goes back to the UPSTREAM pool
    struct Defer {
        signal::Floating &s;
        signal::PoolAllocator *p;
        ~Defer() { s.Free(p); }
    } free_input{m.Signal, in.allocator.get()};

    if (error e = m.Mutations.ApplyTo(Context))
        return e;
    signal::Floating output = out.allocator->Float64();
    int processed = 0;
    if (error e = ProcessFunc(m.Signal, output, &processed)) {
        out.sender->Close();
        return e;  // the output buffer is not freed on this path (pipe.go:438-440)
    }
    if (processed != out.allocator->Length)
        output = output.Slice(0, processed);
    if (!out.sender->Send(ctx, fitting::Message{output, m.Mutations})) {
        out.sender->Close();
        output.Free(out.allocator.get());
        return io::EOF_();
    }
    return nullptr;
}

error Processor::batchBegin(const ::pipe::Context &ctx, Pending *pd)
{
    bool ok = false;
    pd->m = in.receiver->Receive(ctx, &ok);
    if (!ok) {
        out.sender->Close();
        return io::EOF_();
    }
    if (error e = pd->m.Mutations.ApplyTo(Context)) {
        pd->m.Signal.Free(in.allocator.get());
        return e;
    }
    pd->output = out.allocator->Float64();
    return nullptr;
}

error Processor::batchEnd(const ::pipe::Context &ctx, Pending &pd, int processed, const error &procErr)
{
    struct Defer {
        signal::Floating &s;
        signal::PoolAllocator *p;
        ~Defer() { s.Free(p); }
    } free_input{pd.m.Signal, in.allocator.get()};
    if (procErr) {
        out.sender->Close();
        return procErr;
    }
    signal::Floating output = pd.output;
    if (processed != out.allocator->Length)
        output = output.Slice(0, processed);
    if (!out.sender->Send(ctx, fitting::Message{output, pd.m.Mutations})) {
        out.sender->Close();
        output.Free(out.allocator.get());
        return io::EOF_();
    }
    return nullptr;
}

error Processor::startHook(const ::pipe::Context &ctx) { return callHook(StartFunc, ctx); }
error Processor::flushHook(const ::pipe::Context &ctx) { return callHook(FlushFunc, ctx); }

void Sink::connect(int, const out_link &prev) { in.insert(prev); }

error Sink::execute(const ::pipe::Context &ctx)
{
    bool ok = false;
    fitting::Message m = in.receiver->Receive(ctx, &ok);
    if (!ok)
        return io::EOF_();
    struct Defer {
        signal::Floating &s;
        signal::PoolAllocator *p;
        ~Defer() { s.Free(p); }
    } free_input{m.Signal, in.allocator.get()};
    if (error e = m.Mutations.ApplyTo(Context))
        return e;
    return SinkFunc(m.Signal);
}

error Sink::startHook(const ::pipe::Context &ctx) { return callHook(StartFunc, ctx); }
error Sink::flushHook(const ::pipe::Context &ctx) { return callHook(FlushFunc, ctx); }

// ---- binding                                                        line.go:62-118 ----
namespace {

mut::Context componentContext(const mut::Context &lineCtx)  // line.go:160-165
{
    return lineCtx.IsMutable() ? lineCtx : mut::Mutable();
}

struct route {  // line.go:44-49
    mut::Context context;
    std::shared_ptr<Source> source;
    std::vector<std::shared_ptr<Processor>> processors;
    std::shared_ptr<Sink> sink;

    void connect(int bufferSize)  // line.go:92-104
    {
        fitting::New fn = context.IsMutable() ? fitting::New(fitting::Sync) : fitting::New(fitting::Async);
        source->connect(bufferSize, fn);
        out_link prev = source->out;
        for (auto &p : processors) {
            p->connect(bufferSize, fn, prev);
            prev = p->out;
        }
        sink->connect(bufferSize, prev);
    }
};

// Line.route: run the allocators in order, threading SignalProperties; the
// component's Context is overwritten with the pipe-chosen one (line.go:133,142,151)
error bindLine(const Line &l, int bufferSize, std::shared_ptr<route> *out)
{
    auto r = std::make_shared<route>();
    r->context = l.Context;
    r->source = std::make_shared<Source>();
    mut::Context c = componentContext(l.Context);
    if (!l.Source)
        return Wrap("source", NewError("nil allocator"));
    if (error e = l.Source(c, bufferSize, r->source.get()))
        return Wrap("source", e);
    r->source->Context = c;
    SignalProperties prev = r->source->SignalProperties;
    for (auto &alloc : l.Processors) {
        auto p = std::make_shared<Processor>();
        c = componentContext(l.Context);
        if (error e = alloc(c, bufferSize, prev, p.get()))
            return Wrap("processor", e);
        p->Context = c;
        prev = p->SignalProperties;
        r->processors.push_back(p);
    }
    r->sink = std::make_shared<Sink>();
    c = componentContext(l.Context);
    if (!l.Sink)
        return Wrap("sink", NewError("nil allocator"));
    if (error e = l.Sink(c, bufferSize, prev, r->sink.get()))
        return Wrap("sink", e);
    r->sink->Context = c;
    *out = r;
    return nullptr;
}

error joinErrors(const std::vector<error> &errs)  // execErrors.ret()  error.go:41-57
{
    if (errs.empty())
        return nullptr;
    if (errs.size() == 1)
        return errs[0];
    std::string msg = "multiple errors:";
    for (auto &e : errs)
        msg += " [" + ErrorString(e) + "]";
    auto v = std::make_shared<ErrorValue>();
    v->msg = msg;
    v->cause = errs[0];
    return v;
}

// lineExecutor: the stages of one Line in one thread                 run.go:20-74
struct lineExecutor final : executor {
    int routeIdx = 0;
    int started = 0;
    std::vector<std::shared_ptr<executor>> executors;

    error execute(const Context &ctx) override  // run.go:37-52
    {
        error err;
        for (int i = 0; i < started; ++i) {
            err = executors[(size_t)i]->execute(ctx);
            if (!err)
                continue;
            if (err == io::EOF_())
                continue;  // keep executing so that EOF propagates downstream
            return err;
        }
        return err;
    }
    error flushHook(const Context &ctx) override  // run.go:54-62
    {
        std::vector<error> errs;
        for (int i = 0; i < started; ++i)
            if (error e = executors[(size_t)i]->flushHook(ctx))
                errs.push_back(e);
        return joinErrors(errs);
    }
    error startHook(const Context &ctx) override  // run.go:64-74
    {
        for (auto &e : executors) {
            if (error err = e->startHook(ctx))
                return err;
            ++started;
        }
        return nullptr;
    }
};

std::shared_ptr<lineExecutor> makeLineExecutor(const std::shared_ptr<route> &r, Destination d, int idx)
{
    auto le = std::make_shared<lineExecutor>();  // route.executor  line.go:106-118
    le->routeIdx = idx;
    r->source->dest = std::move(d);
    le->executors.push_back(r->source);
    for (auto &p : r->processors)
        le->executors.push_back(p);
    le->executors.push_back(r->sink);
    return le;
}

// multiLineExecutor: many Lines, one thread, one buffer per Line per pass  run.go:28-132
struct multiLineExecutor final : executor {
    std::vector<std::shared_ptr<lineExecutor>> executors;

    error flushHook(const Context &ctx) override  // run.go:101-110
    {
        std::vector<error> errs;
        for (auto &l : executors)
            if (error e = l->flushHook(ctx))
                errs.push_back(e);
        return joinErrors(errs);
    }
    error startHook(const Context &ctx) override  // run.go:76-99
    {
        error startErr;
        for (auto &l : executors) {
            if (error e = l->startHook(ctx)) {
                startErr = e;
                break;
            }
        }
        if (!startErr)
            return nullptr;
        error err = Wrap("error starting lines", startErr);
        if (error flushErr = flushHook(ctx)) {  // flush what did start
            auto v = std::make_shared<ErrorValue>();
            v->msg = "error flushing lines: " + ErrorString(flushErr) + " during start error: " + ErrorString(err);
            v->cause = flushErr;
            return v;
        }
        return err;
    }
    error execute(const Context &ctx) override  // run.go:112-132
    {
        error err;
        for (size_t i = 0; i < executors.size();) {
            err = executors[i]->execute(ctx);
            if (!err) {
                ++i;
                continue;
            }
            if (err == io::EOF_()) {
                if (error flushErr = executors[i]->flushHook(ctx))
                    return flushErr;
                executors.erase(executors.begin() + (long)i);
                if (!executors.empty())
                    continue;
            }
            return err;
        }
        return nullptr;
    }
};

// stageMajorExecutor: the multiLineExecutor pass turned inside out (stage-major instead of
// Line-major) so that the Processors of a BatchGroup meet in one call.  Start / flush / EOF
// removal follow multiLineExecutor (run.go:76-132).
struct stageMajorExecutor final : executor {
    struct entry {
        std::shared_ptr<route> r;
        std::shared_ptr<lineExecutor> le;
    };
    std::vector<entry> lines;
    // live edits: applied between passes, like the mutations of run.go:134-169
    std::vector<BatchedEdit> edits;
    int pass = 0, bufferSize = 0, nextRoute = 0;
    mut::Context mctx;

    // Pipe.AddLine for a sync Line joining this executor: bind, connect, start, append
    // (pipe.go:260-300; multiLineExecutor.addRoute run.go:134-145)
    error addLine(const Context &ctx, Line l)
    {
        l.Context = mctx;
        std::shared_ptr<route> r;
        if (error err = bindLine(l, bufferSize, &r))
            return Wrap("error adding line", err);
        r->connect(bufferSize);
        auto le = makeLineExecutor(r, nullptr, nextRoute++);
        if (error err = le->startHook(ctx))
            return Wrap("line failed to start", err);
        lines.push_back({r, le});
        return nullptr;
    }
    // Pipe.InsertProcessor on a sync Line: allocate against the previous stage's properties,
    // connect behind its out, hand the new out to the next component, start, splice
    // (runtime.insertProcessor pipe.go:314-333; startSyncProcessor run.go:147-169)
    error insertProcessor(const Context &ctx, int routeIdx, int pos, const ProcessorAllocatorFunc &alloc)
    {
        for (auto &l : lines) {
            if (l.le->routeIdx != routeIdx)
                continue;
            route &r = *l.r;
            if (pos < 0 || (size_t)pos > r.processors.size())
                return NewError("failed to insert processor: position out of range");
            const SignalProperties prevProps = pos == 0 ? r.source->SignalProperties : r.processors[(size_t)pos - 1]->SignalProperties;
            const out_link prevOut = pos == 0 ? r.source->out : r.processors[(size_t)pos - 1]->out;
            auto proc = std::make_shared<Processor>();
            const mut::Context c = componentContext(r.context);
            if (error err = alloc(c, bufferSize, prevProps, proc.get()))
                return Wrap("failed to insert processor", err);
            proc->Context = c;
            proc->connect(bufferSize, fitting::New(fitting::Sync), prevOut);
            if ((size_t)pos < r.processors.size())
                r.processors[(size_t)pos]->in.insert(proc->out);
            else
                r.sink->in.insert(proc->out);
            if (error err = proc->startHook(ctx))
                return Wrap("error starting processor", err);
            r.processors.insert(r.processors.begin() + pos, proc);
            l.le->executors.insert(l.le->executors.begin() + pos + 1, proc);
            l.le->started++;
            return nullptr;
        }
        return NewError("failed to insert processor: no such line");
    }
    error applyEdits(const Context &ctx)
    {
        for (auto &e : edits) {
            if (e.before_pass != pass)
                continue;
            if (error err = e.kind == BatchedEdit::kAddLine ? addLine(ctx, e.line)
                                                            : insertProcessor(ctx, e.route, e.pos, e.alloc))
                return err;
        }
        ++pass;
        return nullptr;
    }

    error flushHook(const Context &ctx) override
    {
        std::vector<error> errs;
        for (auto &l : lines)
            if (error e = l.le->flushHook(ctx))
                errs.push_back(e);
        return joinErrors(errs);
    }
    error startHook(const Context &ctx) override
    {
        error startErr;
        for (auto &l : lines) {
            if (error e = l.le->startHook(ctx)) {
                startErr = e;
                break;
            }
        }
        if (!startErr)
            return nullptr;
        error err = Wrap("error starting lines", startErr);
        if (error flushErr = flushHook(ctx)) {
            auto v = std::make_shared<ErrorValue>();
            v->msg = "error flushing lines: " + ErrorString(flushErr) + " during start error: " + ErrorString(err);
            v->cause = flushErr;
            return v;
        }
        return err;
    }

    // Line i hit EOF before stage `from`: let the EOF travel through its remaining stages
    // (run.go:44-46), flush it and drop it (run.go:120-128).
    error retire(const Context &ctx, size_t i, size_t from)
    {
        entry &l = lines[i];
        for (size_t p = from; p < l.r->processors.size(); ++p) {
            error e = l.r->processors[p]->execute(ctx);
            if (e && e != io::EOF_())
                return e;
        }
        if (from <= l.r->processors.size()) {
            error e = l.r->sink->execute(ctx);
            if (e && e != io::EOF_())
                return e;
        }
        if (error flushErr = l.le->flushHook(ctx))
            return flushErr;
        lines.erase(lines.begin() + (long)i);
        return nullptr;
    }

    struct pending {
        Processor *proc;
        Processor::Pending pd;
    };

    error execute(const Context &ctx) override
    {
        if (!edits.empty())
            if (error err = applyEdits(ctx))
                return err;
        for (size_t i = 0; i < lines.size();) {
            error err = lines[i].r->source->execute(ctx);
            if (!err) {
                ++i;
                continue;
            }
            if (err != io::EOF_())
                return err;
            if (error e = retire(ctx, i, 0))
                return e;
        }
        size_t depth = 0;
        for (auto &l : lines)
            depth = std::max(depth, l.r->processors.size());
        for (size_t p = 0; p < depth; ++p) {
            std::vector<std::pair<BatchGroup *, std::vector<pending>>> groups;
            for (size_t i = 0; i < lines.size();) {
                auto &procs = lines[i].r->processors;
                if (p >= procs.size()) {
                    ++i;
                    continue;
                }
                Processor &pr = *procs[p];
                error err;
                if (pr.Batch) {
                    pending pe{&pr, {}};
                    err = pr.batchBegin(ctx, &pe.pd);
                    if (!err) {
                        auto it = std::find_if(groups.begin(), groups.end(),
                                               [&](const auto &g) { return g.first == pr.Batch.get(); });
                        if (it == groups.end()) {
                            groups.emplace_back(pr.Batch.get(), std::vector<pending>{});
                            it = groups.end() - 1;
                        }
                        it->second.push_back(std::move(pe));
                    }
                } else {
                    err = pr.execute(ctx);
                }
                if (!err) {
                    ++i;
                    continue;
                }
                if (err != io::EOF_())
                    return err;
                if (error e = retire(ctx, i, p + 1))
                    return e;
            }
            std::vector<Processor *> ended;
            error abortErr;  // first failure of this stage: the remaining groups are not processed,
                             // but every pending still gets its batchEnd (input freed, sender
                             // closed) like the deferred Free of pipe.go:431
            for (auto &g : groups) {
                const size_t slots = (size_t)g.first->Slots();
                std::vector<const signal::Floating *> ins(slots, nullptr);
                std::vector<signal::Floating *> outs(slots, nullptr);
                std::vector<int> processed(slots, 0);
                error procErr;
                for (auto &pe : g.second) {
                    const size_t s = (size_t)pe.proc->BatchSlot;
                    if (s >= slots || ins[s])
                        procErr = NewError("batch group: bad or duplicate slot");
                    else {
                        ins[s] = &pe.pd.m.Signal;
                        outs[s] = &pe.pd.output;
                    }
                }
                if (!procErr && abortErr)
                    procErr = abortErr;
                if (!procErr)
                    procErr = g.first->ProcessLines(ins, outs, &processed);
                error first;
                for (auto &pe : g.second) {
                    const size_t s = (size_t)pe.proc->BatchSlot;
                    error e = pe.proc->batchEnd(ctx, pe.pd, s < slots ? processed[s] : 0, procErr);
                    if (e == io::EOF_())
                        ended.push_back(pe.proc);
                    else if (e && !first)
                        first = e;
                }
                if (first && !abortErr)
                    abortErr = first;
            }
            if (abortErr)
                return abortErr;
            for (Processor *pr : ended) {  // a Send refused (context done): EOF for that Line
                for (size_t i = 0; i < lines.size(); ++i)
                    if (p < lines[i].r->processors.size() && lines[i].r->processors[p].get() == pr) {
                        if (error e = retire(ctx, i, p + 1))
                            return e;
                        break;
                    }
            }
        }
        for (size_t i = 0; i < lines.size();) {
            error err = lines[i].r->sink->execute(ctx);
            if (!err) {
                ++i;
                continue;
            }
            if (err != io::EOF_())
                return err;
            if (error e = retire(ctx, i, lines[i].r->processors.size() + 1))
                return e;
        }
        return lines.empty() ? io::EOF_() : nullptr;
    }
};

// run(): sync context                                               run.go:198-224
error runSync(const Context &ctx, executor &e, ErrorRun *detail)
{
    if (error errStart = e.startHook(ctx))
        return Wrap("error starting", errStart);
    error errExec;
    while (!errExec)
        errExec = e.execute(ctx);
    if (errExec == io::EOF_())
        errExec = nullptr;
    else
        errExec = Wrap("error running", errExec);
    error errFlush = e.flushHook(ctx);  // deferred
    if (!errFlush && !errExec)
        return nullptr;
    ErrorRun er{errExec, errFlush ? Wrap("error flushing", errFlush) : nullptr};
    if (detail)
        *detail = er;
    auto v = std::make_shared<ErrorValue>();  // ErrorRun.Error()  error.go:17-27
    v->msg = "pipe error: " + ErrorString(er.ErrExec) + "; " + ErrorString(er.ErrFlush);
    v->cause = er.ErrExec ? er.ErrExec : er.ErrFlush;
    return v;
}

}  // namespace

error Run(const Context &ctx, int bufferSize, std::vector<Line> lines, ErrorRun *detail)
{
    multiLineExecutor e;
    const mut::Context mctx = mut::Mutable();  // forces sync fittings  pipe.go:92-94
    for (size_t i = 0; i < lines.size(); ++i) {
        lines[i].Context = mctx;
        std::shared_ptr<route> r;
        if (error err = bindLine(lines[i], bufferSize, &r))
            return err;
        r->connect(bufferSize);
        e.executors.push_back(makeLineExecutor(r, nullptr, (int)i));
    }
    return runSync(ctx, e, detail);
}

error RunBatched(const Context &ctx, int bufferSize, std::vector<Line> lines, ErrorRun *detail,
                 std::vector<BatchedEdit> edits)
{
    stageMajorExecutor e;
    const mut::Context mctx = mut::Mutable();
    e.edits = std::move(edits);
    e.bufferSize = bufferSize;
    e.mctx = mctx;
    e.nextRoute = (int)lines.size();
    for (size_t i = 0; i < lines.size(); ++i) {
        lines[i].Context = mctx;
        std::shared_ptr<route> r;
        if (error err = bindLine(lines[i], bufferSize, &r))
            return err;
        r->connect(bufferSize);
        e.lines.push_back({r, makeLineExecutor(r, nullptr, (int)i)});
    }
    return runSync(ctx, e, detail);
}

// ---- async pipe: New / Start / Wait ---------------------------------------------------
struct ErrChan {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    error err;
};

struct Pipe::Impl {
    int bufferSize = 0;
    std::vector<std::shared_ptr<route>> routes;
    // pusher: component Context -> Destination, pending Mutations per Destination  pusher.go
    std::map<mut::Context, Destination> destinations;
    std::map<MutationChan *, std::pair<Destination, mut::Mutations>> pending;
    std::mutex push_mu;
    Context run_ctx;
    std::vector<std::thread> threads;
    std::thread supervisor;

    void put(const std::vector<mut::Mutation> &ms)
    {
        for (auto &m : ms) {
            auto it = destinations.find(m.ctx);
            if (it == destinations.end())
                throw std::logic_error("unknown mutable context");  // pusher.go panics
            auto &slot = pending[it->second.get()];
            slot.first = it->second;
            slot.second.Put(m);
        }
    }
    void push(const Context &ctx)
    {
        for (auto &kv : pending)
            if (!kv.second.second.nil() && kv.second.second.size())
                kv.second.first->send(ctx, kv.second.second);
        pending.clear();
    }
    void join()
    {
        for (auto &t : threads)
            if (t.joinable())
                t.join();
        threads.clear();
        if (supervisor.joinable())
            supervisor.join();
    }
};

Pipe::~Pipe()
{
    if (impl_) {
        impl_->run_ctx.Cancel();
        impl_->join();
    }
}

error New(int bufferSize, std::vector<Line> lines, std::unique_ptr<Pipe> *out)
{
    if (lines.empty())
        throw std::logic_error("pipe without lines");  // pipe.go:108-110 panics
    auto p = std::unique_ptr<Pipe>(new Pipe());
    p->impl_.reset(new Pipe::Impl());
    p->impl_->bufferSize = bufferSize;
    for (auto &l : lines) {
        std::shared_ptr<route> r;
        if (error e = bindLine(l, bufferSize, &r))
            return e;
        p->impl_->routes.push_back(r);
    }
    *out = std::move(p);
    return nullptr;
}

void Pipe::Push(std::vector<mut::Mutation> mutations)
{
    std::lock_guard<std::mutex> lk(impl_->push_mu);
    impl_->put(mutations);
    impl_->push(impl_->run_ctx);
}

std::shared_ptr<ErrChan> Pipe::Start(const Context &parent, std::vector<mut::Mutation> initializers)
{
    Impl &I = *impl_;
    I.join();  // a finished pipe may be started again (pipe_test.go:108-131)
    I.run_ctx = Context();
    const Context ctx = I.run_ctx;
    auto errc = std::make_shared<ErrChan>();

    // newRuntime: sync routes share one executor per Line context, async routes
    // get one executor per component                                 pipe.go:128-184
    std::vector<std::shared_ptr<executor>> executors;
    std::map<mut::Context, std::shared_ptr<multiLineExecutor>> sync_groups;
    I.destinations.clear();
    I.pending.clear();
    for (size_t idx = 0; idx < I.routes.size(); ++idx) {
        auto &r = I.routes[idx];
        r->connect(I.bufferSize);  // pipe.go:201-203
        if (r->context.IsMutable()) {
            auto it = sync_groups.find(r->context);
            if (it == sync_groups.end()) {
                auto mle = std::make_shared<multiLineExecutor>();
                Destination d = std::make_shared<MutationChan>();
                I.destinations[r->context] = d;
                it = sync_groups.emplace(r->context, mle).first;
                executors.push_back(mle);
            }
            it->second->executors.push_back(makeLineExecutor(r, I.destinations[r->context], (int)idx));
        } else {
            Destination d = std::make_shared<MutationChan>();
            r->source->dest = d;
            I.destinations[r->source->Context] = d;
            executors.push_back(r->source);
            for (auto &p : r->processors) {
                I.destinations[p->Context] = d;
                executors.push_back(p);
            }
            I.destinations[r->sink->Context] = d;
            executors.push_back(r->sink);
        }
    }
    {
        std::lock_guard<std::mutex> lk(I.push_mu);
        I.put(initializers);  // pipe.go:205-206
        I.push(ctx);
    }

    // errorMerger: each executor in its own thread, first error wins   merger.go, run.go:171-196
    struct Shared {
        std::mutex mu;
        error first;
        std::atomic<int> live{0};
    };
    auto sh = std::make_shared<Shared>();
    sh->live = (int)executors.size();
    auto report = [sh, ctx](const error &e) {
        std::lock_guard<std::mutex> lk(sh->mu);
        if (!sh->first)
            sh->first = e;
        ctx.Cancel();  // pipe.go:233: cancel on first error
    };
    for (auto &e : executors) {
        I.threads.emplace_back([e, ctx, sh, report]() {
            if (error err = e->startHook(ctx)) {  // run.go:177-180
                report(Wrap("error starting", err));
            } else {
                error x;
                while (!x)
                    x = e->execute(ctx);
                if (x != io::EOF_())
                    report(Wrap("error running", x));
                if (error f = e->flushHook(ctx))  // deferred  run.go:181-185
                    report(Wrap("error flushing", f));
            }
            sh->live.fetch_sub(1);
        });
    }
    I.supervisor = std::thread([sh, errc, parent, ctx]() {
        while (sh->live.load() > 0) {
            if (parent.Done())
                ctx.Cancel();
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        std::lock_guard<std::mutex> lk(errc->mu);
        errc->err = sh->first;
        errc->done = true;
        errc->cv.notify_all();
    });
    return errc;
}

error Wait(const std::shared_ptr<ErrChan> &errc)
{
    std::unique_lock<std::mutex> lk(errc->mu);
    errc->cv.wait(lk, [&] { return errc->done; });
    return errc->err;
}

}  // namespace pipe
