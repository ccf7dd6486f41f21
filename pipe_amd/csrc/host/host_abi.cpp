// C entry points of the host-side mirror (include/pipe_host.h).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "hip_processors.hpp"
#include "mock.hpp"
#include "pipe_host.h"

using namespace pipe;

namespace {

const error &mockError()
{
    static const error e = NewError("mock error");
    return e;
}
error inj(int flag) { return flag ? mockError() : nullptr; }

uint64_t splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1u) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct LineMocks {
    mock::Source source;
    std::vector<std::unique_ptr<mock::Processor>> procs;  // counters for every processor slot
    std::vector<std::shared_ptr<hip::Handle>> handles;
    mock::Sink sink;
};

// wraps a HIP allocator so the mock counters/hooks observe the stage like the
// reference's tests observe mock.Processor
ProcessorAllocatorFunc counted(ProcessorAllocatorFunc inner, mock::Processor *m)
{
    return [inner, m](mut::Context mctx, int bufferSize, SignalProperties props, Processor *out) -> error {
        if (m->ErrorOnMake)
            return m->ErrorOnMake;
        if (error e = inner(mctx, bufferSize, props, out))
            return e;
        m->Mutability = mctx;
        auto start = out->StartFunc;
        auto flush = out->FlushFunc;
        auto proc = out->ProcessFunc;
        out->StartFunc = [m, start](const Context &c) -> error {
            if (error e = m->Start(c))
                return e;
            return start ? start(c) : nullptr;
        };
        out->FlushFunc = [m, flush](const Context &c) -> error {
            error fe = flush ? flush(c) : nullptr;
            error me = m->Flush(c);
            return me ? me : fe;
        };
        out->ProcessFunc = [m, proc](const signal::Floating &in, signal::Floating &o, int *n) -> error {
            if (m->ErrorOnCall)
                return m->ErrorOnCall;
            if (error e = proc(in, o, n))
                return e;
            m->advance(*n);
            return nullptr;
        };
        return nullptr;
    };
}

pipe_host_counter cnt(const mock::Counter &c, const mock::Starter &s, const mock::Flusher &f)
{
    return pipe_host_counter{c.Messages, c.Samples, s.Started ? 1 : 0, f.Flushed ? 1 : 0};
}

}  // namespace

extern "C" int pipe_host_run(int32_t mode, int32_t buffer_size, int32_t n_lines, const pipe_host_line_desc *descs,
                             pipe_host_line_result *results, pipe_host_error *err, int32_t runs, int32_t device)
{
    if (!descs || !results || !err || n_lines < 1 || buffer_size < 0)
        return 1;
    std::memset(err, 0, sizeof *err);
    std::memset(results, 0, sizeof(*results) * (size_t)n_lines);
    std::vector<std::unique_ptr<LineMocks>> mocks;
    std::vector<Line> lines;           // bound at the start
    std::vector<BatchedEdit> edits;    // RUN_BATCHED: Lines / Processors that arrive while the pipe runs
    std::vector<int> route_of((size_t)n_lines, -1);
    std::vector<mut::Mutation> initializers_src;  // built after binding
    bool any_hip = false;
    hip::Options opt;
    opt.device = device;
    // {ntaps, taps..., nsections, coeffs..., gain}
    auto parse_chain = [](const std::vector<double> &params, std::vector<hip::StageSpec> *st) -> bool {
        size_t p = 0;
        if (params.size() < 3)
            return false;
        const size_t nt = (size_t)params[p++];
        if (p + nt + 1 > params.size())
            return false;
        st->push_back({hip::StageSpec::kFir, std::vector<double>(params.begin() + (long)p, params.begin() + (long)(p + nt))});
        p += nt;
        const size_t ns = (size_t)params[p++];
        if (p + 5 * ns + 1 > params.size())
            return false;
        st->push_back({hip::StageSpec::kBiquad,
                       std::vector<double>(params.begin() + (long)p, params.begin() + (long)(p + 5 * ns))});
        p += 5 * ns;
        st->push_back({hip::StageSpec::kGain, {params[p]}});
        return true;
    };
    // batched mode: a HIP chain with identical parameters at position k of EVERY Line becomes
    // one BatchGroup (one device handle, one launch per pass)
    std::vector<std::vector<ProcessorAllocatorFunc>> batched((size_t)PIPE_HOST_MAX_PROCS);
    std::vector<std::shared_ptr<hip::Handle>> batch_handles((size_t)PIPE_HOST_MAX_PROCS);
    if (mode == PIPE_HOST_MODE_RUN_BATCHED) {
        for (int k = 0; k < descs[0].n_procs && k < PIPE_HOST_MAX_PROCS; ++k) {
            bool same = true;
            for (int i = 0; i < n_lines && same; ++i) {
                const pipe_host_proc_desc &a = descs[0].procs[k];
                same = descs[i].n_procs > k && descs[i].procs[k].kind == PIPE_HOST_PROC_HIP_CHAIN &&
                       descs[i].procs[k].n_params == a.n_params && descs[i].procs[k].params &&
                       std::memcmp(descs[i].procs[k].params, a.params, sizeof(double) * (size_t)a.n_params) == 0;
            }
            if (!same)
                continue;
            std::vector<hip::StageSpec> st;
            const pipe_host_proc_desc &a = descs[0].procs[k];
            if (!parse_chain(std::vector<double>(a.params, a.params + a.n_params), &st))
                return 1;
            batched[(size_t)k] = hip::BatchedChain(st, n_lines, opt, &batch_handles[(size_t)k]);
        }
    }
    for (int i = 0; i < n_lines; ++i) {
        const pipe_host_line_desc &d = descs[i];
        auto m = std::make_unique<LineMocks>();
        m->source.Limit = (int)d.src_limit;
        m->source.Value = d.src_value;
        m->source.Channels = d.src_channels;
        m->source.SampleRate = 48000;
        m->source.ErrorOnCall = inj(d.src_err_on_call);
        m->source.ErrorOnMake = inj(d.src_err_on_make);
        m->source.ErrorOnStart = inj(d.src_err_on_start);
        m->source.ErrorOnFlush = inj(d.src_err_on_flush);
        if (d.src_kind == PIPE_HOST_SRC_SYNTH) {
            const uint64_t seed = d.src_seed;
            m->source.Generator = [seed](int64_t k) {
                return (double)(splitmix64_at(seed, (uint64_t)k) >> 40) * 0x1p-23 - 1.0;
            };
        } else if (d.src_kind == PIPE_HOST_SRC_ARRAY) {
            const double *data = d.src_data;
            m->source.Generator = [data](int64_t k) { return data[k]; };
        }
        m->sink.Discard = d.sink_discard != 0;
        m->sink.ErrorOnCall = inj(d.sink_err_on_call);
        m->sink.ErrorOnMake = inj(d.sink_err_on_make);
        m->sink.ErrorOnStart = inj(d.sink_err_on_start);
        m->sink.ErrorOnFlush = inj(d.sink_err_on_flush);
        Line l;
        l.Source = m->source.Allocator();
        l.Sink = m->sink.Allocator();
        m->handles.resize((size_t)d.n_procs);
        for (int k = 0; k < d.n_procs && k < PIPE_HOST_MAX_PROCS; ++k) {
            const pipe_host_proc_desc &pd = d.procs[k];
            auto pm = std::make_unique<mock::Processor>();
            pm->ErrorOnCall = inj(pd.err_on_call);
            pm->ErrorOnMake = inj(pd.err_on_make);
            pm->ErrorOnStart = inj(pd.err_on_start);
            pm->ErrorOnFlush = inj(pd.err_on_flush);
            std::vector<double> params(pd.params ? pd.params : nullptr, pd.params ? pd.params + pd.n_params : nullptr);
            std::shared_ptr<hip::Handle> *hslot = &m->handles[(size_t)k];
            const size_t before = l.Processors.size();
            switch (pd.kind) {
            case PIPE_HOST_PROC_MOCK:
                l.Processors.push_back(pm->Allocator());
                break;
            case PIPE_HOST_PROC_HIP_COPY:
                l.Processors.push_back(counted(hip::Copy(opt, hslot), pm.get()));
                any_hip = true;
                break;
            case PIPE_HOST_PROC_HIP_GAIN:
                l.Processors.push_back(counted(hip::Gain(params.empty() ? 1.0 : params[0], opt, hslot), pm.get()));
                any_hip = true;
                break;
            case PIPE_HOST_PROC_HIP_FIR:
                l.Processors.push_back(counted(hip::Fir(params, opt, hslot), pm.get()));
                any_hip = true;
                break;
            case PIPE_HOST_PROC_HIP_BIQUAD:
                l.Processors.push_back(counted(hip::Biquad(params, opt, hslot), pm.get()));
                any_hip = true;
                break;
            case PIPE_HOST_PROC_HIP_CHAIN: {
                any_hip = true;
                if (!batched[(size_t)k].empty()) {
                    l.Processors.push_back(batched[(size_t)k][(size_t)i]);
                    break;
                }
                std::vector<hip::StageSpec> st;
                if (!parse_chain(params, &st))
                    return 1;
                l.Processors.push_back(counted(hip::Chain(st, opt, hslot), pm.get()));
                break;
            }
            default:
                return 1;
            }
            if (mode == PIPE_HOST_MODE_RUN_BATCHED && pd.insert_before_pass > 0 && l.Processors.size() > before) {
                // not part of the bound Line: inserted live at the position it would have had
                BatchedEdit e;
                e.kind = BatchedEdit::kInsertProcessor;
                e.before_pass = pd.insert_before_pass;
                e.route = i;  // fixed up below
                e.pos = (int)before;
                e.alloc = l.Processors.back();
                l.Processors.pop_back();
                edits.push_back(std::move(e));
            }
            m->procs.push_back(std::move(pm));
        }
        if (mode == PIPE_HOST_MODE_RUN_BATCHED && d.join_before_pass > 0) {
            BatchedEdit e;
            e.kind = BatchedEdit::kAddLine;
            e.before_pass = d.join_before_pass;
            e.line = std::move(l);
            e.route = -1 - i;  // marks the description it came from
            edits.push_back(std::move(e));
        } else {
            route_of[(size_t)i] = (int)lines.size();
            lines.push_back(std::move(l));
        }
        mocks.push_back(std::move(m));
    }
    {   // route indices: the Lines bound at the start in order, then the added ones in arrival order
        std::stable_sort(edits.begin(), edits.end(),
                         [](const BatchedEdit &a, const BatchedEdit &b) { return a.before_pass < b.before_pass; });
        int next = (int)lines.size();
        for (auto &e : edits)
            if (e.kind == BatchedEdit::kAddLine)
                route_of[(size_t)(-1 - e.route)] = next++;
        for (auto &e : edits)
            if (e.kind == BatchedEdit::kInsertProcessor)
                e.route = route_of[(size_t)e.route];
    }
    if (any_hip)
        hip::UsePinnedPools();

    error run_err;
    bool bind_err = false;
    const Context ctx = Context::Background();
    if (mode == PIPE_HOST_MODE_RUN || mode == PIPE_HOST_MODE_RUN_BATCHED) {
        for (int r = 0; r < (runs < 1 ? 1 : runs) && !run_err; ++r) {
            if (r > 0)
                for (auto &m : mocks)
                    m->source.Reset().Apply();
            run_err = mode == PIPE_HOST_MODE_RUN ? Run(ctx, buffer_size, lines) : RunBatched(ctx, buffer_size, lines, nullptr, edits);
        }
    } else {
        std::unique_ptr<Pipe> p;
        run_err = New(buffer_size, lines, &p);
        bind_err = static_cast<bool>(run_err);
        for (int r = 0; r < (runs < 1 ? 1 : runs) && !run_err; ++r) {
            std::vector<mut::Mutation> init;
            if (r > 0)
                for (auto &m : mocks)
                    init.push_back(m->source.Reset());  // waitPipe(t, p, timeout, source.Reset())
            if (r == 0)
                for (int i = 0; i < n_lines; ++i)
                    for (int k = 0; k < descs[i].n_procs; ++k)
                        if (descs[i].procs[k].mutate_gain && mocks[(size_t)i]->handles[(size_t)k])
                            init.push_back(mut::Mutate(mocks[(size_t)i]->procs[(size_t)k]->Mutability,
                                                       mocks[(size_t)i]->handles[(size_t)k]->SetGain(
                                                           descs[i].procs[k].mutated_gain)));
            run_err = Wait(p->Start(ctx, init));
        }
    }
    if (run_err) {
        err->failed = 1;
        err->is_mock_error = Is(run_err, mockError()) ? 1 : 0;
        err->is_bind_error = bind_err ? 1 : 0;
        std::strncpy(err->message, ErrorString(run_err).c_str(), sizeof(err->message) - 1);
    }
    for (int i = 0; i < n_lines; ++i) {
        LineMocks &m = *mocks[(size_t)i];
        results[i].source = cnt(m.source, m.source, m.source);
        for (size_t k = 0; k < m.procs.size(); ++k)
            results[i].procs[k] = cnt(*m.procs[k], *m.procs[k], *m.procs[k]);
        results[i].sink = cnt(m.sink, m.sink, m.sink);
        if (!m.sink.Discard && m.sink.Values.valid()) {
            const int n = m.sink.Values.Len();
            results[i].sink_values = static_cast<double *>(std::malloc(sizeof(double) * (size_t)(n > 0 ? n : 1)));
            for (int k = 0; k < n; ++k)
                results[i].sink_values[k] = m.sink.Values.Sample(k);
            results[i].sink_values_len = n;
        }
    }
    return 0;
}

extern "C" void pipe_host_free_values(double *values) { std::free(values); }
extern "C" int64_t pipe_host_pool_buffers_created(void) { return pipe::signal::PoolBuffersCreated(); }
