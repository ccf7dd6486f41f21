// A Line's Processors slice (line.go:17) as one device-side chain: stage i's
// output feeds stage i+1 without leaving the GPU, with float64 intermediates (the
// element type the reference pipe itself carries between stages: pipe.go:437).
// Only the first stage reads, and only the last stage writes, buffers of the
// handle's I/O dtype -- so a float32 chain rounds once, at the end, exactly like
// `(float)` applied to the oracle's float64 chain.
#include <cstdio>
#include <cstdlib>

#include "chain_fused.hpp"
#include "common.hpp"

namespace pipehip {
namespace {

class Chain final : public pipe_hip_processor {
public:
    std::vector<std::unique_ptr<pipe_hip_processor>> stages;

    int init()
    {
        const size_t n = sizeof(double) * (size_t)cfg.lines * (size_t)cfg.buffer_size *
                         (size_t)cfg.max_batch * (size_t)cfg.channels;
        if (stages.size() > 1) {
            PH_TRY(tmp_[0].alloc(n));
            if (stages.size() > 2)
                PH_TRY(tmp_[1].alloc(n));
        }
        return PIPE_HIP_OK;
    }
    int start(hipStream_t s) override
    {
        if (fused_)
            fused_->drop_state();
        for (auto &st : stages)
            PH_TRY(st->start(s));
        return PIPE_HIP_OK;
    }
    int start_lines(int first, int count, hipStream_t s) override
    {
        if (fused_)
            PH_TRY(fused_->export_state(s));  // the other Lines keep their state
        for (auto &st : stages)
            PH_TRY(st->start_lines(first, count, s));
        return PIPE_HIP_OK;
    }
    int run(const void *d_in, int in_dtype, void *d_out, int out_dtype, int64_t frames,
            hipStream_t s) override
    {
        const void *src = d_in;
        int src_dtype = in_dtype;
        const size_t ns = stages.size();
        // FIR -> biquad (-> gain) on float32 buffers, a call large enough for the overlap-save
        // form: ONE kernel, one read and one write of the buffers (chain_fused.hip)
        last_fused_.valid = false;
        if (!no_fuse_ && !queued_run && fusable(d_in, in_dtype, d_out, out_dtype, frames)) {
            FirFuseView fv{};
            BiquadFuseView bv{};
            double g = 1.0;
            const bool has_gain = ns == 3 && gain_value(stages[2].get(), &g);
            const bool f64 = in_dtype == PIPE_HIP_F64;  // (float64 buffers, pipe.go:394,437: only with PIPE_HIP_PARAM_RELAXED_F64 on both stages)
            fv.f64_stream = f64;
            if (stages[0]->fuse_view_fir(&fv, s, false) && stages[1]->fuse_view_biquad(&bv) && fv.relaxed && bv.relaxed &&
                (!f64 || (fv.relaxed_f64 && bv.relaxed_f64)) &&
                bv.sections <= fused::kMaxFusedSections && fv.ntaps >= 16 && fv.ntaps <= 512) {
                const int64_t L = 1024 - (fv.ntaps - 1 + 31) / 32 * 32;
                const int64_t items = ((frames + L - 1) / L) * ((cfg.channels + 1) / 2) * (int64_t)cfg.lines;
                if (!fused_)
                    fused_.reset(new fused::Plan());
                // Where the one kernel passes the three launches (profiles/r06_chain_small_calls.txt: 64 / 256 / 512 taps,
                // 2 and 8 channels).  Its launch has a floor: 18 - 19 us when every workgroup owns whole Lines (the
                // block-local look-back: Lines a multiple of the CUs, or eight times as many), 29 - 31 us with the global
                // look-back; the staged chain is 17.5 us + the ordered FIR's 2.76 ps a sample and 0.060 ps a sample and tap
                // + the biquad's and the gain's 7 ps a sample.  Whole Lines per workgroup: from two transforms a CU, as
                // before; otherwise where the two lines cross,
                //     frames x Lines x channel pairs x (taps + 116) >= 1.92e8     (256 taps: 32 Lines x 8 ch x 4096)
                // -- until round 6 two transforms a CU for every shape and tap count: 128 Lines x 2 ch through 64 taps
                // took the fused kernel's 30.7 us where the three launches take 20.8.  PIPE_HIP_FIR_OLS_MIN_ITEMS set:
                // that count of transforms alone.
                bool wanted;
                if (fv.min_items >= 0) {
                    wanted = items >= fv.min_items;
                } else {
                    const int cus = fv.cus > 0 ? fv.cus : 256;
                    const bool whole_lines = cfg.lines >= cus && (cfg.lines % cus == 0 || cfg.lines >= 8 * cus);
                    const double pf = (double)frames * (double)cfg.lines * (double)((cfg.channels + 1) / 2);
                    wanted = whole_lines ? items >= 2 * (int64_t)cus : pf * (double)(fv.ntaps + 116) >= 1.92e8 * (double)cus / 256.0;
                }
                if (wanted && fused_->accepts(bv.coeffs, bv.sections, fv.ntaps, frames, s)) {
                    if (!stages[0]->fuse_view_fir(&fv, s, true))  // (history into the fused kernel's layout)
                        return PIPE_HIP_EHIP;
                    PH_TRY(fused_->run(fv, bv, has_gain, g, d_in, d_out, f64, frames, cfg.channels, cfg.lines, s, &timer,
                                       &last_kernel));
                    last_fused_ = FusedCall{d_in, d_out, in_dtype, out_dtype, frames, true};
                    return stages[0]->fuse_commit_fir(s);
                }
            }
        }
        if (fused_)
            PH_TRY(fused_->export_state(s));  // the staged form reads the biquad stage's own array
        last_staged_ = FusedCall{d_in, d_out, in_dtype, out_dtype, frames, true};
        PH_TRY(timer.begin(s));
        int hop = 0;
        for (size_t i = 0; i < ns; ++i) {
            // biquad immediately followed by gain: one pass, the gain applied to the
            // float64 result before it is stored (identical arithmetic, one stage less of
            // float64 traffic through HBM)
            double g = 1.0;
            const bool fold = i + 1 < ns && gain_value(stages[i + 1].get(), &g) &&
                              biquad_set_post_gain(stages[i].get(), true, g);
            const size_t done = fold ? i + 1 : i;
            const bool last = done + 1 == ns;
            void *dst = last ? d_out : tmp_[hop & 1].p;
            const int dst_dtype = last ? out_dtype : (int)PIPE_HIP_F64;
            stages[i]->timer.enable(false);
            // float64 intermediates of a chain that ends in float32 may take the FIR's
            // overlap-save form: its O(1e-16) perturbation stays far below the final ulp
            stages[i]->relaxed_f64_out = !last && out_dtype == PIPE_HIP_F32;
            stages[i]->queued_run = queued_run;
            if (last) {  // the buffer's completion event may ride on the last stage's last launch
                stages[i]->completion = completion;
                completion = nullptr;
            }
            const int rc = stages[i]->run(src, src_dtype, dst, dst_dtype, frames, s);
            if (last && stages[i]->completion) {  // not taken: submit records the event itself
                completion = stages[i]->completion;
                stages[i]->completion = nullptr;
            }
            if (fold)
                biquad_set_post_gain(stages[i].get(), false, 1.0);
            PH_TRY(rc);
            src = dst;
            src_dtype = dst_dtype;
            ++hop;
            i = done;
        }
        PH_TRY(timer.end(s));
        last_kernel = stages.empty() ? "" : stages[0]->last_kernel;
        return PIPE_HIP_OK;
    }
    // PIPE_HIP_PARAM_RESIDENT: a chain of stages each of which can take a queued launch back (FIR, gain, the tile
    // biquad); a run that is queued ahead (queued_run) never takes the fused kernel, whose state lives in tagged
    // slots that rollback_launch() does not reach and whose plan may allocate and synchronise on the way
    bool take_failure_flag() override
    {
        bool f = false;
        for (auto &st : stages)
            f = st->take_failure_flag() || f;
        return f;
    }
    bool armable() const override
    {
        for (auto &st : stages)
            if (!st->armable())
                return false;
        return !stages.empty();
    }
    bool armable_for(int64_t frames, int out_dtype) override
    {
        // (the dtypes a stage sees inside run(): float64 between stages, the chain's own at its end; a gain behind a
        // biquad is folded into the biquad's store)
        const size_t ns = stages.size();
        for (size_t i = 0; i < ns; ++i) {
            double g;
            const bool fold = i + 1 < ns && gain_value(stages[i + 1].get(), &g) && biquad_set_post_gain(stages[i].get(), false, 1.0);
            const size_t done = fold ? i + 1 : i;
            const bool last = done + 1 == ns;
            stages[i]->relaxed_f64_out = !last && out_dtype == PIPE_HIP_F32;
            if (!stages[i]->armable_for(frames, last ? out_dtype : (int)PIPE_HIP_F64))
                return false;
            i = done;
        }
        return ns > 0;
    }
    void rollback_launch() override
    {
        for (auto &st : stages)
            st->rollback_launch();
    }
    void set_window(int first, int count) override
    {
        pipe_hip_processor::set_window(first, count);
        for (auto &st : stages)
            st->set_window(first, count);
    }
    // a mutation addressed to the chain goes to the first stage that owns the parameter
    int set_param(int32_t param, const double *values, int32_t count) override
    {
        if (param == PIPE_HIP_PARAM_DEBUG && count == 2)
            return debug_fail_next((int)values[0], values[1]);
        if (param == PIPE_HIP_PARAM_EXACT || param == PIPE_HIP_PARAM_RELAXED_F64) {  // applies to every stage that has a relaxed form
            int rc = PIPE_HIP_EINVAL;
            for (auto &st : stages)
                if (st->set_param(param, values, count) == PIPE_HIP_OK)
                    rc = PIPE_HIP_OK;
            return rc;
        }
        for (auto &st : stages)
            if (st->set_param(param, values, count) == PIPE_HIP_OK)
                return PIPE_HIP_OK;
        return PIPE_HIP_EINVAL;
    }
    // the fused launch of a synchronous entry has been waited for: if its look-back gave up, run the call again staged
    int settle(hipStream_t s, bool *reran) override
    {
        if (reran)
            *reran = false;
        if (!fused_ || !last_fused_.valid) {
            int rc = fused_ ? fused_->poll_error() : PIPE_HIP_OK;
            for (size_t i = 0; i < stages.size(); ++i) {  // (the staged chain: a stage that ran a look-back form looks
                bool again = false;  // after itself -- its input, the chain's float64 intermediate, is still there)
                const int r = stages[i]->settle(s, &again);
                rc = rc != PIPE_HIP_OK ? rc : r;
                if (again && rc == PIPE_HIP_OK) {
                    rc = rerun_staged(s, i);
                    if (reran)
                        *reran = true;
                    break;
                }
            }
            return rc;
        }
        const FusedCall c = last_fused_;
        last_fused_.valid = false;
        if (fused_->poll_error() == PIPE_HIP_OK)
            return PIPE_HIP_OK;
        PH_TRY(take_back(s));
        static bool said = false;
        if (!said) {
            said = true;
            std::fprintf(stderr, "pipe_hip: a fused chain launch gave up waiting for a predecessor tile; the call was run "
                                 "again on the staged chain (further occurrences are not reported)\n");
        }
        no_fuse_ = true;
        int rc = run(c.d_in, c.in_dtype, c.d_out, c.out_dtype, c.frames, s);
        no_fuse_ = false;
        PH_TRY(rc);
        PH_HIP(hipStreamSynchronize(s));
        if (reran)
            *reran = true;
        bool again = false;
        return settle(s, &again);  // (the staged run's own look-back stages)
    }
    int poll_error() override
    {
        int rc = fused_ ? fused_->poll_error() : PIPE_HIP_OK;
        if (rc != PIPE_HIP_OK && last_fused_.valid) {
            // an asynchronous call (pipe_hip_process_batch) whose buffers are no longer ours: the call cannot be run
            // again from here, but the stream's state can be what it was before it -- the caller may submit it again
            last_fused_.valid = false;
            (void)take_back(stream);
        }
        for (auto &st : stages) {
            const int r = st->poll_error();
            rc = rc != PIPE_HIP_OK ? rc : r;
        }
        return rc;
    }
    int set_stage_param(int32_t stage, int32_t param, const double *values, int32_t count) override
    {
        if (stage < 0 || (size_t)stage >= stages.size())
            return PIPE_HIP_EINVAL;
        return stages[(size_t)stage]->set_param(param, values, count);
    }

    // debug: {tile, limit in microseconds} -- the next fused launch fails the way a preempted predecessor tile
    // would make it fail (tests/test_gpu_chain_fused.py)
    int debug_fail_next(int tile, double limit_us)
    {
        if (!fused_)
            fused_.reset(new fused::Plan());
        fused_->debug_fail_next(tile, limit_us);
        return PIPE_HIP_OK;
    }

private:
    struct FusedCall {
        const void *d_in;
        void *d_out;
        int in_dtype, out_dtype;
        int64_t frames;
        bool valid;
    };
    FusedCall last_fused_{nullptr, nullptr, 0, 0, 0, false}, last_staged_{nullptr, nullptr, 0, 0, 0, false};
    bool no_fuse_ = false;
    // A stage of the staged chain ran its call again (its state is right again, and it rewrote the buffer the next
    // stage reads).  The stages BEHIND it have consumed what it wrote before and advanced on it; running them again
    // needs their state of before the call, which only look-back stages keep.  So: fine when nothing with state
    // sits behind the stage that ran again (FIR -> biquad (-> gain): a gain is stateless and folded into the biquad's
    // own store); any other order (biquad -> FIR, biquad -> biquad) has consumed wrong samples -- reported, loudly,
    // as the device failure it is (pipe.go:438-440: a ProcessFunc error ends the run).
    int rerun_staged(hipStream_t s, size_t stage)
    {
        PH_HIP(hipStreamSynchronize(s));
        for (size_t j = stage + 1; j < stages.size(); ++j) {
            double g;
            if (!gain_value(stages[j].get(), &g))
                return PIPE_HIP_EHIP;
        }
        return PIPE_HIP_OK;
    }
    // the state of before the failed launch: the cascade's from the slot the launch did not write, the FIR's
    // history from the half it did not write
    int take_back(hipStream_t s)
    {
        PH_TRY(fused_->rollback(s));
        stages[0]->rollback_launch();
        return PIPE_HIP_OK;
    }
    bool fusable(const void *d_in, int in_dtype, const void *d_out, int out_dtype, int64_t frames) const
    {
        if (!fused::Plan::enabled() || windowed() || frames <= 0 || !fused::Plan::launchable())
            return false;
        if (stages.size() != 2 && stages.size() != 3)
            return false;
        if (in_dtype != out_dtype)
            return false;
        if (stages.size() == 3) {
            double g;
            if (!gain_value(stages[2].get(), &g))
                return false;
        }
        // (channel pairs are one access; an odd channel count -- the last channel alone in its pair, a mono Line: every
        // frame its own "pair" -- leaves them aligned to an element only)
        const uintptr_t pair = (cfg.channels % 2 ? 1 : 2) * dtype_size(in_dtype);
        return reinterpret_cast<uintptr_t>(d_in) % pair == 0 && reinterpret_cast<uintptr_t>(d_out) % pair == 0;
    }

    DevBuf tmp_[2];
    std::unique_ptr<fused::Plan> fused_;
};

}  // namespace

int make_chain(pipe_hip_processor *const *stages, int32_t n, pipe_hip_processor **out)
{
    if (!stages || n < 1 || n > 16 || !stages[0])
        return PIPE_HIP_EINVAL;
    const pipe_hip_config c0 = stages[0]->cfg;
    for (int i = 0; i < n; ++i) {
        const pipe_hip_processor *s = stages[i];
        if (!s || s->owned_by_chain)
            return PIPE_HIP_EINVAL;
        // every stage must keep rate and channel count (no resampler / mix inside)
        int32_t up = 1, down = 1;
        s->rate(&up, &down);
        if (up != down || s->out_channels() != c0.channels || std::memcmp(&s->cfg, &c0, sizeof c0) != 0)
            return PIPE_HIP_EINVAL;
        if (!s->single_input())
            return PIPE_HIP_EINVAL;
    }
    auto p = std::make_unique<Chain>();
    PH_TRY(p->init_common(&c0));
    for (int i = 0; i < n; ++i) {
        stages[i]->owned_by_chain = true;
        p->stages.emplace_back(stages[i]);
    }
    const int rc = p->init();
    if (rc != PIPE_HIP_OK) {
        for (auto &st : p->stages) {  // hand the stages back on failure
            st->owned_by_chain = false;
            (void)st.release();
        }
        return rc;
    }
    *out = p.release();
    return PIPE_HIP_OK;
}

}  // namespace pipehip
