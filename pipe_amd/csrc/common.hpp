// Shared infrastructure of libpipe_hip: the polymorphic Processor handle behind
// the C ABI in include/pipe_hip.h, error plumbing and small RAII helpers.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <mutex>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "pipe_hip.h"

namespace pipehip {

extern thread_local int g_last_hip_error;

#define PH_HIP(call)                                  \
    do {                                              \
        hipError_t e__ = (call);                      \
        if (e__ != hipSuccess) {                      \
            ::pipehip::g_last_hip_error = (int)e__;   \
            (void)hipGetLastError();                  \
            return PIPE_HIP_EHIP;                     \
        }                                             \
    } while (0)

#define PH_TRY(expr)                 \
    do {                             \
        int s__ = (expr);            \
        if (s__ != PIPE_HIP_OK)      \
            return s__;              \
    } while (0)

inline size_t dtype_size(int dtype) { return dtype == PIPE_HIP_F64 ? 8 : 4; }

// hipFree / hipHostFree wait for EVERY queue of the device -- also for one that is parked behind a doorbell only the
// calling thread can ring (PIPE_HIP_PARAM_RESIDENT: the work of the next buffer is queued while the handle's lock is
// held).  While a thread queues such work, g_deferred_frees points at its handle's list: a buffer that grows then
// leaves its old allocation on the list instead of freeing it, and the handle frees the list at its next ordinary
// entry, when nothing of its own is parked (abi.hip: pipe_hip_processor::enter).
struct DeferredFree {
    void *p;
    bool pinned;
};
extern thread_local std::vector<DeferredFree> *g_deferred_frees;
// Before a hipFree / hipHostFree of the calling thread's current device: work another handle has parked behind that
// device's doorbell is rung (it runs on stale input and its owner takes it back), so that the free does not sit out
// the 250 ms until the watchdog does it (abi.hip).
void ring_parked_before_free();

// Device allocation owned by a handle.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n)
    {
        release();
        if (n == 0)
            n = 16;
        PH_HIP(hipMalloc(&p, n));
        bytes = n;
        return PIPE_HIP_OK;
    }
    void release()
    {
        if (p && g_deferred_frees)
            g_deferred_frees->push_back(DeferredFree{p, false});
        else if (p) {
            ring_parked_before_free();
            (void)hipFree(p);
        }
        p = nullptr;
        bytes = 0;
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

// Pinned host allocation (the host half of a DMA staging pair).
struct PinnedBuf {
    void *p = nullptr;
    size_t bytes = 0;
    // coherent: fine-grained memory whose words the host and a RUNNING device queue may hand back and forth
    // (doorbell / completion words); the default serves staging buffers read and written by whole launches
    int alloc(size_t n, bool coherent = false)
    {
        release();
        if (n == 0)
            n = 16;
        PH_HIP(hipHostMalloc(&p, n, coherent ? (hipHostMallocCoherent | hipHostMallocMapped) : hipHostMallocDefault));
        bytes = n;
        return PIPE_HIP_OK;
    }
    void release()
    {
        if (p && g_deferred_frees)
            g_deferred_frees->push_back(DeferredFree{p, true});
        else if (p) {
            ring_parked_before_free();
            (void)hipHostFree(p);
        }
        p = nullptr;
        bytes = 0;
    }
    ~PinnedBuf() { release(); }
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
};

// A parameter upload that never stops the device: two pinned staging slots used in turn, a
// hipMemcpyAsync on the stream the handle's launches go to, an event per slot so that the host only
// waits when it wants a slot whose previous upload has not left yet.  (mutable.Mutation bodies
// run between two buffers of ONE component, pipe.go:433; other Lines on the same device must not
// feel them.)
struct AsyncUpload {
    PinnedBuf slot[2];
    hipEvent_t left[2] = {nullptr, nullptr};
    int next = 0;
    ~AsyncUpload()
    {
        for (hipEvent_t e : left)
            if (e)
                (void)hipEventDestroy(e);
    }
    // host memory for `bytes` of the next upload
    int stage(size_t bytes, void **host)
    {
        if (slot[next].bytes < bytes)
            PH_TRY(slot[next].alloc(bytes));
        if (!left[next])
            PH_HIP(hipEventCreateWithFlags(&left[next], hipEventDisableTiming));
        else
            PH_HIP(hipEventSynchronize(left[next]));  // normally long done
        *host = slot[next].p;
        return PIPE_HIP_OK;
    }
    // queue the copy of what stage() handed out
    int commit(void *dst, size_t bytes, hipStream_t s)
    {
        PH_HIP(hipMemcpyAsync(dst, slot[next].p, bytes, hipMemcpyHostToDevice, s));
        PH_HIP(hipEventRecord(left[next], s));
        next ^= 1;
        return PIPE_HIP_OK;
    }
};

// hipEvent bracket around the dominant kernel of a handle (pipe_hip_set_profiling).
class KernelTimer {
public:
    ~KernelTimer();
    void enable(bool on) { enabled_ = on; }
    bool enabled() const { return enabled_; }
    int begin(hipStream_t s);
    int end(hipStream_t s);
    // events for hipExtLaunchKernelGGL: they are attached to the kernel's own dispatch, so
    // their difference is the kernel's execution time (what rocprofv3 reports), without the
    // marker packets and dispatch gap that begin()/end() around a launch include.  Both are
    // nullptr when profiling is off.
    int pair(hipEvent_t *a, hipEvent_t *b);
    // hands the last pair() back: the launch it was taken for did not happen (never recorded, drain() must not read it)
    void unpair(hipEvent_t a)
    {
        if (a && used_ > 0 && ring_[used_ - 1].a == a)
            used_ -= 1;
    }
    int collect(double *total_ms, int64_t *launches, bool reset);

private:
    int drain();
    struct Pair {
        hipEvent_t a, b;
    };
    std::vector<Pair> ring_;
    size_t used_ = 0;
    bool enabled_ = false;
    bool open_ = false;
    double total_ms_ = 0.0;
    int64_t launches_ = 0;
};

}  // namespace pipehip

// ---- environment switches ------------------------------------------------------------------------------
// The shipped library reads a dozen tuning knobs, once per handle (pipe_hip_processor::Knobs) or once per
// process -- DESIGN.md lists them.  The switches that exist only to A/B kernel variants against each other
// (thresholds under study, alternative forms, traces) are compiled in by `make AB=1` (-DPIPE_HIP_AB,
// lib/libpipe_hip_ab.so) and do not exist in the default build: PH_ENV_AB() is then a null pointer and the
// variable's name is not even in the binary.
#ifdef PIPE_HIP_AB
#define PH_ENV_AB(name) std::getenv(name)
#else
#define PH_ENV_AB(name) (static_cast<const char *>(nullptr))
#endif

// The opaque handle of the C ABI.  One object == one Processor component of a
// Line (pipe.go:49-60) -- or of `lines` identical Lines batched together.
struct pipe_hip_processor {
    pipe_hip_config cfg{};
    // the shipped tuning knobs, read from the environment when the handle is made (init_common)
    struct Knobs {
        int64_t fir_ols_min_items = -1;     // PIPE_HIP_FIR_OLS_MIN_ITEMS: smallest call (1024-point transforms) for the overlap-save FIR (default: by taps and samples, fir.hip ols_wanted)
        int64_t fir_mfma_min_passes = 32;   // PIPE_HIP_FIR_MFMA_MIN_PASSES: smallest call (passes of 1024 frames x 2 ch) for the matrix-pipe FIR
        size_t overlap_min_bytes = (size_t)4 << 20;  // PIPE_HIP_OVERLAP_MIN_BYTES: smallest host call cut into overlapped chunks of Lines
        size_t zero_copy_max = (size_t)1 << 20;      // PIPE_HIP_ZERO_COPY_MAX: largest buffer the kernels read / write in pinned host memory
        int64_t resample_rows_min_blocks = 64;  // PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS: smallest call (blocks of 128 / C rows of a period each) for the row form of the resampler (unset: 6 channels and more from the first block, resampler.hip); negative: never
        bool resample_rows_stereo = false;      //   ... set explicitly, it also admits 2-channel streams (they tie with the wave kernel: profiles/r05_resampler_rows_sweep.txt)
        int bar_upload = -1;                // PIPE_HIP_BAR_UPLOAD: 0 never, 1 also without an HDP flush register, unset: where the device has one
        void read();
    };
    Knobs knobs;
    hipStream_t stream = nullptr;  // the handle's own stream
    hipStream_t batch_stream = nullptr;  // a caller's stream the last device-resident call went to (if not `stream`)
    // StartFunc / FlushFunc order themselves after everything the handle has queued anywhere
    int drain()
    {
        if (batch_stream && batch_stream != stream) {
            // (the caller may have destroyed that stream since: its work is then done or gone)
            if (hipStreamSynchronize(batch_stream) != hipSuccess)
                (void)hipGetLastError();
            batch_stream = nullptr;
        }
        PH_HIP(hipStreamSynchronize(stream));
        return PIPE_HIP_OK;
    }
    pipehip::KernelTimer timer;
    const char *last_kernel = "";

    // host<->device staging for the ProcessFunc form (lazily sized).  Two slots: buffer k + 1 may
    // be submitted while buffer k is still on the device (pipe_hip_submit / pipe_hip_collect)
    struct Staging {
        pipehip::DevBuf d_in, d_out;
        pipehip::PinnedBuf h_in, h_out;
        void *hd_in = nullptr, *hd_out = nullptr;  // device aliases of h_in / h_out (zero-copy path)
        hipEvent_t done = nullptr;
        int32_t out_frames = 0;
        bool zero_copy = false;  // the kernels read / wrote the pinned buffers themselves (no D2H copy to redo after a rerun)
        bool express = false;    // the call completes by a word in pinned memory (express_mail), not by `done`
    };
    Staging stg[2];
    // Set by submit while it queues a buffer: a stage whose LAST device operation for the call is a
    // kernel launch may hand this event to the launch as its stop event (one API call and one
    // barrier packet less than a hipEventRecord behind it) and clears the field; otherwise submit
    // records the event itself.
    hipEvent_t completion = nullptr;
    int submit_slot = 0;   // the slot the next submit fills
    int in_flight = 0;     // buffers submitted and not collected (0..2); the oldest sits in
                           // slot (submit_slot - in_flight) & 1
    pipehip::PinnedBuf line_tab;  // pointer / length tables of process_lines_pinned
    // large host calls: copy streams, per-chunk events and the copy-out worker (abi.hip), made on
    // first use and released by the destructor
    struct Overlap;
    Overlap *overlap = nullptr;
    bool owned_by_chain = false;
    // set by a chain for a stage whose float64 output feeds a chain that ends in float32:
    // the stage may then use a form that is exact to O(1e-16) instead of bit-exact
    bool relaxed_f64_out = false;

    // The Lines one run() advances: [win_first, win_first + win_count) of cfg.lines; the
    // buffers passed to run() then hold only those Lines, packed.  win_count == 0: all Lines
    // (the default).  pipe_hip_process_lines narrows it when the Lines of one pass bring
    // different frame counts (a short read in the middle of a stream, pipe.go:404-406): every
    // Line's state must advance by its OWN frames.
    int win_first = 0, win_count = 0;
    int active_lines() const { return win_count > 0 ? win_count : cfg.lines; }
    bool windowed() const { return win_count > 0 && win_count < cfg.lines; }
    virtual void set_window(int first, int count)
    {
        win_first = count > 0 ? first : 0;
        win_count = count;
    }

    virtual ~pipe_hip_processor();

    // properties of the OUTPUT signal
    virtual int out_channels() const { return cfg.channels; }
    virtual void rate(int32_t *up, int32_t *down) const
    {
        *up = 1;
        *down = 1;
    }
    // frames this stage emits per Line for `in_frames` new input frames
    virtual int64_t out_frames_for(int64_t in_frames) const { return in_frames; }
    // max output frames per Line for the staging buffers
    virtual int64_t max_out_frames(int64_t in_frames) const { return in_frames; }

    // false for the n-input mix, which ProcessFunc's single `in` cannot feed
    virtual bool single_input() const { return true; }

    // zero per-Line state, asynchronously on `s`
    virtual int start(hipStream_t s) = 0;
    // ... of Lines [first, first + count) only: a Line that joins a running batch handle
    // (Pipe.AddLine, pipe.go:260-300) starts from silence without disturbing the others
    virtual int start_lines(int first, int count, hipStream_t s)
    {
        (void)first;
        (void)count;
        (void)s;
        return PIPE_HIP_EINVAL;
    }
    // advance every Line by `frames` frames.  Device pointers, line-major.
    virtual int run(const void *d_in, int in_dtype, void *d_out, int out_dtype, int64_t frames,
                    hipStream_t s) = 0;
    // rate-changing form: emits *out_frames (<= out_cap) frames per Line; the
    // output Line stride is out_cap frames.  Default: out == in frames.
    virtual int run_var(const void *d_in, int in_dtype, int64_t in_frames, void *d_out, int out_dtype,
                        int64_t out_cap, int64_t *out_frames, hipStream_t s)
    {
        if (in_frames > out_cap)
            return PIPE_HIP_ECAP;
        if (out_frames)
            *out_frames = in_frames;
        return run(d_in, in_dtype, d_out, out_dtype, in_frames, s);
    }
    virtual bool fixed_rate() const { return true; }
    virtual int set_param(int32_t param, const double *values, int32_t count)
    {
        (void)param;
        (void)values;
        (void)count;
        return PIPE_HIP_EINVAL;
    }
    // chains only: a parameter of stage `stage`
    virtual int set_stage_param(int32_t stage, int32_t param, const double *values, int32_t count)
    {
        (void)stage;
        (void)param;
        (void)values;
        (void)count;
        return PIPE_HIP_EINVAL;
    }

    // ---- hooks of the fused chain kernel (chain_fused.hip) --------------------------------
    // A FIR stage with an overlap-save plan / a biquad stage hand out what the fused kernel
    // needs; everything else answers false and the chain runs its stages one after the other.
    struct FirFuseView {
        const void *hist;      // current history of all Lines (float32 elements: the stream's type)
        void *hist_new;        // the half the launch writes
        const void *plan;      // ols::Plan::Impl
        const double *taps;    // float64 taps on the device (the direct form's copy)
        int ntaps;
        bool relaxed;          // the stage may use a form that is not bit-exact
        int64_t min_items;     // smallest call (in FFT items) that takes the fused kernel: PIPE_HIP_FIR_OLS_MIN_ITEMS, or -1 = the chain's own rule (chain.hip)
        int cus;               // compute units of the stage's device (that rule's constants are a 256-CU chip's)
        bool f64_stream;       // IN: the chain's buffers are float64 (the history then stays float64)
        bool relaxed_f64;      // PIPE_HIP_PARAM_RELAXED_F64 is set on the stage: float64 RESULTS may be relaxed too
    };
    struct BiquadFuseView {
        double *state;         // [lines][C][S][2]
        const double *coeffs;  // [S][5], host
        int sections;
        bool relaxed;
        bool relaxed_f64;      // PIPE_HIP_PARAM_RELAXED_F64 is set on the stage
    };
    // prepare: the chain WILL launch the fused kernel on this stream (the history goes to its layout)
    virtual bool fuse_view_fir(FirFuseView *, hipStream_t, bool /*prepare*/) { return false; }
    virtual int fuse_commit_fir(hipStream_t) { return PIPE_HIP_EINVAL; }  // the launch wrote hist_new
    virtual bool fuse_view_biquad(BiquadFuseView *) { return false; }
    // device-side failures that cannot be reported by the asynchronous call that caused them
    virtual int poll_error() { return PIPE_HIP_OK; }
    // Called by the SYNCHRONOUS entries once the call's work HAS BEEN WAITED FOR (its event or its stream), while
    // the call's buffers are still theirs: a stage whose launch can fail on the device (the look-back forms: a
    // predecessor tile that never shows up) looks at its flag and, if the launch failed, puts its state back,
    // runs the call again in a form that cannot fail that way, waits for it, and says so in *reran (the entry
    // then moves the results again) -- the reference aborts the whole run on a ProcessFunc error
    // (pipe.go:438-440), and this one is not the stream's fault.  Everything else: a flag read at most.
    virtual int settle(hipStream_t, bool *reran)
    {
        if (reran)
            *reran = false;
        return poll_error();
    }

    // ---- the per-buffer path with the next call's work queued ahead of it (PIPE_HIP_PARAM_RESIDENT) ----
    // armable(): a run() that has been queued can be executed on stale input and its effect dropped --
    // the stage is stateless, or its state is double-buffered and rollback_launch() points it back at the
    // half the dropped launch did not write.  Stages that update state in place answer false.
    virtual bool armable() const { return false; }
    // ... for a call of `frames` frames whose results leave the stage as out_dtype (a stage whose form -- and with it
    // whether the launch can be taken back -- depends on the call's size)
    virtual bool armable_for(int64_t /*frames*/, int /*out_dtype*/) { return armable(); }
    virtual void rollback_launch() {}
    // set around the run() that is queued behind a doorbell: forms that may synchronise, allocate on the way or keep
    // state a rollback_launch() cannot take back (the fused chain kernel) are not chosen for it
    bool queued_run = false;
    // the device-side failure flag of the stage's look-back forms, read and cleared, NOTHING else touched (the
    // queued-ahead path looks at launch k's flag when launch k + 1 is already queued: settle() / poll_error()
    // would take back the wrong launch)
    virtual bool take_failure_flag() { return false; }
    struct Resident {
        // What the measurements allow (scripts/micro/queue_independence.hip): a queue that waits for a
        // doorbell holds up whatever shares its hardware queue, and every OTHER parked queue of the process costs
        // the one that is rung tens of microseconds.  So at most ONE handle per device holds the doorbell, and its
        // work goes to a stream with a hardware queue of its own (one per device, made once, never destroyed).
        bool enabled = false;         // this handle holds its device's doorbell
        pipehip::PinnedBuf mail;      // doorbell word at +0, completion word at +64 (coherent pinned memory)
        unsigned seq = 0;             // the sequence number the queued work waits for
        // idle: nothing queued.  armed: the work of the next buffer waits for doorbell `seq`.  stale: somebody
        // other than the next call rang it (the watchdog, another handle's destruction, this handle's own entries):
        // the work runs on whatever the staging buffer holds; before the handle does anything else it waits for the
        // completion word and takes the launch back.
        enum State : int { kIdle = 0, kArmed = 1, kStale = 2 };
        std::atomic<int> state{kIdle};
        int32_t frames = 0;           // ... for a buffer of this many frames
        int64_t out_frames = 0;
        // `mu` is held by whoever touches the doorbell, the queued work or the staging buffers: the fast path for
        // its whole length, every other entry while it takes queued work back.  Foreign threads (the watchdog,
        // another handle that is about to wait for the whole device) only ever TRY it, ring, mark the work stale and
        // leave: nobody waits for the device while holding a lock that somebody else needs to ring a doorbell.
        std::mutex mu;
        std::chrono::steady_clock::time_point armed_at{};
        int idle_ms = 250;
        std::atomic<int64_t> dropped_by_watchdog{0};  // queued launches run on stale input and dropped: rung by the watchdog
        std::atomic<int64_t> dropped_by_entry{0};     // ... by another entry (a mutation, a short buffer, start, flush, destroy)
        std::vector<pipehip::DeferredFree> frees;     // allocations replaced while work was being queued (freed at the next entry)
        hipStream_t own_stream = nullptr;             // the handle's ordinary stream while `stream` is the device's doorbell stream
        bool failed = false;          // a queued launch failed on the device: the state is unknown until the next StartFunc
        // PIPE_HIP_PARAM_RESIDENT_SHARED (round 6): the handle is one of SEVERAL that queue their next buffer's work on the
        // device's one doorbell queue, in the order they are called (abi.hip: the device's queue lock stands for `mu`)
        bool shared = false;
        unsigned *bell() const { return static_cast<unsigned *>(mail.p); }
        unsigned *done() const { return static_cast<unsigned *>(mail.p) + 16; }
    };
    Resident resident;
    // ---- completion of a per-buffer call by a word in pinned memory (abi.hip: express completion) ----
    // hipStreamWriteValue32 behind the call's last launch, the host spins on the word: 2.6 us less per call than an
    // event (scripts/micro/queue_independence.hip), nothing parked anywhere.  Per staging slot: word at +0 / +64.
    pipehip::PinnedBuf express_mail;
    unsigned express_ticket[2] = {0, 0};
    int express = -1;  // -1: not decided yet, 0: events (PIPE_HIP_COMPLETION_EVENT, or the device cannot), 1: completion words

    int init_common(const pipe_hip_config *c);
    int ensure_staging(int slot = 0);
    int select_device() const;
    // every entry but the resident fast path: the handle's device, and nothing queued behind a doorbell
    int enter();
};

namespace pipehip {

int validate_config(const pipe_hip_config *cfg);

// factories implemented next to their kernels
int make_gain(const pipe_hip_config *cfg, double gain, pipe_hip_processor **out);
int make_fir(const pipe_hip_config *cfg, const double *taps, int32_t ntaps, pipe_hip_processor **out);
int make_biquad(const pipe_hip_config *cfg, const double *coeffs, int32_t nsections,
                pipe_hip_processor **out);
int make_resampler(const pipe_hip_config *cfg, const double *proto, int32_t taps_per_phase,
                   int32_t up, int32_t down, pipe_hip_processor **out);
int make_mix(const pipe_hip_config *cfg, int32_t inputs, pipe_hip_processor **out);
int make_chain(pipe_hip_processor *const *stages, int32_t n, pipe_hip_processor **out);

// n-input mix has its own entry because ProcessFunc has one input
int mix_run(pipe_hip_processor *p, const void *const *d_ins, int32_t n_inputs, void *d_out,
            int64_t frames, hipStream_t s);

int launch_synth_fill(void *d_out, int dtype, uint64_t seed, int64_t first, int64_t n, hipStream_t s);
int launch_gather_rows(const void *const *tab, const int *words, void *dst, int row_words, int lines, hipStream_t s);
int launch_scatter_rows(void *const *tab, const int *words, const void *src, int row_words, int lines, hipStream_t s);

// chain fusion hooks: true when `p` is a gain stage (its current gain in *g) / a biquad stage
bool gain_value(const pipe_hip_processor *p, double *g);
bool biquad_set_post_gain(pipe_hip_processor *p, bool on, double g);

}  // namespace pipehip
