// The extern "C" surface declared in include/pipe_hip.h, plus the parts of the
// handle that every Processor kind shares: device selection, the handle's stream,
// pinned/device staging for the host-pointer ProcessFunc form, and the hipEvent
// bracket used for live kernel timing.
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"

namespace pipehip {

thread_local int g_last_hip_error = 0;
thread_local std::vector<DeferredFree> *g_deferred_frees = nullptr;

// ---- KernelTimer -------------------------------------------------------------
KernelTimer::~KernelTimer()
{
    for (auto &p : ring_) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
}

int KernelTimer::drain()
{
    for (size_t i = 0; i < used_; ++i) {
        PH_HIP(hipEventSynchronize(ring_[i].b));
        float ms = 0.f;
        PH_HIP(hipEventElapsedTime(&ms, ring_[i].a, ring_[i].b));
        total_ms_ += (double)ms;
        launches_ += 1;
    }
    used_ = 0;
    return PIPE_HIP_OK;
}

int KernelTimer::begin(hipStream_t s)
{
    if (!enabled_)
        return PIPE_HIP_OK;
    if (used_ == ring_.size()) {
        if (ring_.size() >= 1024) {
            PH_TRY(drain());
        } else {
            Pair p{};
            PH_HIP(hipEventCreate(&p.a));
            PH_HIP(hipEventCreate(&p.b));
            ring_.push_back(p);
        }
    }
    PH_HIP(hipEventRecord(ring_[used_].a, s));
    open_ = true;
    return PIPE_HIP_OK;
}

int KernelTimer::pair(hipEvent_t *a, hipEvent_t *b)
{
    *a = nullptr;
    *b = nullptr;
    if (!enabled_)
        return PIPE_HIP_OK;
    if (used_ == ring_.size()) {
        if (ring_.size() >= 1024) {
            PH_TRY(drain());
        } else {
            Pair p{};
            PH_HIP(hipEventCreate(&p.a));
            PH_HIP(hipEventCreate(&p.b));
            ring_.push_back(p);
        }
    }
    *a = ring_[used_].a;
    *b = ring_[used_].b;
    used_ += 1;
    return PIPE_HIP_OK;
}

int KernelTimer::end(hipStream_t s)
{
    if (!enabled_ || !open_)
        return PIPE_HIP_OK;
    PH_HIP(hipEventRecord(ring_[used_].b, s));
    used_ += 1;
    open_ = false;
    return PIPE_HIP_OK;
}

int KernelTimer::collect(double *total_ms, int64_t *launches, bool reset)
{
    PH_TRY(drain());
    if (total_ms)
        *total_ms = total_ms_;
    if (launches)
        *launches = launches_;
    if (reset) {
        total_ms_ = 0.0;
        launches_ = 0;
    }
    return PIPE_HIP_OK;
}

int validate_config(const pipe_hip_config *c)
{
    if (!c)
        return PIPE_HIP_EINVAL;
    if (c->buffer_size < 1 || c->channels < 1 || c->channels > 64 || c->lines < 1 ||
        c->max_batch < 1)
        return PIPE_HIP_EINVAL;
    if (c->dtype != PIPE_HIP_F32 && c->dtype != PIPE_HIP_F64)
        return PIPE_HIP_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return PIPE_HIP_ENODEV;
    }
    if (c->device < 0 || c->device >= n)
        return PIPE_HIP_ENODEV;
    return PIPE_HIP_OK;
}

}  // namespace pipehip

using namespace pipehip;

namespace {
// ---- uploads through the large BAR ---------------------------------------------------------------------
// What a device offers for CPU stores straight into its memory: the large-BAR attribute and the address
// of its HDP flush register, asked once PER DEVICE (handles of one process may sit on several GPUs).
struct BarInfo {
    bool known = false, large_bar = false;
    volatile unsigned *hdp_flush = nullptr;
};
BarInfo bar_info(int dev)
{
    static std::mutex mu;
    static BarInfo info[64];
    std::lock_guard<std::mutex> lk(mu);
    BarInfo &b = info[dev & 63];
    if (!b.known) {
        b.known = true;
        int v = 0;
        hipDeviceProp_t prop;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeIsLargeBar, dev) == hipSuccess && v != 0 &&
            hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            b.large_bar = true;
            b.hdp_flush = prop.hdpMemFlushCntl;
        } else {
            (void)hipGetLastError();
        }
    }
    return b;
}
// Is [p, p + bytes) mapped readable AND writable in this process?  The large-BAR attribute is a device
// property; an allocation the runtime did not map for the host (or sits in ROCr's reserved PROT_NONE range,
// where mincore() succeeds and a store faults) must not be written to.  /proc/self/maps says what a store
// will find without trying one.
bool host_writable(const void *p, size_t bytes)
{
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + bytes;
    std::FILE *f = std::fopen("/proc/self/maps", "r");
    if (!f)
        return false;
    char line[512];
    uintptr_t covered = lo;  // the range is covered up to here by writable mappings (maps are sorted)
    while (covered < hi && std::fgets(line, sizeof line, f)) {
        unsigned long long a = 0, b = 0;
        char perms[8] = {0};
        if (std::sscanf(line, "%llx-%llx %7s", &a, &b, perms) != 3)
            continue;
        if ((uintptr_t)b <= covered)
            continue;
        if ((uintptr_t)a > covered)
            break;  // a hole
        if (perms[0] != 'r' || perms[1] != 'w')
            break;
        covered = (uintptr_t)b;
    }
    std::fclose(f);
    return covered >= hi;
}
// The rows of a chunk have been stored: make them visible to the device before the launch's doorbell.
// The store fence orders the write-combined stores on the CPU side only; behind the BAR they pass through
// the device's HDP write path, which is flushed by a write to its flush register (read back: the write has
// landed) -- what the ROCm runtime does after its own copies into device memory.
inline void bar_publish(volatile unsigned *hdp_flush)
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_sfence();
#else
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
    if (hdp_flush) {
        *hdp_flush = 1u;
        (void)*hdp_flush;
    }
}
}  // namespace

// ---- large host calls: H2D(k + 1) | kernel(k) | D2H(k - 1) -------------------------------------
// A call that brings many Lines (cfg.lines = L, one pipe buffer each: tens of MB) used to run
// memcpy -> H2D -> kernels -> D2H -> memcpy strictly one after the other, the kernels 1 % of it.
// Lines share no state (run.go:112-132), so the call is cut into chunks of whole Lines, each a
// window of the handle (set_window): the caller's thread copies chunk k + 1 into pinned staging
// while the DMA engine carries chunk k to the device on its own stream, the kernels of chunk k - 1
// run on the handle's stream and the results of chunk k - 2 come back on a third; a worker thread
// waits for every chunk's download and copies it out to the caller's buffers.  The same structure
// as the capacity-1 channels between a pipe's stages (fitting.go:56-60), inside one call.
struct pipe_hip_processor::Overlap {
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<hipEvent_t> ev;  // [3 * chunk]: uploaded, computed, downloaded

    // One call at a time (the entry points are synchronous): what its tasks need.
    struct Call {
        pipe_hip_processor *p;
        int first, count, per;
        int32_t frames;
        size_t row_in, row_out, nchunks;
        char *h_in, *d_in, *h_out, *d_out;
        const std::function<const void *(int)> *in_of;
        const std::function<void *(int)> *out_of;
        bool bar = false;    // the CPU stores the rows straight into device memory (large BAR): no upload DMA
        volatile unsigned *hdp_flush = nullptr;  // the device's HDP flush register (bar_publish)
        std::mutex enq_mu;   // the handle is not thread-safe: one chunk is queued at a time
        int rc = PIPE_HIP_OK;
        size_t finished = 0;  // chunks copied out (or given up on)
        bool trace = false;
        std::chrono::steady_clock::time_point t0;
        std::vector<double> tr;
        double us() const { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
    };

    // Copy threads: the caller plus `extra` workers.  A chunk is two tasks -- IN: copy its rows into
    // pinned staging, queue upload / kernels / download; OUT: wait for the download, copy the rows to
    // the caller's buffers.  A thread prefers an OUT whose download has landed, then an IN, and only
    // blocks on a download when nothing else is left: the host copies (about 18 GB/s per thread on
    // this box) are the slowest part of the call, several threads share them.
    std::vector<std::thread> ths;
    std::mutex mu;
    std::condition_variable cv, idle;
    std::deque<size_t> q_in, q_out;
    Call *call = nullptr;
    bool stop = false;
    int device = 0;
    const void *bar_checked = nullptr;  // the staging allocation host_writable() was last asked about
    size_t bar_checked_bytes = 0;
    bool bar_ok = false;

    static int copy_threads()
    {
        static const char *e = std::getenv("PIPE_HIP_COPY_THREADS");  // (once per process)
        int n = e ? std::atoi(e) : 4;
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && n > hw)
            n = hw;
        return n < 1 ? 1 : (n > 16 ? 16 : n);
    }

    int task_in(Call &c, size_t k)
    {
        const int l0 = (int)k * c.per, n = l0 + c.per <= c.count ? c.per : c.count - l0;
        const size_t io = c.row_in * (size_t)l0, oo = c.row_out * (size_t)l0;
        char *dst_rows = (c.bar ? c.d_in : c.h_in) + io;
        for (int i = 0; i < n; ++i) {
            const void *src = (*c.in_of)(c.first + l0 + i);
            if (src)
                std::memcpy(dst_rows + c.row_in * (size_t)i, src, c.row_in);
            else  // a Line that has ended rides along as silence, its state is dead
                std::memset(dst_rows + c.row_in * (size_t)i, 0, c.row_in);
        }
        if (c.bar)
            bar_publish(c.hdp_flush);
        if (c.trace)
            c.tr[k * 5] = c.us();
        pipe_hip_processor *p = c.p;
        std::lock_guard<std::mutex> lk(c.enq_mu);
        hipEvent_t up = ev[3 * k], done = ev[3 * k + 1], down = ev[3 * k + 2];
        if (!c.bar) {
            PH_HIP(hipMemcpyAsync(c.d_in + io, c.h_in + io, c.row_in * (size_t)n, hipMemcpyHostToDevice, s_in));
            PH_HIP(hipEventRecord(up, s_in));
            PH_HIP(hipStreamWaitEvent(p->stream, up, 0));
        }
        p->set_window(c.first + l0, (c.first + l0 == 0 && n == p->cfg.lines) ? 0 : n);
        int64_t produced = c.frames;
        PH_TRY(p->run_var(c.d_in + io, p->cfg.dtype, c.frames, c.d_out + oo, p->cfg.dtype, c.frames, &produced, p->stream));
        PH_HIP(hipEventRecord(done, p->stream));
        PH_HIP(hipStreamWaitEvent(s_out, done, 0));
        PH_HIP(hipMemcpyAsync(c.h_out + oo, c.d_out + oo, c.row_out * (size_t)n, hipMemcpyDeviceToHost, s_out));
        PH_HIP(hipEventRecord(down, s_out));
        if (c.trace)
            c.tr[k * 5 + 1] = c.us();
        return PIPE_HIP_OK;
    }
    int task_out(Call &c, size_t k)
    {
        const int l0 = (int)k * c.per, n = l0 + c.per <= c.count ? c.per : c.count - l0;
        const size_t oo = c.row_out * (size_t)l0;
        if (c.trace)
            c.tr[k * 5 + 2] = c.us();
        PH_HIP(hipEventSynchronize(ev[3 * k + 2]));
        if (c.trace)
            c.tr[k * 5 + 3] = c.us();
        for (int i = 0; i < n; ++i) {
            void *dst = (*c.out_of)(c.first + l0 + i);
            if (dst)
                std::memcpy(dst, c.h_out + oo + c.row_out * (size_t)i, c.row_out);
        }
        if (c.trace)
            c.tr[k * 5 + 4] = c.us();
        return PIPE_HIP_OK;
    }
    // run tasks until both queues are empty (called with `lk` held; returns with it held)
    void drain(std::unique_lock<std::mutex> &lk)
    {
        while (call && (!q_in.empty() || !q_out.empty())) {
            Call &c = *call;
            bool out = false;
            size_t k = 0;
            if (!q_out.empty() && (q_in.empty() || hipEventQuery(ev[3 * q_out.front() + 2]) == hipSuccess)) {
                out = true;
                k = q_out.front();
                q_out.pop_front();
            } else {
                (void)hipGetLastError();  // (hipErrorNotReady from the query)
                k = q_in.front();
                q_in.pop_front();
            }
            lk.unlock();
            const int rc = out ? task_out(c, k) : task_in(c, k);
            lk.lock();
            if (rc != PIPE_HIP_OK && c.rc == PIPE_HIP_OK)
                c.rc = rc;
            if (!out && rc == PIPE_HIP_OK) {
                q_out.push_back(k);
                cv.notify_one();
            } else if (++c.finished == c.nchunks) {
                idle.notify_all();
            }
        }
    }
    void loop()
    {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [this] { return stop || (call && (!q_in.empty() || !q_out.empty())); });
            if (stop)
                return;
            drain(lk);
        }
    }
    // the caller's side: all chunks of one call
    int run(Call &c)
    {
        std::unique_lock<std::mutex> lk(mu);
        const int want = copy_threads() - 1;
        while ((int)ths.size() < want)
            ths.emplace_back([this] { loop(); });
        call = &c;
        for (size_t k = 0; k < c.nchunks; ++k)
            q_in.push_back(k);
        cv.notify_all();
        drain(lk);
        idle.wait(lk, [&] { return c.finished == c.nchunks; });
        call = nullptr;
        return c.rc;
    }
    int events(size_t n)
    {
        while (ev.size() < n) {
            hipEvent_t e = nullptr;
            PH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ev.push_back(e);
        }
        return PIPE_HIP_OK;
    }
    ~Overlap()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            cv.notify_all();
        }
        for (std::thread &t : ths)
            t.join();
        for (hipEvent_t e : ev)
            (void)hipEventDestroy(e);
        if (s_in)
            (void)hipStreamDestroy(s_in);
        if (s_out)
            (void)hipStreamDestroy(s_out);
    }
};

// ---- shared handle plumbing ---------------------------------------------------
// ---- PIPE_HIP_PARAM_RESIDENT: the next buffer's work queued on the device ahead of its call ------------------
// Round 4 parked a wait packet (hipStreamWaitValue32) on every resident handle's own stream.  What that does on this
// runtime (scripts/micro/queue_independence.hip, profiles/r05_queue_independence.txt):
//  * HIP deals streams onto FOUR hardware queues; with 16 streams parked only the 4 at the heads of the queues ever
//    answer their doorbell -- a handle's kernels sit behind another handle's wait until a watchdog rings that one
//    (the async host loop: one 250 ms / 10 s rescue after the other -- GPUTEST_r04's 1200 s);
//  * a stream made with hipExtStreamCreateWithCUMask has a hardware queue of its own (16 of 16 and 40 of 40 answer),
//    but making one after another one was DESTROYED hangs now and then, so such a stream is made once and kept;
//  * every other parked queue costs the one that is rung: doorbell -> completion 9 us alone, 42 us next to ONE other
//    parked queue, 91 us next to two, 108 next to six -- against 12 us for launch + completion word with nothing parked.
// So: ONE doorbell per device.  The handle that holds it runs on the device's doorbell stream (own hardware queue,
// made on first use, never destroyed); a second handle that asks is answered PIPE_HIP_EBUSY and stays on the plain
// path, which completes by a word in pinned memory as well (express completion, below) and parks nothing.
// Nobody waits for the device while holding a lock another thread needs in order to ring: foreign threads (watchdog,
// another handle about to free memory or to wait for the whole device) try the handle's lock, ring, mark the work
// stale and leave; the owner waits for its own stale work and takes it back at its next entry.
namespace {
constexpr int kMaxDevices = 64;
constexpr size_t kMaxSharers = 16;
struct Door {
    pipe_hip_processor *owner = nullptr;  // the handle that holds this device's doorbell (exclusive: PIPE_HIP_PARAM_RESIDENT)
    hipStream_t stream = nullptr;         // hardware queue of its own; made once, never destroyed
    // PIPE_HIP_PARAM_RESIDENT_SHARED: handles that share the doorbell queue, and whose work is parked in it, oldest first.
    // `qmu` is held for the whole of a sharing handle's call (the calls are not concurrent in the mode this is for);
    // foreign threads -- the watchdog, a handle about to free memory -- only TRY it.  Lock order: g_door_mu, then qmu.
    std::vector<pipe_hip_processor *> sharers;
    std::deque<pipe_hip_processor *> order;
    std::mutex *qmu = new std::mutex;
    std::chrono::steady_clock::time_point last_call{};
};
thread_local int t_holds_qmu = -1;  // device whose qmu the calling thread holds
// (never destroyed: the watchdog thread and the exit hook may still look at them while statics are torn down)
std::mutex &g_door_mu = *new std::mutex;  // guards g_door[].owner; lock order: g_door_mu, then TRY a handle's resident.mu
Door *g_door = new Door[kMaxDevices];

// ---- PIPE_HIP_STALL_DUMP_MS=<ms>: where every thread inside the library stands, printed by the watchdog when one of
// them has stood in the same place for that long (a debugging aid for hangs that only a GPU box shows: no debugger
// there).  Markers are two relaxed stores each; without the variable nothing is ever printed.
struct StallSlot {
    std::atomic<const char *> at{nullptr};
    std::atomic<const void *> who{nullptr};
    std::atomic<long long> since_us{0};
};
constexpr int kStallSlots = 256;
StallSlot *g_stall = new StallSlot[kStallSlots];
std::atomic<int> g_stall_used{0};
thread_local StallSlot *t_stall = nullptr;
inline long long stall_now_us()
{
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct At {  // RAII marker: "this thread is at `name` on behalf of handle `who`"; nests
    const char *prev_at = nullptr;
    const void *prev_who = nullptr;
    long long prev_since = 0;
    At(const char *name, const void *who)
    {
        if (!t_stall) {
            const int i = g_stall_used.fetch_add(1, std::memory_order_relaxed);
            t_stall = &g_stall[i < kStallSlots ? i : kStallSlots - 1];
        }
        prev_at = t_stall->at.load(std::memory_order_relaxed);
        prev_who = t_stall->who.load(std::memory_order_relaxed);
        prev_since = t_stall->since_us.load(std::memory_order_relaxed);
        t_stall->who.store(who, std::memory_order_relaxed);
        t_stall->since_us.store(stall_now_us(), std::memory_order_relaxed);
        t_stall->at.store(name, std::memory_order_release);
    }
    ~At()
    {
        t_stall->who.store(prev_who, std::memory_order_relaxed);
        t_stall->since_us.store(prev_since, std::memory_order_relaxed);
        t_stall->at.store(prev_at, std::memory_order_release);
    }
};
void stall_report_if_stuck()  // (the watchdog thread, g_door_mu held)
{
    static const long long limit_us = [] {
        const char *e = std::getenv("PIPE_HIP_STALL_DUMP_MS");
        return e ? std::atoll(e) * 1000 : 0LL;
    }();
    static int dumps = 0;
    static long long last = 0;
    if (limit_us <= 0 || dumps >= 6)
        return;
    const long long now = stall_now_us();
    if (now - last < 2000000)
        return;
    bool stuck = false;
    const int n = std::min(g_stall_used.load(), kStallSlots);
    for (int i = 0; i < n; ++i)
        if (g_stall[i].at.load(std::memory_order_acquire) && now - g_stall[i].since_us.load() > limit_us)
            stuck = true;
    if (!stuck)
        return;
    last = now;
    ++dumps;
    std::fprintf(stderr, "[pipe_hip stall] ---- dump %d ----\n", dumps);
    for (int i = 0; i < n; ++i)
        if (const char *a = g_stall[i].at.load(std::memory_order_acquire))
            std::fprintf(stderr, "[pipe_hip stall] thread slot %d: %s (handle %p) for %.1f ms\n", i, a, g_stall[i].who.load(),
                         (now - g_stall[i].since_us.load()) / 1000.0);
    for (int d = 0; d < kMaxDevices; ++d) {
        Door &D = g_door[d];
        if (!D.owner && D.sharers.empty())
            continue;
        std::unique_lock<std::mutex> ql(*D.qmu, std::try_to_lock);
        std::fprintf(stderr, "[pipe_hip stall] device %d: owner %p, %zu sharers, queue lock %s, ms since the last shared call %.1f\n", d,
                     (void *)D.owner, D.sharers.size(), ql.owns_lock() ? "free" : "HELD",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - D.last_call).count());
        for (pipe_hip_processor *q : D.sharers)  // (racy reads when the lock is held by somebody: a debugging aid)
            std::fprintf(stderr, "[pipe_hip stall]   sharer %p: state %d seq %u bell %u done %u frames %d\n", (void *)q,
                         q->resident.state.load(), q->resident.seq, q->resident.mail.p ? *q->resident.bell() : 0u,
                         q->resident.mail.p ? *q->resident.done() : 0u, (int)q->resident.frames);
        if (ql.owns_lock())
            for (pipe_hip_processor *q : D.order)
                std::fprintf(stderr, "[pipe_hip stall]   queued: %p\n", (void *)q);
    }
    std::fflush(stderr);
}

// Wait for a word in coherent pinned memory to become k: pause-spin for the length of a per-buffer kernel, then
// yield the core, then sleep in 50 us steps (a host core spins per call in flight, for at most ~50 us).
bool wait_word(const unsigned *w, unsigned k, std::chrono::milliseconds limit)
{
    if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == k)
        return true;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; ++spins) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == k)
            return true;
        if ((spins & 31u) != 0)
            continue;
        const auto dt = std::chrono::steady_clock::now() - t0;
        if (dt > limit)
            return false;
        if (dt > std::chrono::milliseconds(2))
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        else if (dt > std::chrono::microseconds(50))
            std::this_thread::yield();
    }
}

// The stream a handle's queued work sits on.  The exclusive holder of the doorbell runs EVERYTHING on the device's doorbell
// stream (`stream` is that stream while it holds the doorbell: nobody else ever parks anything there).  A sharer keeps
// its own stream in `stream` for everything but its queued work: whatever blocks on the shared stream outside the
// queue's lock -- the hipStreamSynchronize of pipe_hip_start / pipe_hip_flush / drain() -- waits behind other handles'
// parked doorbells WHILE HOLDING THE RUNTIME'S LOCK ON THAT STREAM, and the handle that could ring them (it holds the
// queue's lock, so the watchdog cannot) then blocks in its own next launch onto the stream: seen as 2 hangs in 6 runs of
// the async host loop (a thread per component; profiles/r06_shared_queue_async_hang.txt).
hipStream_t queue_stream(const pipe_hip_processor *p)
{
    return p->resident.shared ? g_door[p->cfg.device].stream : p->stream;
}
// the completion word of doorbell k (the bell HAS been rung: the stream cannot wait for anything of ours any more)
int resident_wait(pipe_hip_processor *p, unsigned k)
{
    At at("resident_wait: the completion word", p);
    if (wait_word(p->resident.done(), k, std::chrono::milliseconds(10000)))
        return PIPE_HIP_OK;
    // ten seconds: a device busy with somebody else's work, or gone.  The runtime's own wait ends either way.
    At at2("resident_wait: 10 s over, hipStreamSynchronize(queue_stream)", p);
    if (hipStreamSynchronize(queue_stream(p)) != hipSuccess) {
        g_last_hip_error = (int)hipGetLastError();
        return PIPE_HIP_EHIP;
    }
    return __atomic_load_n(p->resident.done(), __ATOMIC_ACQUIRE) == k ? PIPE_HIP_OK : PIPE_HIP_EHIP;
}
constexpr int kResidentUnsupported = -1000;  // resident_arm: the stream wait could not be queued (not an ABI status)
// ---- the shared doorbell queue (PIPE_HIP_PARAM_RESIDENT_SHARED); the device's qmu is held by the caller ----
// every parked entry is rung, its owner's work marked stale (it runs on whatever its staging buffer holds and is
// taken back by its owner): nothing is waited for
void shared_ring_all(Door &D, bool by_watchdog)
{
    using RS = pipe_hip_processor::Resident;
    while (!D.order.empty()) {
        pipe_hip_processor *q = D.order.front();
        D.order.pop_front();
        if (q->resident.state.load(std::memory_order_acquire) == RS::kArmed) {
            __atomic_store_n(q->resident.bell(), q->resident.seq, __ATOMIC_RELEASE);
            q->resident.state.store(RS::kStale, std::memory_order_release);
            (by_watchdog ? q->resident.dropped_by_watchdog : q->resident.dropped_by_entry).fetch_add(1, std::memory_order_relaxed);
        }
    }
}
// the entries AHEAD of p's are rung and marked stale (the prediction of the call order failed), then p's own is rung:
// true when p had an entry in the queue
bool shared_ring_through(Door &D, pipe_hip_processor *p)
{
    using RS = pipe_hip_processor::Resident;
    while (!D.order.empty()) {
        pipe_hip_processor *q = D.order.front();
        D.order.pop_front();
        __atomic_store_n(q->resident.bell(), q->resident.seq, __ATOMIC_RELEASE);
        if (q == p)
            return true;
        if (q->resident.state.load(std::memory_order_acquire) == RS::kArmed) {
            q->resident.state.store(RS::kStale, std::memory_order_release);
            q->resident.dropped_by_entry.fetch_add(1, std::memory_order_relaxed);
        }
    }
    return false;
}
// queue the work of a buffer of `frames` frames behind the next doorbell value (resident.mu held, state idle)
int resident_arm(pipe_hip_processor *p, int32_t frames)
{
    pipe_hip_processor::Resident &R = p->resident;
    pipe_hip_processor::Staging &g = p->stg[0];
    const unsigned k = R.seq + 1;
    const hipStream_t qs = queue_stream(p);
    if (hipStreamWaitValue32(qs, R.bell(), k, hipStreamWaitValueEq, 0xFFFFFFFFu) != hipSuccess) {
        (void)hipGetLastError();  // (a platform without stream memory operations: nothing was queued)
        return kResidentUnsupported;
    }
    // From here on the stream waits for doorbell value k: whatever fails below, somebody must ring it -- and nothing
    // below may wait for the device (no synchronisation, no hipFree: g_deferred_frees, queued_run).
    R.seq = k;
    R.frames = frames;
    R.armed_at = std::chrono::steady_clock::now();
    R.state.store(pipe_hip_processor::Resident::kArmed, std::memory_order_release);
    int64_t out_frames = frames;
    g_deferred_frees = &R.frees;
    p->queued_run = true;
    const int rc = p->run_var(g.hd_in, p->cfg.dtype, frames, g.hd_out, p->cfg.dtype, frames, &out_frames, qs);
    p->queued_run = false;
    g_deferred_frees = nullptr;
    R.out_frames = out_frames;
    const hipError_t we = hipStreamWriteValue32(qs, R.done(), k, 0);
    if (we != hipSuccess || rc != PIPE_HIP_OK) {
        // the queued work is incomplete (or no completion word will be written for k): release the wait now, drain the
        // stream the slow way and take back whatever part of the launch was queued
        if (we != hipSuccess)
            (void)hipGetLastError();
        if (R.shared)
            shared_ring_all(g_door[p->cfg.device], false);  // (other handles' work ahead of ours in the queue: the stream could not drain)
        __atomic_store_n(R.bell(), k, __ATOMIC_RELEASE);
        (void)hipStreamSynchronize(qs);  // (shared: everything parked has been rung, and the queue's lock is held)
        __atomic_store_n(R.done(), k, __ATOMIC_RELEASE);  // (what the missing store would have written)
        p->rollback_launch();
        R.state.store(pipe_hip_processor::Resident::kIdle, std::memory_order_release);
        return rc != PIPE_HIP_OK ? rc : PIPE_HIP_EHIP;
    }
    return PIPE_HIP_OK;
}
// Take back what is queued (resident.mu held): ring if nobody has, wait for the stale run, point the state back.
int resident_cancel_locked(pipe_hip_processor *p)
{
    pipe_hip_processor::Resident &R = p->resident;
    using RS = pipe_hip_processor::Resident;
    int st = R.state.load(std::memory_order_acquire);
    if (st == RS::kIdle)
        return PIPE_HIP_OK;
    if (st == RS::kArmed) {
        __atomic_store_n(R.bell(), R.seq, __ATOMIC_RELEASE);
        R.dropped_by_entry.fetch_add(1, std::memory_order_relaxed);
    }
    const int rc = resident_wait(p, R.seq);
    R.state.store(RS::kIdle, std::memory_order_release);
    PH_TRY(rc);
    p->rollback_launch();
    return PIPE_HIP_OK;
}
// Shared queue: EVERYTHING parked on the device is taken back (qmu held) -- rung, waited for, pointed back -- so that
// whatever the calling entry does next on the doorbell stream finds the queue empty; what the queued runs replaced is freed.
int shared_cancel_all_locked(Door &D)
{
    using RS = pipe_hip_processor::Resident;
    shared_ring_all(D, false);
    int rc = PIPE_HIP_OK;
    for (pipe_hip_processor *q : D.sharers) {
        pipe_hip_processor::Resident &R = q->resident;
        if (R.state.load(std::memory_order_acquire) == RS::kStale) {
            const int rq = resident_wait(q, R.seq);
            R.state.store(RS::kIdle, std::memory_order_release);
            if (rq == PIPE_HIP_OK)
                q->rollback_launch();
            else
                rc = rq;
        }
    }
    for (pipe_hip_processor *q : D.sharers) {
        for (const DeferredFree &f : q->resident.frees)
            (void)(f.pinned ? hipHostFree(f.p) : hipFree(f.p));
        q->resident.frees.clear();
    }
    return rc;
}
struct QmuHold {  // the device's queue lock, and the note that this thread holds it
    std::unique_lock<std::mutex> lk;
    int prev;
    explicit QmuHold(Door &D, int device) : lk(*D.qmu, std::defer_lock), prev(t_holds_qmu)
    {
        {
            At at("waiting for the device's queue lock", nullptr);
            lk.lock();
        }
        t_holds_qmu = device;
    }
    ~QmuHold() { t_holds_qmu = prev; }
};
int resident_cancel(pipe_hip_processor *p)
{
    if (!p->resident.mail.p)
        return PIPE_HIP_OK;
    if (p->resident.shared) {
        Door &D = g_door[p->cfg.device];
        QmuHold hold(D, p->cfg.device);
        return shared_cancel_all_locked(D);
    }
    std::lock_guard<std::mutex> lk(p->resident.mu);
    PH_TRY(resident_cancel_locked(p));
    // (nothing of this handle is parked now, and it is the only handle of its device that ever parks anything: what
    // it replaced while it queued work can be freed without waiting for a doorbell)
    for (const DeferredFree &f : p->resident.frees)
        (void)(f.pinned ? hipHostFree(f.p) : hipFree(f.p));
    p->resident.frees.clear();
    return PIPE_HIP_OK;
}
// A foreign thread's part: ring, mark stale, leave (g_door_mu held).  Never waits; skips a handle whose owner is busy
// with it (the owner rings its own doorbells).
void resident_ring_foreign(pipe_hip_processor *p, bool by_watchdog, bool only_if_idle_too_long)
{
    pipe_hip_processor::Resident &R = p->resident;
    using RS = pipe_hip_processor::Resident;
    if (R.state.load(std::memory_order_acquire) != RS::kArmed)
        return;
    std::unique_lock<std::mutex> hl(R.mu, std::try_to_lock);
    if (!hl.owns_lock() || R.state.load(std::memory_order_acquire) != RS::kArmed)
        return;
    if (only_if_idle_too_long && std::chrono::steady_clock::now() - R.armed_at < std::chrono::milliseconds(R.idle_ms))
        return;
    __atomic_store_n(R.bell(), R.seq, __ATOMIC_RELEASE);
    R.state.store(RS::kStale, std::memory_order_release);
    (by_watchdog ? R.dropped_by_watchdog : R.dropped_by_entry).fetch_add(1, std::memory_order_relaxed);
}
// before a device-wide wait of our own (hipFree, hipDeviceSynchronize): nothing of ours should be parked on that
// device.  `self`: the calling handle (its own queued work has been taken back by enter() already).
void resident_ring_device(int device, const pipe_hip_processor *self)
{
    if (device < 0 || device >= kMaxDevices)
        return;
    std::lock_guard<std::mutex> lk(g_door_mu);
    pipe_hip_processor *o = g_door[device].owner;
    if (o && o != self)
        resident_ring_foreign(o, false, false);
    Door &D = g_door[device];
    if (!D.sharers.empty() && t_holds_qmu != device) {  // (a sharing handle's own entry has emptied the queue already)
        std::unique_lock<std::mutex> ql(*D.qmu, std::try_to_lock);
        if (ql.owns_lock())
            shared_ring_all(D, false);
    }
}
void resident_ring_all()  // process exit: no queue may be left waiting for a host that has gone
{
    std::lock_guard<std::mutex> lk(g_door_mu);
    for (int d = 0; d < kMaxDevices; ++d)
        if (pipe_hip_processor *o = g_door[d].owner)
            if (o->resident.mail.p && o->resident.state.load() == pipe_hip_processor::Resident::kArmed)
                __atomic_store_n(o->resident.bell(), o->resident.seq, __ATOMIC_RELEASE);
    for (int d = 0; d < kMaxDevices; ++d)
        for (pipe_hip_processor *q : g_door[d].sharers)
            if (q->resident.mail.p)
                __atomic_store_n(q->resident.bell(), q->resident.seq, __ATOMIC_RELEASE);
}
void resident_watchdog()
{
    for (;;) {
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        std::lock_guard<std::mutex> lk(g_door_mu);
        stall_report_if_stuck();
        for (int d = 0; d < kMaxDevices; ++d) {
            if (pipe_hip_processor *o = g_door[d].owner)
                resident_ring_foreign(o, true, true);
            Door &D = g_door[d];
            if (!D.sharers.empty()) {  // (sharers change under g_door_mu; the queue itself is looked at under its lock)
                std::unique_lock<std::mutex> ql(*D.qmu, std::try_to_lock);
                if (ql.owns_lock() && !D.order.empty()) {
                    int idle_ms = 250;
                    for (pipe_hip_processor *q : D.sharers)
                        idle_ms = q->resident.idle_ms < idle_ms ? q->resident.idle_ms : idle_ms;
                    if (std::chrono::steady_clock::now() - D.last_call >= std::chrono::milliseconds(idle_ms))
                        shared_ring_all(D, true);
                }
            }
        }
    }
}
int door_prepare(int d);
// the device's doorbell for `p` (its device selected): PIPE_HIP_EBUSY when another handle holds it
int resident_acquire(pipe_hip_processor *p)
{
    const int d = p->cfg.device;
    if (d < 0 || d >= kMaxDevices)
        return PIPE_HIP_EINVAL;
    std::lock_guard<std::mutex> lk(g_door_mu);
    if (g_door[d].owner == p)
        return PIPE_HIP_OK;
    if (g_door[d].owner || !g_door[d].sharers.empty())
        return PIPE_HIP_EBUSY;
    PH_TRY(door_prepare(d));
    g_door[d].owner = p;
    return PIPE_HIP_OK;
}
// the device's doorbell stream, the exit hook and the watchdog (g_door_mu held)
int door_prepare(int d)
{
    static bool hooked = false;
    if (!hooked) {
        hooked = true;
        std::atexit(resident_ring_all);
        std::thread(resident_watchdog).detach();
    }
    if (!g_door[d].stream) {
        // a stream with every CU enabled in its mask: the runtime gives it a hardware queue of its own instead of one of
        // the four that all other streams share (nothing is parked on this device now: no owner)
        hipDeviceProp_t prop;
        PH_HIP(hipGetDeviceProperties(&prop, d));
        const int cus = prop.multiProcessorCount;
        std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0xFFFFFFFFu);
        if (cus % 32)
            mask.back() = (1u << (cus % 32)) - 1u;
        hipStream_t s = nullptr;
        if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            return PIPE_HIP_EINVAL;  // (no such stream here: the handle stays on the plain path)
        }
        g_door[d].stream = s;
    }
    return PIPE_HIP_OK;
}
void resident_release(pipe_hip_processor *p)
{
    const int d = p->cfg.device;
    if (d < 0 || d >= kMaxDevices)
        return;
    std::lock_guard<std::mutex> lk(g_door_mu);
    if (g_door[d].owner == p)
        g_door[d].owner = nullptr;
}
hipStream_t resident_stream(int device) { return g_door[device].stream; }

// switch the doorbell path of `p` on / off (its entry has run: nothing of its own is queued)
int resident_enable_shared(pipe_hip_processor *p, double value);
int resident_enable(pipe_hip_processor *p, double value)
{
    pipe_hip_processor::Resident &R = p->resident;
    if (value == 0.0) {
        if (R.enabled && R.shared)
            return resident_enable_shared(p, 0.0);
        if (R.enabled) {
            R.enabled = false;
            (void)hipStreamSynchronize(p->stream);
            if (R.own_stream) {
                p->stream = R.own_stream;
                R.own_stream = nullptr;
            }
            resident_release(p);
        }
        return PIPE_HIP_OK;
    }
    // one Line, a fixed-rate single-input stage that can take a queued launch back, buffers small enough
    // for the zero-copy staging path
    const size_t bytes = dtype_size(p->cfg.dtype) * (size_t)p->cfg.buffer_size * (size_t)p->cfg.channels * (size_t)p->cfg.lines;
    if (p->owned_by_chain || !p->armable() || !p->fixed_rate() || !p->single_input() || bytes > ((size_t)1 << 20) || p->in_flight)
        return PIPE_HIP_EINVAL;
    PH_TRY(p->ensure_staging(0));
    if (!p->stg[0].hd_in || !p->stg[0].hd_out)
        return PIPE_HIP_EINVAL;
    if (!R.mail.p) {
        PH_TRY(R.mail.alloc(128, true));
        std::memset(R.mail.p, 0, 128);
    }
    R.idle_ms = value > 1.0 ? (int)value : 250;  // (a value above 1: the idle limit in milliseconds)
    if (R.enabled)
        return R.shared ? PIPE_HIP_EBUSY : PIPE_HIP_OK;  // (a sharer: leave the shared queue first)
    PH_TRY(resident_acquire(p));  // PIPE_HIP_EBUSY: another handle of this device holds the doorbell
    PH_HIP(hipStreamSynchronize(p->stream));  // (the two streams trade places with nothing in flight on either)
    R.own_stream = p->stream;
    p->stream = resident_stream(p->cfg.device);
    R.failed = false;
    R.enabled = true;
    return PIPE_HIP_OK;
}
// PIPE_HIP_PARAM_RESIDENT_SHARED on / off (the handle's entry has run: nothing of the device is parked)
int resident_enable_shared(pipe_hip_processor *p, double value)
{
    pipe_hip_processor::Resident &R = p->resident;
    const int d = p->cfg.device;
    if (d < 0 || d >= kMaxDevices)
        return PIPE_HIP_EINVAL;
    Door &D = g_door[d];
    if (value == 0.0) {
        if (R.enabled && R.shared) {
            {
                QmuHold hold(D, d);
                PH_TRY(shared_cancel_all_locked(D));
            }
            {
                At at("leaving the shared queue: hipStreamSynchronize(p->stream)", p);
                (void)hipStreamSynchronize(p->stream);
            }
            QmuHold hold(D, d);  // (lock order: the queue's lock, then g_door_mu; whoever holds g_door_mu only TRIES the queue's)
            std::lock_guard<std::mutex> lk(g_door_mu);
            D.sharers.erase(std::remove(D.sharers.begin(), D.sharers.end(), p), D.sharers.end());
            R.enabled = R.shared = false;
        }
        return PIPE_HIP_OK;
    }
    if (R.enabled)
        return R.shared ? PIPE_HIP_OK : PIPE_HIP_EBUSY;  // (it holds the doorbell alone: give that back first)
    const size_t bytes = dtype_size(p->cfg.dtype) * (size_t)p->cfg.buffer_size * (size_t)p->cfg.channels * (size_t)p->cfg.lines;
    if (p->owned_by_chain || !p->armable() || !p->fixed_rate() || !p->single_input() || bytes > ((size_t)1 << 20) || p->in_flight)
        return PIPE_HIP_EINVAL;
    // (a stage whose full buffers would run the plain path -- a float64 biquad without PIPE_HIP_PARAM_RELAXED_F64: the
    // ordered recurrence cannot be taken back -- would empty the whole queue at every call: it stays outside)
    if (!p->armable_for(p->cfg.buffer_size, p->cfg.dtype))
        return PIPE_HIP_EINVAL;
    PH_TRY(p->ensure_staging(0));
    if (!p->stg[0].hd_in || !p->stg[0].hd_out)
        return PIPE_HIP_EINVAL;
    if (!R.mail.p) {
        PH_TRY(R.mail.alloc(128, true));
        std::memset(R.mail.p, 0, 128);
    }
    R.idle_ms = value > 1.0 ? (int)value : 250;
    {
        std::lock_guard<std::mutex> lk(g_door_mu);
        if (D.owner || D.sharers.size() >= kMaxSharers)
            return PIPE_HIP_EBUSY;
        PH_TRY(door_prepare(d));
    }
    {
        // (a handle that joins: whatever the others have parked is taken back first -- the streams trade places with
        // nothing in flight, and this handle's first entry goes to the queue's tail like everybody's)
        QmuHold hold(D, d);
        PH_TRY(shared_cancel_all_locked(D));
    }
    {
        At at("joining the shared queue: hipStreamSynchronize(p->stream)", p);
        PH_HIP(hipStreamSynchronize(p->stream));
    }
    QmuHold hold(D, d);
    std::lock_guard<std::mutex> lk(g_door_mu);
    if (D.owner || D.sharers.size() >= kMaxSharers)
        return PIPE_HIP_EBUSY;
    D.sharers.push_back(p);  // (`stream` stays the handle's own: queue_stream())
    R.failed = false;
    R.shared = true;
    R.enabled = true;
    return PIPE_HIP_OK;
}
}  // namespace

namespace pipehip {
void ring_parked_before_free()
{
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    if (d >= 0 && d < kMaxDevices && g_door[d].owner)  // (a racy peek: the common case is no doorbell owner at all)
        resident_ring_device(d, nullptr);
}
}  // namespace pipehip

int pipe_hip_processor::enter()
{
    PH_TRY(select_device());
    return resident_cancel(this);
}

pipe_hip_processor::~pipe_hip_processor()
{
    if (cfg.buffer_size > 0)
        (void)hipSetDevice(cfg.device);
    if (resident.mail.p) {
        (void)resident_cancel(this);
        if (resident.enabled && resident.shared)
            (void)resident_enable_shared(this, 0.0);
        else if (resident.enabled)
            (void)resident_enable(this, 0.0);  // (the handle's own stream back in `stream`, the doorbell free again)
    }
    // (the frees below wait for every queue of the device: another handle's queued work is rung first, as a mutation
    // of that handle would -- it runs on stale input and is taken back by its owner)
    if (cfg.buffer_size > 0)
        resident_ring_device(cfg.device, this);
    delete overlap;
    if (stream) {
        At at("destructor: hipStreamSynchronize + hipStreamDestroy", this);
        (void)hipStreamSynchronize(stream);
        (void)hipStreamDestroy(stream);
    }
    for (Staging &g : stg)
        if (g.done)
            (void)hipEventDestroy(g.done);
}

int pipe_hip_processor::select_device() const
{
    PH_HIP(hipSetDevice(cfg.device));
    return PIPE_HIP_OK;
}

void pipe_hip_processor::Knobs::read()
{
    if (const char *e = std::getenv("PIPE_HIP_FIR_OLS_MIN_ITEMS"))
        fir_ols_min_items = std::atoll(e);
    if (const char *e = std::getenv("PIPE_HIP_FIR_MFMA_MIN_PASSES"))
        fir_mfma_min_passes = std::atoll(e);
    if (const char *e = std::getenv("PIPE_HIP_OVERLAP_MIN_BYTES"))
        overlap_min_bytes = (size_t)std::atoll(e);
    if (const char *e = std::getenv("PIPE_HIP_ZERO_COPY_MAX"))
        zero_copy_max = (size_t)std::atoll(e);
    if (const char *e = std::getenv("PIPE_HIP_RESAMPLE_ROWS_MIN_BLOCKS")) {
        resample_rows_min_blocks = std::atoll(e);
        resample_rows_stereo = true;
    }
    if (const char *e = std::getenv("PIPE_HIP_BAR_UPLOAD"))
        bar_upload = e[0] != '0' ? 1 : 0;
}

int pipe_hip_processor::init_common(const pipe_hip_config *c)
{
    PH_TRY(validate_config(c));
    cfg = *c;
    knobs.read();
    PH_TRY(select_device());
    PH_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    for (Staging &g : stg)
        PH_HIP(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
    return PIPE_HIP_OK;
}

// Staging for the host-pointer form: one pipe buffer per Line in, the largest
// possible output per Line out.  Allocated on first use so that handles driven
// only through the device-resident batch entry never pay for it.
int pipe_hip_processor::ensure_staging(int slot)
{
    Staging &g = stg[slot];
    if (g.d_in.p)
        return PIPE_HIP_OK;
    const size_t es = dtype_size(cfg.dtype);
    // (+16 bytes per Line: the runs of a ragged pipe_hip_process_lines pass start 16-byte aligned)
    const size_t in_b = es * (size_t)cfg.lines * (size_t)cfg.buffer_size * (size_t)cfg.channels + 16u * (size_t)cfg.lines;
    const size_t out_f = (size_t)max_out_frames(cfg.buffer_size);
    const size_t out_b = es * (size_t)cfg.lines * out_f * (size_t)out_channels() + 16u * (size_t)cfg.lines;
    PH_TRY(g.d_in.alloc(in_b));
    PH_TRY(g.d_out.alloc(out_b));
    PH_TRY(g.h_in.alloc(in_b));
    PH_TRY(g.h_out.alloc(out_b));
    // device-side aliases of the pinned buffers (zero-copy path of small buffers)
    if (hipHostGetDevicePointer(&g.hd_in, g.h_in.p, 0) != hipSuccess ||
        hipHostGetDevicePointer(&g.hd_out, g.h_out.p, 0) != hipSuccess) {
        (void)hipGetLastError();
        g.hd_in = g.hd_out = nullptr;
    }
    return PIPE_HIP_OK;
}

namespace {

int finish_create(int rc, pipe_hip_processor **out)
{
    if (rc != PIPE_HIP_OK && out)
        *out = nullptr;
    return rc;
}

// completion words instead of events for this handle's small calls?  (decided once per handle)
bool express_on(pipe_hip_processor *p)
{
    if (p->express >= 0)
        return p->express != 0;
    p->express = 0;
    static const bool by_event = std::getenv("PIPE_HIP_COMPLETION_EVENT") != nullptr;
    int can = 0;
    if (by_event || hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, p->cfg.device) != hipSuccess || !can) {
        (void)hipGetLastError();
        return false;
    }
    if (p->express_mail.alloc(128, true) != PIPE_HIP_OK)
        return false;
    std::memset(p->express_mail.p, 0, 128);
    p->express = 1;
    return true;
}

// stage `in` and queue H2D -> stage body -> D2H on the handle's stream.  Up to two buffers may be
// in flight: the second one's staging and launch overlap the first one's kernels and transfers.
int submit_impl(pipe_hip_processor *p, const void *in, int32_t in_frames, int32_t out_cap_hint)
{
    if (!p->single_input())
        return PIPE_HIP_EINVAL;
    if (in_frames < 0 || in_frames > p->cfg.buffer_size || (!in && in_frames > 0))
        return PIPE_HIP_EINVAL;
    if (p->in_flight >= 2)
        return PIPE_HIP_ESTATE;
    PH_TRY(p->enter());
    const int slot = p->submit_slot;
    PH_TRY(p->ensure_staging(slot));
    pipe_hip_processor::Staging &g = p->stg[slot];
    const size_t es = dtype_size(p->cfg.dtype);
    const size_t in_b = es * (size_t)p->cfg.lines * (size_t)in_frames * (size_t)p->cfg.channels;
    int64_t out_frames = in_frames;
    // the output Line stride of a rate changer is its capacity; fixed-rate
    // stages pack Lines at in_frames like the input
    const int64_t cap = p->fixed_rate() ? (int64_t)in_frames : p->max_out_frames(p->cfg.buffer_size);
    if (!p->fixed_rate()) {
        const int64_t need = p->out_frames_for(in_frames);
        if (out_cap_hint >= 0 && need > out_cap_hint)
            return PIPE_HIP_ECAP;
    } else if (out_cap_hint >= 0 && in_frames > out_cap_hint) {
        return PIPE_HIP_ECAP;
    }
    // One pipe buffer is tens of KiB: two DMA launches cost more than the transfer.  Small
    // buffers are therefore processed zero-copy -- the kernels read the pinned staging
    // buffer and write the pinned result buffer straight over PCIe (one launch chain, no
    // hipMemcpyAsync); large ones (many Lines per handle) keep the DMA engines.
    const size_t zero_copy_max = p->knobs.zero_copy_max;
    const size_t out_b_cap = es * (size_t)p->cfg.lines * (size_t)cap * (size_t)p->out_channels();
    const bool zero_copy = in_b <= zero_copy_max && out_b_cap <= zero_copy_max && g.hd_in && g.hd_out;
    if (in_b)
        std::memcpy(g.h_in.p, in, in_b);
    bool recorded = false;  // the stage's last launch carries g.done as its stop event
    // Small calls complete by a WORD: hipStreamWriteValue32 behind the last launch, collect spins on it -- launch +
    // word + spin 10.4 us against 13.0 us for launch + event (scripts/micro/queue_independence.hip); nothing waits in
    // any queue, so any number of handles may do it.  Large calls (DMA path: milliseconds) keep the event.
    g.express = zero_copy && express_on(p);
    if (zero_copy) {
        p->completion = g.express ? nullptr : g.done;
        const int rc = p->run_var(g.hd_in, p->cfg.dtype, in_frames, g.hd_out, p->cfg.dtype, cap, &out_frames, p->stream);
        recorded = !g.express && p->completion == nullptr;
        p->completion = nullptr;
        PH_TRY(rc);
        if (g.express) {
            const unsigned t = ++p->express_ticket[slot];
            if (hipStreamWriteValue32(p->stream, static_cast<unsigned *>(p->express_mail.p) + 16 * slot, t, 0) != hipSuccess) {
                (void)hipGetLastError();  // (no stream memory operations after all: events from now on)
                p->express = 0;
                g.express = false;
            } else {
                recorded = true;
            }
        }
    } else {
        if (in_b)
            PH_HIP(hipMemcpyAsync(g.d_in.p, g.h_in.p, in_b, hipMemcpyHostToDevice, p->stream));
        PH_TRY(p->run_var(g.d_in.p, p->cfg.dtype, in_frames, g.d_out.p, p->cfg.dtype, cap, &out_frames, p->stream));
        const size_t out_b = es * (size_t)p->cfg.lines * (size_t)(p->fixed_rate() ? out_frames : cap) *
                             (size_t)p->out_channels();
        if (out_b)
            PH_HIP(hipMemcpyAsync(g.h_out.p, g.d_out.p, out_b, hipMemcpyDeviceToHost, p->stream));
    }
    if (!recorded)
        PH_HIP(hipEventRecord(g.done, p->stream));
    g.out_frames = (int32_t)out_frames;
    g.zero_copy = zero_copy;
    p->submit_slot ^= 1;
    p->in_flight += 1;
    return PIPE_HIP_OK;
}

// the OLDEST buffer in flight
int collect_impl(pipe_hip_processor *p, void *out, int32_t out_cap_frames, int32_t *out_frames)
{
    if (p->in_flight < 1)
        return PIPE_HIP_ESTATE;
    PH_TRY(p->enter());
    const int slot = (p->submit_slot - p->in_flight) & 1;
    pipe_hip_processor::Staging &g = p->stg[slot];
    if (g.express) {
        const unsigned *w = static_cast<const unsigned *>(p->express_mail.p) + 16 * slot;
        if (!wait_word(w, p->express_ticket[slot], std::chrono::milliseconds(30000))) {
            // half a minute: a device busy with somebody else's work, or gone -- the runtime's own wait ends either way
            PH_HIP(hipStreamSynchronize(p->stream));
            if (__atomic_load_n(w, __ATOMIC_ACQUIRE) != p->express_ticket[slot])
                return PIPE_HIP_EHIP;
        }
    } else {
        PH_HIP(hipEventSynchronize(g.done));
    }
    p->in_flight -= 1;
    // a look-back launch that gave up: with this buffer the only one in flight the call is run again here (with a
    // second one queued behind it -- submit / collect -- its work has already run on the failed one's state: reported)
    bool reran = false;
    int late = PIPE_HIP_OK;
    if (p->in_flight == 0) {
        PH_TRY(p->settle(p->stream, &reran));
        if (reran && !g.zero_copy) {
            const size_t es0 = dtype_size(p->cfg.dtype);
            const size_t out_b = es0 * (size_t)p->cfg.lines * (size_t)(p->fixed_rate() ? g.out_frames : p->max_out_frames(p->cfg.buffer_size)) *
                                 (size_t)p->out_channels();
            PH_HIP(hipMemcpy(g.h_out.p, g.d_out.p, out_b, hipMemcpyDeviceToHost));
        }
    } else {
        late = p->poll_error();
    }
    const int32_t n = g.out_frames;
    if (n > out_cap_frames)
        return PIPE_HIP_ECAP;
    const size_t es = dtype_size(p->cfg.dtype);
    const size_t row = es * (size_t)n * (size_t)p->out_channels();
    if (row && !out)
        return PIPE_HIP_EINVAL;
    if (p->fixed_rate() || p->cfg.lines == 1) {
        if (row)
            std::memcpy(out, g.h_out.p, row * (size_t)p->cfg.lines);
    } else {
        // rate changer with several Lines: device rows are `cap` frames apart,
        // the caller's are out_cap_frames apart
        const size_t src_stride = es * (size_t)p->max_out_frames(p->cfg.buffer_size) * (size_t)p->out_channels();
        const size_t dst_stride = es * (size_t)out_cap_frames * (size_t)p->out_channels();
        for (int l = 0; l < p->cfg.lines; ++l)
            std::memcpy((char *)out + dst_stride * l, (const char *)g.h_out.p + src_stride * l, row);
    }
    if (out_frames)
        *out_frames = n;
    return late;
}

struct WindowGuard {  // whatever happens, the handle goes back to "all Lines"
    pipe_hip_processor *p;
    ~WindowGuard() { p->set_window(0, 0); }
};

// Lines [first, first + count) of a fixed-rate handle, every one `frames` frames: in_of(l) /
// out_of(l) are the HOST rows of Line l; the chunk rows sit packed in the handle's staging buffers
// from byte in_off0 / out_off0.  Synchronous: on return the outputs are in the caller's buffers.
int process_overlapped(pipe_hip_processor *p, int first, int count, int32_t frames,
                       const std::function<const void *(int)> &in_of, const std::function<void *(int)> &out_of,
                       size_t in_off0, size_t out_off0)
{
    const size_t es = dtype_size(p->cfg.dtype);
    const size_t row_in = es * (size_t)frames * (size_t)p->cfg.channels;
    const size_t row_out = es * (size_t)frames * (size_t)p->out_channels();
    if (!p->overlap) {
        p->overlap = new pipe_hip_processor::Overlap();
        p->overlap->device = p->cfg.device;
    }
    pipe_hip_processor::Overlap &ov = *p->overlap;
    if (!ov.s_in) {
        PH_HIP(hipStreamCreateWithFlags(&ov.s_in, hipStreamNonBlocking));
        PH_HIP(hipStreamCreateWithFlags(&ov.s_out, hipStreamNonBlocking));
    }
    // chunks of about 4 MiB of input, at least four of them, whole Lines
    const size_t total_in = row_in * (size_t)count;
    size_t nchunks = total_in / ((size_t)4 << 20);
    nchunks = nchunks < 4 ? 4 : (nchunks > 32 ? 32 : nchunks);
    if (nchunks > (size_t)count)
        nchunks = (size_t)count;
    const int per = (int)((count + nchunks - 1) / nchunks);
    nchunks = (size_t)((count + per - 1) / per);
    PH_TRY(ov.events(3 * nchunks));
    pipe_hip_processor::Staging &g = p->stg[0];
    pipe_hip_processor::Overlap::Call c{};
    c.p = p;
    c.first = first;
    c.count = count;
    c.per = per;
    c.frames = frames;
    c.row_in = row_in;
    c.row_out = row_out;
    c.nchunks = nchunks;
    c.h_in = static_cast<char *>(g.h_in.p) + in_off0;
    c.d_in = static_cast<char *>(g.d_in.p) + in_off0;
    c.h_out = static_cast<char *>(g.h_out.p) + out_off0;
    c.d_out = static_cast<char *>(g.d_out.p) + out_off0;
    c.in_of = &in_of;
    c.out_of = &out_of;
    // Large BAR: device memory is mapped into the host's address space, so the copy INTO staging can
    // be the upload itself (one pass over the caller's rows, write-combined stores over PCIe) instead
    // of a copy into pinned memory plus a DMA.  PIPE_HIP_BAR_UPLOAD=0 keeps the DMA path.
    {
        const BarInfo bi = bar_info(p->cfg.device);
        // default: where the device has a large BAR AND says where its HDP flush register is; without the
        // register only on request (PIPE_HIP_BAR_UPLOAD=1: a platform whose host writes are coherent without it)
        c.bar = bi.large_bar && (p->knobs.bar_upload >= 0 ? p->knobs.bar_upload != 0 : bi.hdp_flush != nullptr);
        c.hdp_flush = bi.hdp_flush;
        if (c.bar) {
            // ... and the staging buffer really is writable in this process's address space (asked once per
            // staging allocation)
            if (ov.bar_checked != static_cast<const void *>(g.d_in.p) || ov.bar_checked_bytes != g.d_in.bytes) {
                ov.bar_checked = g.d_in.p;
                ov.bar_checked_bytes = g.d_in.bytes;
                ov.bar_ok = host_writable(g.d_in.p, g.d_in.bytes);
            }
            c.bar = ov.bar_ok;
        }
    }
    c.trace = PH_ENV_AB("PIPE_HIP_OVERLAP_TRACE") != nullptr;  // debug: where a call's time goes
    c.t0 = std::chrono::steady_clock::now();
    c.tr.assign(c.trace ? nchunks * 5 : 0, 0.0);
    WindowGuard guard{p};
    const int rc = ov.run(c);
    // (on an error too: nothing of this call may still be running when its frame goes away)
    const hipError_t e1 = hipStreamSynchronize(p->stream), e2 = hipStreamSynchronize(ov.s_out);
    if (rc != PIPE_HIP_OK)
        return rc;
    PH_HIP(e1);
    PH_HIP(e2);
    if (c.trace) {
        std::fprintf(stderr, "[overlap] %zu chunks of %d Lines, %d copy threads, %.0f us; per chunk (us since the call began): "
                             "copied in, queued | picked up for copy-out, downloaded, copied out\n",
                     nchunks, per, pipe_hip_processor::Overlap::copy_threads(), c.us());
        for (size_t k = 0; k < nchunks; ++k)
            std::fprintf(stderr, "[overlap]   %2zu: %7.0f %7.0f | %7.0f %7.0f %7.0f\n", k, c.tr[k * 5], c.tr[k * 5 + 1],
                         c.tr[k * 5 + 2], c.tr[k * 5 + 3], c.tr[k * 5 + 4]);
    }
    return PIPE_HIP_OK;
}

}  // namespace

// ---- extern "C" ----------------------------------------------------------------
extern "C" {

int pipe_hip_abi_version(void) { return PIPE_HIP_ABI_VERSION; }

int pipe_hip_build_flags(void)
{
#ifdef PIPE_HIP_AB
    return 1;
#else
    return 0;
#endif
}

const char *pipe_hip_strerror(int status)
{
    switch (status) {
    case PIPE_HIP_OK: return "ok";
    case PIPE_HIP_EINVAL: return "invalid argument";
    case PIPE_HIP_ENODEV: return "no such HIP device";
    case PIPE_HIP_EHIP: return "HIP runtime error";
    case PIPE_HIP_ENOMEM: return "out of memory";
    case PIPE_HIP_ECAP: return "output exceeds buffer capacity";
    case PIPE_HIP_ESTATE: return "call out of order";
    case PIPE_HIP_EBUSY: return "the device's doorbell is held by another handle";
    default: return "unknown status";
    }
}

int pipe_hip_last_hip_error(void) { return g_last_hip_error; }

int pipe_hip_device_count(int32_t *count)
{
    if (!count)
        return PIPE_HIP_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        *count = 0;
        return PIPE_HIP_ENODEV;
    }
    *count = n;
    return PIPE_HIP_OK;
}

int pipe_hip_gain_create(const pipe_hip_config *cfg, double gain, pipe_hip_processor **out)
{
    if (!out)
        return PIPE_HIP_EINVAL;
    return finish_create(make_gain(cfg, gain, out), out);
}

int pipe_hip_fir_create(const pipe_hip_config *cfg, const double *taps, int32_t ntaps,
                        pipe_hip_processor **out)
{
    if (!out)
        return PIPE_HIP_EINVAL;
    return finish_create(make_fir(cfg, taps, ntaps, out), out);
}

int pipe_hip_biquad_create(const pipe_hip_config *cfg, const double *coeffs, int32_t nsections,
                           pipe_hip_processor **out)
{
    if (!out)
        return PIPE_HIP_EINVAL;
    return finish_create(make_biquad(cfg, coeffs, nsections, out), out);
}

int pipe_hip_resampler_create(const pipe_hip_config *cfg, const double *proto, int32_t taps_per_phase,
                              int32_t up, int32_t down, pipe_hip_processor **out)
{
    if (!out)
        return PIPE_HIP_EINVAL;
    return finish_create(make_resampler(cfg, proto, taps_per_phase, up, down, out), out);
}

int pipe_hip_mix_create(const pipe_hip_config *cfg, int32_t inputs, pipe_hip_processor **out)
{
    if (!out)
        return PIPE_HIP_EINVAL;
    return finish_create(make_mix(cfg, inputs, out), out);
}

int pipe_hip_chain_create(pipe_hip_processor *const *stages, int32_t n_stages, pipe_hip_processor **out)
{
    if (!out)
        return PIPE_HIP_EINVAL;
    return finish_create(make_chain(stages, n_stages, out), out);
}

int pipe_hip_output_properties(const pipe_hip_processor *p, int32_t *channels, int32_t *rate_up,
                               int32_t *rate_down)
{
    if (!p)
        return PIPE_HIP_EINVAL;
    int32_t up = 1, down = 1;
    p->rate(&up, &down);
    if (channels)
        *channels = p->out_channels();
    if (rate_up)
        *rate_up = up;
    if (rate_down)
        *rate_down = down;
    return PIPE_HIP_OK;
}

int pipe_hip_start(pipe_hip_processor *p)
{
    if (!p)
        return PIPE_HIP_EINVAL;
    At at("pipe_hip_start", p);
    PH_TRY(p->enter());
    PH_TRY(p->drain());  // a restarted pipe drops whatever was in flight, on whichever stream
    p->in_flight = 0;
    p->resident.failed = false;
    At at2("pipe_hip_start: start(p->stream) + hipStreamSynchronize(p->stream)", p);
    PH_TRY(p->start(p->stream));
    PH_HIP(hipStreamSynchronize(p->stream));
    return PIPE_HIP_OK;
}

int pipe_hip_start_lines(pipe_hip_processor *p, int32_t first, int32_t count)
{
    if (!p || first < 0 || count < 0 || first > p->cfg.lines || count > p->cfg.lines - first)
        return PIPE_HIP_EINVAL;
    if (p->in_flight)
        return PIPE_HIP_ESTATE;
    PH_TRY(p->enter());
    PH_TRY(p->drain());
    PH_TRY(p->start_lines(first, count, p->stream));
    PH_HIP(hipStreamSynchronize(p->stream));
    return PIPE_HIP_OK;
}

int pipe_hip_flush(pipe_hip_processor *p)
{
    if (!p)
        return PIPE_HIP_EINVAL;
    At at("pipe_hip_flush", p);
    PH_TRY(p->enter());
    PH_TRY(p->drain());
    p->in_flight = 0;
    return p->poll_error();
}

int pipe_hip_destroy(pipe_hip_processor *p)
{
    if (!p)
        return PIPE_HIP_OK;
    if (p->owned_by_chain)
        return PIPE_HIP_EINVAL;
    At at("pipe_hip_destroy", p);
    (void)p->enter();   // (its device; its own queued work taken back BEFORE any member is freed: a free waits for every queue)
    (void)p->drain();   // everything the handle has queued anywhere (its own stream, a caller's stream of a batch call)
    resident_ring_device(p->cfg.device, p);  // (another handle's queued work would hold the frees up until its watchdog)
    delete p;
    return PIPE_HIP_OK;
}

int pipe_hip_submit(pipe_hip_processor *p, const void *in, int32_t in_frames)
{
    if (!p)
        return PIPE_HIP_EINVAL;
    return submit_impl(p, in, in_frames, -1);
}

int pipe_hip_collect(pipe_hip_processor *p, void *out, int32_t out_cap_frames, int32_t *out_frames)
{
    if (!p)
        return PIPE_HIP_EINVAL;
    return collect_impl(p, out, out_cap_frames, out_frames);
}

int pipe_hip_process(pipe_hip_processor *p, const void *in, int32_t in_frames, void *out,
                     int32_t out_cap_frames, int32_t *out_frames)
{
    if (!p || out_cap_frames < 0)
        return PIPE_HIP_EINVAL;
    if (p->in_flight)  // collect would hand back an older buffer
        return PIPE_HIP_ESTATE;
    if (p->resident.failed)  // a queued launch failed on the device: the state is unknown until the next StartFunc
        return PIPE_HIP_ESTATE;
    // many Lines in one call: chunks of Lines, transfers and kernels overlapped
    if (p->fixed_rate() && p->single_input() && p->cfg.lines >= 2 && in && out && in_frames > 0 &&
        in_frames <= p->cfg.buffer_size && in_frames <= out_cap_frames) {
        const size_t es = dtype_size(p->cfg.dtype);
        const size_t row_in = es * (size_t)in_frames * (size_t)p->cfg.channels;
        const size_t row_out = es * (size_t)in_frames * (size_t)p->out_channels();
        if (row_in * (size_t)p->cfg.lines >= p->knobs.overlap_min_bytes) {  // (calls this large: chunks of Lines)
            PH_TRY(p->enter());
            PH_TRY(p->ensure_staging());
            const char *ib = static_cast<const char *>(in);
            char *ob = static_cast<char *>(out);
            PH_TRY(process_overlapped(
                p, 0, p->cfg.lines, in_frames, [=](int l) -> const void * { return ib + row_in * (size_t)l; },
                [=](int l) -> void * { return ob + row_out * (size_t)l; }, 0, 0));
            if (out_frames)
                *out_frames = in_frames;
            return p->poll_error();
        }
    }
    // (a stage whose form depends on the call's size -- the biquad: the tile form can be taken back, the ordered
    // recurrence cannot -- says per call whether its launch may be queued ahead; if not, the plain path below)
    if (p->resident.enabled && p->resident.shared && in && out && in_frames > 0 && in_frames <= p->cfg.buffer_size &&
        in_frames <= out_cap_frames && !p->resident.failed && p->armable_for(in_frames, p->cfg.dtype)) {
        // PIPE_HIP_PARAM_RESIDENT_SHARED: the buffer's work waits in the device's ONE doorbell queue, normally at its head
        // (the handles are called in the order they were called last time: run.go:112-132).  Whatever lies ahead of it
        // is rung first -- other handles' work, run on stale input and taken back by them -- then its own doorbell; the
        // successor goes to the queue's tail while this buffer runs.
        pipe_hip_processor::Resident &R = p->resident;
        using RS = pipe_hip_processor::Resident;
        PH_TRY(p->select_device());
        Door &D = g_door[p->cfg.device];
        At at("pipe_hip_process: shared queue", p);
        QmuHold hold(D, p->cfg.device);
        D.last_call = std::chrono::steady_clock::now();
        const int st = R.state.load(std::memory_order_acquire);
        if (st == RS::kStale || (st == RS::kArmed && R.frames != in_frames)) {
            if (st == RS::kArmed) {  // (a short buffer, pipe.go:441-443: the queued work was made for another frame count)
                (void)shared_ring_through(D, p);
                R.dropped_by_entry.fetch_add(1, std::memory_order_relaxed);
            }
            const int rcw = resident_wait(p, R.seq);
            R.state.store(RS::kIdle, std::memory_order_release);
            PH_TRY(rcw);
            p->rollback_launch();
        }
        bool plain = false;
        if (R.state.load() == RS::kIdle && !D.order.empty()) {
            // Nothing of this handle is queued (its first call, or its work was taken back) and OTHER handles' work is:
            // queued now, this buffer's work would sit BEHIND theirs, and ringing them to get at it would cost each of
            // them a launch -- every handle's first round would knock out its predecessor's successor, for ever.  This
            // one buffer runs on the handle's own stream instead (nothing parked there), and its successor takes its
            // place at the queue's tail: after one round of such calls the queue is in the callers' order.
            pipe_hip_processor::Staging &g = p->stg[0];
            const size_t es = dtype_size(p->cfg.dtype);
            std::memcpy(g.h_in.p, in, es * (size_t)in_frames * (size_t)p->cfg.channels * (size_t)p->cfg.lines);
            int64_t produced = in_frames;
            At at3("pipe_hip_process: shared queue: this buffer on the handle's own stream (run_var + sync)", p);
            PH_TRY(p->run_var(g.hd_in, p->cfg.dtype, in_frames, g.hd_out, p->cfg.dtype, in_frames, &produced, p->stream));
            PH_HIP(hipStreamSynchronize(p->stream));
            bool reran = false;
            PH_TRY(p->settle(p->stream, &reran));
            std::memcpy(out, g.h_out.p, es * (size_t)produced * (size_t)p->out_channels() * (size_t)p->cfg.lines);
            if (out_frames)
                *out_frames = (int32_t)produced;
            const int rc_next = resident_arm(p, in_frames);
            if (rc_next == PIPE_HIP_OK)
                D.order.push_back(p);
            return rc_next == kResidentUnsupported ? PIPE_HIP_OK : rc_next;
        }
        if (R.state.load() == RS::kIdle) {  // the queue is empty: this buffer's work at its head
            const int rc0 = resident_arm(p, in_frames);
            if (rc0 == kResidentUnsupported) {
                plain = true;  // no stream memory operations here: the plain path from now on
            } else {
                PH_TRY(rc0);
                D.order.push_back(p);
            }
        }
        if (!plain) {
            pipe_hip_processor::Staging &g = p->stg[0];
            const size_t es = dtype_size(p->cfg.dtype);
            std::memcpy(g.h_in.p, in, es * (size_t)in_frames * (size_t)p->cfg.channels * (size_t)p->cfg.lines);
            const unsigned k = R.seq;
            const int64_t produced = R.out_frames;
            (void)shared_ring_through(D, p);
            R.state.store(RS::kIdle, std::memory_order_release);  // (rung by its own call: running)
            int rc_next = resident_arm(p, in_frames);
            if (rc_next == PIPE_HIP_OK)
                D.order.push_back(p);
            else if (rc_next == kResidentUnsupported)
                rc_next = PIPE_HIP_EHIP;
            PH_TRY(resident_wait(p, k));
            if (p->take_failure_flag()) {  // (as on the exclusive path: launch k failed and its successor is queued on its state)
                (void)shared_cancel_all_locked(D);
                (void)p->take_failure_flag();
                R.failed = true;
                return PIPE_HIP_EHIP;
            }
            std::memcpy(out, g.h_out.p, es * (size_t)produced * (size_t)p->out_channels() * (size_t)p->cfg.lines);
            if (out_frames)
                *out_frames = (int32_t)produced;
            return rc_next;
        }
        hold.lk.unlock();
        t_holds_qmu = hold.prev;
        (void)resident_enable_shared(p, 0.0);
        goto plain_path;
    }
    if (p->resident.enabled && !p->resident.shared && in && out && in_frames > 0 && in_frames <= p->cfg.buffer_size &&
        in_frames <= out_cap_frames && !p->resident.failed && p->armable_for(in_frames, p->cfg.dtype)) {
        // The buffer's work is already on the device, behind the doorbell (queued while the last buffer ran):
        // copy in, ring, queue the NEXT buffer's work while this one runs, spin on the completion word.
        pipe_hip_processor::Resident &R = p->resident;
        using RS = pipe_hip_processor::Resident;
        PH_TRY(p->select_device());
        std::unique_lock<std::mutex> lk(R.mu);  // (the watchdog keeps its hands off until the result is out)
        if (R.state.load() == RS::kStale || (R.state.load() == RS::kArmed && R.frames != in_frames))
            PH_TRY(resident_cancel_locked(p));  // (rung by somebody else meanwhile / a short buffer: pipe.go:441-443)
        if (R.state.load() == RS::kIdle) {  // the first call, or the one after a cancellation
            const int rc0 = resident_arm(p, in_frames);
            if (rc0 == kResidentUnsupported) {  // no stream memory operations here: the plain path from now on
                lk.unlock();
                (void)resident_enable(p, 0.0);
                goto plain_path;
            }
            PH_TRY(rc0);
        }
        pipe_hip_processor::Staging &g = p->stg[0];
        const size_t es = dtype_size(p->cfg.dtype);
        std::memcpy(g.h_in.p, in, es * (size_t)in_frames * (size_t)p->cfg.channels * (size_t)p->cfg.lines);
        const unsigned k = R.seq;
        const int64_t produced = R.out_frames;
        __atomic_store_n(R.bell(), k, __ATOMIC_RELEASE);
        R.state.store(RS::kIdle, std::memory_order_release);  // (rung by its own call: running, nothing parked)
        if (!R.frees.empty()) {
            // what the queued runs replaced: freed here, with nothing of ours parked (the free waits for launch k -- this
            // one call does not queue its successor under it), so that a handle that stays on this path does not pile
            // them up until its next ordinary entry
            for (const DeferredFree &f : R.frees)
                (void)(f.pinned ? hipHostFree(f.p) : hipFree(f.p));
            R.frees.clear();
        }
        int rc_next = resident_arm(p, in_frames);
        if (rc_next == kResidentUnsupported)  // (it worked a call ago: a runtime error of this call's successor)
            rc_next = PIPE_HIP_EHIP;
        PH_TRY(resident_wait(p, k));
        if (p->take_failure_flag()) {
            // Launch k failed on the device (a look-back that gave up) and launch k + 1 is queued on its state: neither
            // can be taken back any more.  Drop what is queued and say so -- a ProcessFunc error ends the run
            // (pipe.go:438-440); the next StartFunc begins from silence.
            (void)resident_cancel_locked(p);
            (void)p->take_failure_flag();
            R.failed = true;
            return PIPE_HIP_EHIP;
        }
        std::memcpy(out, g.h_out.p, es * (size_t)produced * (size_t)p->out_channels() * (size_t)p->cfg.lines);
        if (out_frames)
            *out_frames = (int32_t)produced;
        return rc_next;
    }
plain_path:
    At at_plain("pipe_hip_process: plain path (submit + collect)", p);
    PH_TRY(submit_impl(p, in, in_frames, out_cap_frames));
    return collect_impl(p, out, out_cap_frames, out_frames);
}

// ---- one pass of many Lines (run.go:112-132 batched) ---------------------------------
// A pass is cut into RUNS of consecutive Lines that bring the same frame count; each run is one
// launch over exactly its Lines (set_window), so every Line's state advances by its own frames --
// also when a Source returns a short read in the middle of its stream and keeps going
// (pipe.go:404-406).  The normal pass (every live Line brings the same count) is one run.
namespace {

struct LineRun {
    int first, count;
    int32_t frames;
    size_t in_off, out_off;  // byte offsets of the run's packed rows in the staging buffers
};

int plan_line_runs(const pipe_hip_processor *p, const void *const *ins, const int32_t *in_frames,
                   void *const *outs, std::vector<LineRun> *runs)
{
    const int L = p->cfg.lines;
    const size_t es = dtype_size(p->cfg.dtype);
    runs->clear();
    int32_t common = -1;
    bool uniform = true;
    for (int l = 0; l < L; ++l) {
        if (in_frames[l] < 0 || in_frames[l] > p->cfg.buffer_size || (ins[l] && in_frames[l] > 0 && !outs[l]))
            return PIPE_HIP_EINVAL;
        if (!ins[l])
            continue;  // a Line that has ended: it rides along zero-padded, its state is dead
        if (common < 0)
            common = in_frames[l];
        else if (in_frames[l] != common)
            uniform = false;
    }
    if (common < 0)
        return PIPE_HIP_OK;  // nothing live
    if (uniform) {
        if (common > 0)
            runs->push_back(LineRun{0, L, common, 0, 0});
        return PIPE_HIP_OK;
    }
    size_t in_off = 0, out_off = 0;
    for (int l = 0; l < L;) {
        if (!ins[l] || in_frames[l] == 0) {  // ended, or an empty read: not advanced
            ++l;
            continue;
        }
        const int32_t f = in_frames[l];
        int end = l + 1, last_live = l;
        while (end < L && (!ins[end] || in_frames[end] == f)) {
            if (ins[end])
                last_live = end;
            ++end;
        }
        const int n = last_live - l + 1;
        runs->push_back(LineRun{l, n, f, in_off, out_off});
        in_off += es * (size_t)f * (size_t)p->cfg.channels * (size_t)n;
        out_off += es * (size_t)f * (size_t)p->out_channels() * (size_t)n;
        // 16-byte alignment of the next run's rows (vector loads in the kernels)
        in_off = (in_off + 15) & ~(size_t)15;
        out_off = (out_off + 15) & ~(size_t)15;
        l = last_live + 1;
    }
    return PIPE_HIP_OK;
}

}  // namespace

int pipe_hip_process_lines(pipe_hip_processor *p, const void *const *ins, const int32_t *in_frames,
                           void *const *outs, int32_t *out_frames)
{
    if (!p || !ins || !in_frames || !outs || !p->fixed_rate() || !p->single_input())
        return PIPE_HIP_EINVAL;
    if (p->in_flight)
        return PIPE_HIP_ESTATE;
    PH_TRY(p->enter());
    PH_TRY(p->ensure_staging());
    const int L = p->cfg.lines;
    std::vector<LineRun> runs;
    PH_TRY(plan_line_runs(p, ins, in_frames, outs, &runs));
    if (out_frames)
        for (int l = 0; l < L; ++l)
            out_frames[l] = ins[l] ? in_frames[l] : 0;
    if (runs.empty())
        return PIPE_HIP_OK;
    const size_t es = dtype_size(p->cfg.dtype);
    // the usual pass -- one run, many Lines, tens of MB: chunks of Lines, transfers and kernels overlapped
    if (runs.size() == 1 && runs[0].count >= 2 &&
        es * (size_t)runs[0].frames * (size_t)p->cfg.channels * (size_t)runs[0].count >= p->knobs.overlap_min_bytes) {
        const LineRun &r = runs[0];
        PH_TRY(process_overlapped(
            p, r.first, r.count, r.frames,
            [=](int l) -> const void * { return ins[l] && in_frames[l] > 0 ? ins[l] : nullptr; },
            [=](int l) -> void * { return ins[l] && in_frames[l] > 0 ? outs[l] : nullptr; }, r.in_off, r.out_off));
        return p->poll_error();
    }
    WindowGuard guard{p};
    // gather: Line l of a run occupies [l - first][frames][channels] of the run's staging rows
    for (const LineRun &r : runs) {
        const size_t row_in = es * (size_t)r.frames * (size_t)p->cfg.channels;
        for (int i = 0; i < r.count; ++i) {
            const int l = r.first + i;
            char *dst = static_cast<char *>(p->stg[0].h_in.p) + r.in_off + row_in * (size_t)i;
            const size_t have = ins[l] ? es * (size_t)in_frames[l] * (size_t)p->cfg.channels : 0;
            if (have)
                std::memcpy(dst, ins[l], have);
            if (have < row_in)
                std::memset(dst + have, 0, row_in - have);
        }
    }
    for (const LineRun &r : runs) {
        const size_t row_in = es * (size_t)r.frames * (size_t)p->cfg.channels;
        const size_t row_out = es * (size_t)r.frames * (size_t)p->out_channels();
        int64_t produced = r.frames;
        p->set_window(r.first, r.count == L ? 0 : r.count);
        PH_HIP(hipMemcpyAsync(static_cast<char *>(p->stg[0].d_in.p) + r.in_off, static_cast<char *>(p->stg[0].h_in.p) + r.in_off,
                              row_in * (size_t)r.count, hipMemcpyHostToDevice, p->stream));
        PH_TRY(p->run_var(static_cast<char *>(p->stg[0].d_in.p) + r.in_off, p->cfg.dtype, r.frames,
                          static_cast<char *>(p->stg[0].d_out.p) + r.out_off, p->cfg.dtype, r.frames, &produced, p->stream));
        PH_HIP(hipMemcpyAsync(static_cast<char *>(p->stg[0].h_out.p) + r.out_off, static_cast<char *>(p->stg[0].d_out.p) + r.out_off,
                              row_out * (size_t)r.count, hipMemcpyDeviceToHost, p->stream));
    }
    PH_HIP(hipStreamSynchronize(p->stream));
    // a look-back launch that gave up: the usual pass (one run: the handle's window is still that run's) is run
    // again and its rows fetched again; a ragged pass of several launches reports it
    int late = PIPE_HIP_OK;
    if (runs.size() == 1) {
        bool reran = false;
        PH_TRY(p->settle(p->stream, &reran));
        if (reran) {
            const LineRun &r = runs[0];
            const size_t row_out = es * (size_t)r.frames * (size_t)p->out_channels();
            PH_HIP(hipMemcpy(static_cast<char *>(p->stg[0].h_out.p) + r.out_off, static_cast<char *>(p->stg[0].d_out.p) + r.out_off,
                             row_out * (size_t)r.count, hipMemcpyDeviceToHost));
        }
    } else {
        late = p->poll_error();
    }
    for (const LineRun &r : runs) {
        const size_t row_out = es * (size_t)r.frames * (size_t)p->out_channels();
        for (int i = 0; i < r.count; ++i) {
            const int l = r.first + i;
            if (!ins[l] || in_frames[l] == 0)
                continue;
            std::memcpy(outs[l], static_cast<const char *>(p->stg[0].h_out.p) + r.out_off + row_out * (size_t)i,
                        es * (size_t)in_frames[l] * (size_t)p->out_channels());
        }
    }
    return late;
}

int pipe_hip_process_lines_pinned(pipe_hip_processor *p, const void *const *ins, const int32_t *in_frames,
                                  void *const *outs, int32_t *out_frames)
{
    if (!p || !ins || !in_frames || !outs || !p->fixed_rate() || !p->single_input())
        return PIPE_HIP_EINVAL;
    if (p->in_flight)
        return PIPE_HIP_ESTATE;
    PH_TRY(p->enter());
    PH_TRY(p->ensure_staging());
    const int L = p->cfg.lines;
    const size_t es = dtype_size(p->cfg.dtype);
    const size_t fb_in = es * (size_t)p->cfg.channels, fb_out = es * (size_t)p->out_channels();
    if (fb_in % 8 != 0 || fb_out % 8 != 0)  // the row kernels move 8-byte words
        return pipe_hip_process_lines(p, ins, in_frames, outs, out_frames);
    std::vector<LineRun> runs;
    PH_TRY(plan_line_runs(p, ins, in_frames, outs, &runs));
    if (out_frames)
        for (int l = 0; l < L; ++l)
            out_frames[l] = ins[l] ? in_frames[l] : 0;
    if (runs.empty())
        return PIPE_HIP_OK;
    // tables the kernels read: [L] in pointers, [L] out pointers, [L] in words, [L] out words
    const size_t tab_bytes = (size_t)L * (2 * sizeof(void *) + 2 * sizeof(int));
    if (p->line_tab.bytes < tab_bytes)
        PH_TRY(p->line_tab.alloc(tab_bytes));
    const void **tin = static_cast<const void **>(p->line_tab.p);
    void **tout = const_cast<void **>(tin + L);
    int *win = reinterpret_cast<int *>(tout + L);
    int *wout = win + L;
    for (int l = 0; l < L; ++l) {
        const bool live = ins[l] && in_frames[l] > 0;
        tin[l] = live ? ins[l] : nullptr;
        tout[l] = live ? outs[l] : nullptr;
        win[l] = live ? (int)((size_t)in_frames[l] * fb_in / 8) : 0;
        wout[l] = live ? (int)((size_t)in_frames[l] * fb_out / 8) : 0;
    }
    WindowGuard guard{p};
    // The usual pass -- one run, many Lines, tens of MB: chunks of Lines on three streams, the row kernels of chunk
    // k + 1 reading the caller's pinned buffers over PCIe while chunk k's stage kernels run and chunk k - 1's rows
    // go back (PCIe is full duplex: the call costs the longer direction, not the sum).  Everything is queued up
    // front; no host thread copies anything.
    if (runs.size() == 1 && runs[0].count >= 2 &&
        fb_in * (size_t)runs[0].frames * (size_t)runs[0].count >= p->knobs.overlap_min_bytes) {
        const LineRun &r = runs[0];
        if (!p->overlap) {
            p->overlap = new pipe_hip_processor::Overlap();
            p->overlap->device = p->cfg.device;
        }
        pipe_hip_processor::Overlap &ov = *p->overlap;
        if (!ov.s_in) {
            PH_HIP(hipStreamCreateWithFlags(&ov.s_in, hipStreamNonBlocking));
            PH_HIP(hipStreamCreateWithFlags(&ov.s_out, hipStreamNonBlocking));
        }
        const size_t row_in = fb_in * (size_t)r.frames, row_out = fb_out * (size_t)r.frames;
        const char *cmb = PH_ENV_AB("PIPE_HIP_PINNED_CHUNK_MB");  // A/B: chunk size
        size_t nchunks = row_in * (size_t)r.count / ((size_t)(cmb ? std::atoi(cmb) : 8) << 20);
        nchunks = nchunks < 4 ? 4 : (nchunks > 64 ? 64 : nchunks);
        if (nchunks > (size_t)r.count)
            nchunks = (size_t)r.count;
        const int per = (int)((r.count + nchunks - 1) / nchunks);
        nchunks = (size_t)((r.count + per - 1) / per);
        PH_TRY(ov.events(3 * nchunks));
        // Rows that lie back to back in ONE pinned block (a pool carved from one pipe_hip_host_alloc) move by the
        // DMA engines, a chunk per copy: both directions at once reach 48 GB/s each way on this link
        // (scripts/micro/pcie_rates.hip), where a reading and a writing kernel side by side share 56 -- they
        // take turns.  Scattered rows keep the row kernels.
        bool dense = true;
        for (int i = 0; i < r.count && dense; ++i) {
            const int l = r.first + i;
            dense = tin[l] && tout[l] && win[l] == (int)(row_in / 8) &&
                    static_cast<const char *>(tin[l]) == static_cast<const char *>(tin[r.first]) + row_in * (size_t)i &&
                    static_cast<char *>(tout[l]) == static_cast<char *>(tout[r.first]) + row_out * (size_t)i;
        }
        // (s_in / s_out start behind whatever the handle's stream has queued: state and parameter uploads)
        PH_HIP(hipEventRecord(ov.ev[0], p->stream));
        PH_HIP(hipStreamWaitEvent(ov.s_in, ov.ev[0], 0));
        int rc = PIPE_HIP_OK;
        for (size_t k = 0; k < nchunks && rc == PIPE_HIP_OK; ++k) {
            const int l0 = (int)k * per, n = l0 + per <= r.count ? per : r.count - l0;
            char *din = static_cast<char *>(p->stg[0].d_in.p) + r.in_off + row_in * (size_t)l0;
            char *dout = static_cast<char *>(p->stg[0].d_out.p) + r.out_off + row_out * (size_t)l0;
            hipEvent_t up = ov.ev[3 * k + 1], done = ov.ev[3 * k + 2];
            if (dense)
                PH_HIP(hipMemcpyAsync(din, static_cast<const char *>(tin[r.first]) + row_in * (size_t)l0, row_in * (size_t)n,
                                      hipMemcpyHostToDevice, ov.s_in));
            else
                rc = launch_gather_rows(tin + r.first + l0, win + r.first + l0, din, (int)(row_in / 8), n, ov.s_in);
            if (rc != PIPE_HIP_OK)
                break;
            PH_HIP(hipEventRecord(up, ov.s_in));
            PH_HIP(hipStreamWaitEvent(p->stream, up, 0));
            p->set_window(r.first + l0, (r.first + l0 == 0 && n == L) ? 0 : n);
            int64_t produced = r.frames;
            rc = p->run_var(din, p->cfg.dtype, r.frames, dout, p->cfg.dtype, r.frames, &produced, p->stream);
            if (rc != PIPE_HIP_OK)
                break;
            PH_HIP(hipEventRecord(done, p->stream));
            PH_HIP(hipStreamWaitEvent(ov.s_out, done, 0));
            if (dense)
                PH_HIP(hipMemcpyAsync(static_cast<char *>(tout[r.first]) + row_out * (size_t)l0, dout, row_out * (size_t)n,
                                      hipMemcpyDeviceToHost, ov.s_out));
            else
                rc = launch_scatter_rows(tout + r.first + l0, wout + r.first + l0, dout, (int)(row_out / 8), n, ov.s_out);
        }
        // (on an error too: nothing of this call may still be running when its tables go away)
        const hipError_t e0 = hipStreamSynchronize(ov.s_in), e1 = hipStreamSynchronize(p->stream), e2 = hipStreamSynchronize(ov.s_out);
        PH_TRY(rc);
        PH_HIP(e0);
        PH_HIP(e1);
        PH_HIP(e2);
        return p->poll_error();
    }
    for (const LineRun &r : runs) {
        int64_t produced = r.frames;
        char *din = static_cast<char *>(p->stg[0].d_in.p) + r.in_off;
        char *dout = static_cast<char *>(p->stg[0].d_out.p) + r.out_off;
        p->set_window(r.first, r.count == L ? 0 : r.count);
        PH_TRY(launch_gather_rows(tin + r.first, win + r.first, din, (int)((size_t)r.frames * fb_in / 8), r.count,
                                  p->stream));
        PH_TRY(p->run_var(din, p->cfg.dtype, r.frames, dout, p->cfg.dtype, r.frames, &produced, p->stream));
        PH_TRY(launch_scatter_rows(tout + r.first, wout + r.first, dout, (int)((size_t)r.frames * fb_out / 8), r.count,
                                   p->stream));
    }
    PH_HIP(hipStreamSynchronize(p->stream));
    if (runs.size() != 1)
        return p->poll_error();
    bool reran = false;
    PH_TRY(p->settle(p->stream, &reran));
    if (reran) {  // (the rows of the launch that was run again go back once more)
        const LineRun &r = runs[0];
        PH_TRY(launch_scatter_rows(tout + r.first, wout + r.first, static_cast<char *>(p->stg[0].d_out.p) + r.out_off,
                                   (int)((size_t)r.frames * fb_out / 8), r.count, p->stream));
        PH_HIP(hipStreamSynchronize(p->stream));
    }
    return PIPE_HIP_OK;
}

int pipe_hip_mix_process(pipe_hip_processor *p, const void *const *ins, int32_t n_inputs,
                         int32_t frames, void *out)
{
    if (!p || !ins || frames < 0 || frames > p->cfg.buffer_size || n_inputs < 2 || n_inputs > 8)
        return PIPE_HIP_EINVAL;
    if (p->single_input() || (frames > 0 && !out))
        return PIPE_HIP_EINVAL;
    PH_TRY(p->enter());
    const size_t es = dtype_size(p->cfg.dtype);
    const size_t one = es * (size_t)p->cfg.lines * (size_t)frames * (size_t)p->cfg.channels;
    const size_t cap = es * (size_t)p->cfg.lines * (size_t)p->cfg.buffer_size * (size_t)p->cfg.channels;
    // staging: n input slots + 1 output slot, allocated once
    if (!p->stg[0].d_in.p) {
        PH_TRY(p->stg[0].d_in.alloc(cap * 8));
        PH_TRY(p->stg[0].d_out.alloc(cap));
        PH_TRY(p->stg[0].h_in.alloc(cap * 8));
        PH_TRY(p->stg[0].h_out.alloc(cap));
    }
    const void *d_ins[8] = {};
    for (int i = 0; i < n_inputs; ++i) {
        if (!ins[i] && one)
            return PIPE_HIP_EINVAL;
        char *h = (char *)p->stg[0].h_in.p + cap * i;
        char *d = (char *)p->stg[0].d_in.p + cap * i;
        if (one) {
            std::memcpy(h, ins[i], one);
            PH_HIP(hipMemcpyAsync(d, h, one, hipMemcpyHostToDevice, p->stream));
        }
        d_ins[i] = d;
    }
    PH_TRY(mix_run(p, d_ins, n_inputs, p->stg[0].d_out.p, frames, p->stream));
    if (one)
        PH_HIP(hipMemcpyAsync(p->stg[0].h_out.p, p->stg[0].d_out.p, one, hipMemcpyDeviceToHost, p->stream));
    PH_HIP(hipStreamSynchronize(p->stream));
    if (one)
        std::memcpy(out, p->stg[0].h_out.p, one);
    return PIPE_HIP_OK;
}

int pipe_hip_set_param(pipe_hip_processor *p, int32_t param, const double *values, int32_t count)
{
    if (!p || !values || count < 1)
        return PIPE_HIP_EINVAL;
    PH_TRY(p->enter());
    if (param == PIPE_HIP_PARAM_RESIDENT) {
        if (count != 1)
            return PIPE_HIP_EINVAL;
        return resident_enable(p, values[0]);  // (enter() has taken back what was queued)
    }
    if (param == PIPE_HIP_PARAM_RESIDENT_SHARED) {
        if (count != 1)
            return PIPE_HIP_EINVAL;
        return resident_enable_shared(p, values[0]);
    }
    return p->set_param(param, values, count);
}

int pipe_hip_chain_set_param(pipe_hip_processor *p, int32_t stage, int32_t param, const double *values,
                             int32_t count)
{
    if (!p || !values || count < 1)
        return PIPE_HIP_EINVAL;
    PH_TRY(p->enter());
    return p->set_stage_param(stage, param, values, count);
}

int pipe_hip_process_batch(pipe_hip_processor *p, const void *d_in, void *d_out,
                           int64_t frames_per_line, void *stream)
{
    if (!p || frames_per_line < 0 || (frames_per_line > 0 && (!d_in || !d_out)))
        return PIPE_HIP_EINVAL;
    if (frames_per_line > (int64_t)p->cfg.buffer_size * p->cfg.max_batch)
        return PIPE_HIP_EINVAL;
    if (!p->fixed_rate() || !p->single_input())
        return PIPE_HIP_EINVAL;
    PH_TRY(p->enter());
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : p->stream;
    p->batch_stream = s;
    return p->run(d_in, p->cfg.dtype, d_out, p->cfg.dtype, frames_per_line, s);
}

int pipe_hip_resample_batch(pipe_hip_processor *p, const void *d_in, int64_t in_frames_per_line,
                            void *d_out, int64_t out_cap_frames, int64_t *out_frames, void *stream)
{
    if (!p || in_frames_per_line < 0 || out_cap_frames < 0 || p->fixed_rate())
        return PIPE_HIP_EINVAL;
    if (in_frames_per_line > (int64_t)p->cfg.buffer_size * p->cfg.max_batch)
        return PIPE_HIP_EINVAL;
    PH_TRY(p->enter());
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : p->stream;
    p->batch_stream = s;
    return p->run_var(d_in, p->cfg.dtype, in_frames_per_line, d_out, p->cfg.dtype, out_cap_frames,
                      out_frames, s);
}

int pipe_hip_mix_batch(pipe_hip_processor *p, const void *const *d_ins, int32_t n_inputs, void *d_out,
                       int64_t frames_per_line, void *stream)
{
    if (!p || !d_ins || !d_out || frames_per_line < 0 || p->single_input())
        return PIPE_HIP_EINVAL;
    PH_TRY(p->enter());
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : p->stream;
    p->batch_stream = s;
    return mix_run(p, d_ins, n_inputs, d_out, frames_per_line, s);
}

int pipe_hip_set_profiling(pipe_hip_processor *p, int32_t enabled)
{
    if (!p)
        return PIPE_HIP_EINVAL;
    p->timer.enable(enabled != 0);
    return PIPE_HIP_OK;
}

int pipe_hip_kernel_time(pipe_hip_processor *p, double *total_ms, int64_t *launches, int32_t reset)
{
    if (!p)
        return PIPE_HIP_EINVAL;
    PH_TRY(p->enter());
    return p->timer.collect(total_ms, launches, reset != 0);
}

const char *pipe_hip_kernel_name(const pipe_hip_processor *p) { return p ? p->last_kernel : ""; }

int pipe_hip_resident_info(pipe_hip_processor *p, int32_t *holds_doorbell, int64_t *dropped_by_watchdog,
                           int64_t *dropped_by_entry)
{
    if (!p)
        return PIPE_HIP_EINVAL;
    if (holds_doorbell)
        *holds_doorbell = p->resident.enabled ? 1 : 0;
    if (dropped_by_watchdog)
        *dropped_by_watchdog = p->resident.dropped_by_watchdog.load();
    if (dropped_by_entry)
        *dropped_by_entry = p->resident.dropped_by_entry.load();
    return PIPE_HIP_OK;
}

int pipe_hip_host_alloc(int64_t bytes, void **ptr)
{
    if (!ptr || bytes < 0)
        return PIPE_HIP_EINVAL;
    *ptr = nullptr;
    PH_HIP(hipHostMalloc(ptr, bytes > 0 ? (size_t)bytes : 16, hipHostMallocDefault));
    return PIPE_HIP_OK;
}

int pipe_hip_host_free(void *ptr)
{
    if (ptr)
        PH_HIP(hipHostFree(ptr));
    return PIPE_HIP_OK;
}

int pipe_hip_synth_fill(int32_t device, void *d_out, int32_t dtype, uint64_t seed, int64_t first_index,
                        int64_t samples, void *stream)
{
    if (!d_out || samples < 0 || (dtype != PIPE_HIP_F32 && dtype != PIPE_HIP_F64))
        return PIPE_HIP_EINVAL;
    PH_HIP(hipSetDevice(device));
    return launch_synth_fill(d_out, dtype, seed, first_index, samples, static_cast<hipStream_t>(stream));
}

}  // extern "C"
