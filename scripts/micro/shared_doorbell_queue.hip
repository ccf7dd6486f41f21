// MORE THAN ONE resident handle per device through ONE library-owned doorbell queue?
// Round 5 measured that every PARKED hardware queue costs the one that is rung (9 us alone, 42 us next to one other
// parked queue, 108 next to six): hence one doorbell per device, one handle.  The proposal: keep ONE queue, and let
// every resident handle of the device queue its next buffer's work on it -- wait(bell_i == k) -> kernel -> write(done_i = k) --
// in the order the handles were called last time (the synchronous host loop calls them round-robin, run.go:112-132, so
// the prediction is exact there).  This probe measures what a call then costs, N handles sharing the queue:
//   mode "rr"     : calls come in the predicted order (every call finds its entry at the head of the queue);
//   mode "random" : calls come in random order (the async host loop: a goroutine per component, merger.go:25-30) -- a
//                   call whose entry is NOT at the head rings every entry ahead of it (those run on stale input and are
//                   taken back by their owners: one wasted kernel each), and a handle that finds its entry consumed
//                   queues and rings at once (the plain path's cost) before queueing its successor;
//   "plain"       : launch + completion word + spin on an ordinary stream, nothing parked (what every handle but the
//                   doorbell's holder pays today).
// hipcc --offload-arch=gfx950 -O2 shared_doorbell_queue.hip -o shared_doorbell_queue
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <random>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

// the size of a per-buffer stage: one 4096 x 2 float32 buffer read and written in pinned host memory (zero-copy path)
__global__ void stage(const float *in, float *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = in[i] * 0.5f;
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static bool wait_word(volatile unsigned *w, unsigned k, double limit_us)
{
    const double t0 = now_us();
    while (__atomic_load_n(w, __ATOMIC_ACQUIRE) != k) {
        __builtin_ia32_pause();
        if (now_us() - t0 > limit_us)
            return false;
    }
    return true;
}

struct Handle {
    unsigned *bell, *done;  // coherent pinned words
    float *in, *out;        // pinned staging (device-visible)
    unsigned seq = 0;       // value the queued entry waits for (0: nothing queued)
    bool queued = false, stale = false;
};

int main(int argc, char **argv)
{
    const int calls = argc > 1 ? std::atoi(argv[1]) : 2000;
    const int n = 8192;
    int cus = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    cus = prop.multiProcessorCount;
    std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0xFFFFFFFFu);
    if (cus % 32)
        mask.back() = (1u << (cus % 32)) - 1u;
    hipStream_t q = nullptr, plain = nullptr;
    CK(hipExtStreamCreateWithCUMask(&q, (uint32_t)mask.size(), mask.data()));  // a hardware queue of its own
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    unsigned *words = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void **>(&words), 4096, hipHostMallocCoherent | hipHostMallocMapped));
    std::vector<Handle> H(16);
    for (int i = 0; i < 16; ++i) {
        H[i].bell = words + 32 * i;
        H[i].done = words + 32 * i + 16;
        *H[i].bell = *H[i].done = 0;
        CK(hipHostMalloc(reinterpret_cast<void **>(&H[i].in), sizeof(float) * n, hipHostMallocDefault));
        CK(hipHostMalloc(reinterpret_cast<void **>(&H[i].out), sizeof(float) * n, hipHostMallocDefault));
        for (int j = 0; j < n; ++j)
            H[i].in[j] = (float)j;
    }
    auto arm = [&](Handle &h) {  // queue the handle's next buffer at the tail of the shared queue
        h.seq += 1;
        CK(hipStreamWaitValue32(q, h.bell, h.seq, hipStreamWaitValueEq, 0xFFFFFFFFu));
        hipLaunchKernelGGL(stage, dim3(n / 256), dim3(256), 0, q, h.in, h.out, n);
        CK(hipStreamWriteValue32(q, h.done, h.seq, 0));
        h.queued = true;
        h.stale = false;
    };
    // ---- plain: nothing parked
    {
        std::vector<double> t;
        unsigned k = 0;
        unsigned *pd = words + 1000;
        *pd = 0;
        for (int c = 0; c < calls; ++c) {
            const double t0 = now_us();
            hipLaunchKernelGGL(stage, dim3(n / 256), dim3(256), 0, plain, H[0].in, H[0].out, n);
            CK(hipStreamWriteValue32(plain, pd, ++k, 0));
            if (!wait_word(pd, k, 2e6)) {
                std::printf("plain: timeout\n");
                return 1;
            }
            t.push_back(now_us() - t0);
        }
        std::sort(t.begin(), t.end());
        std::printf("plain (launch + word + spin, nothing parked): median %.1f us, p90 %.1f\n", t[t.size() / 2], t[t.size() * 9 / 10]);
    }
    std::mt19937 rng(7);
    for (const char *mode : {"rr", "random"}) {
        for (int N : {1, 2, 4, 8, 16}) {
            CK(hipStreamSynchronize(q));
            std::deque<int> order;  // handles in queue order
            for (int i = 0; i < N; ++i) {
                H[i].queued = false;
                arm(H[i]);
                order.push_back(i);
            }
            std::vector<double> t;
            long wasted = 0, late = 0;
            for (int c = 0; c < calls; ++c) {
                const int i = mode[1] == 'r' ? c % N : (int)(rng() % (unsigned)N);
                Handle &h = H[i];
                const double t0 = now_us();
                if (h.stale) {
                    // somebody rang our entry: it ran on stale input.  Wait it out (the owner's rollback), queue afresh.
                    if (!wait_word(h.done, h.seq, 2e6)) {
                        std::printf("%s N=%d: timeout on a stale entry\n", mode, N);
                        return 1;
                    }
                    arm(h);
                    order.push_back(i);
                    ++late;
                }
                // everything ahead of our entry is rung (mis-predicted order): wasted kernels
                while (!order.empty() && order.front() != i) {
                    Handle &o = H[order.front()];
                    __atomic_store_n(o.bell, o.seq, __ATOMIC_RELEASE);
                    o.stale = true;
                    order.pop_front();
                    ++wasted;
                }
                order.pop_front();
                const unsigned k = h.seq;
                __atomic_store_n(h.bell, k, __ATOMIC_RELEASE);
                arm(h);  // the successor, queued while this buffer runs
                order.push_back(i);
                if (!wait_word(h.done, k, 2e6)) {
                    std::printf("%s N=%d: timeout\n", mode, N);
                    return 1;
                }
                t.push_back(now_us() - t0);
            }
            // drain: ring whatever is still queued
            for (int i = 0; i < N; ++i)
                __atomic_store_n(H[i].bell, H[i].seq, __ATOMIC_RELEASE);
            CK(hipStreamSynchronize(q));
            std::sort(t.begin(), t.end());
            std::printf("shared queue, %-6s order, %2d handles: per call median %.1f us, p90 %.1f, p99 %.1f; wasted runs %.2f per call, re-queued late %.2f per call\n",
                        mode, N, t[t.size() / 2], t[t.size() * 9 / 10], t[t.size() * 99 / 100], (double)wasted / calls, (double)late / calls);
        }
    }
    return 0;
}
