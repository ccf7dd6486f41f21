// Does a stream that waits for a doorbell (hipStreamWaitValue32) hold up OTHER streams' work?
// HIP multiplexes streams onto a few hardware (AQL) queues; a wait packet parked in a queue sits in front of
// whatever another stream put into the same queue behind it.  This probe parks N waits, then rings them in
// REVERSE order and measures, per stream, doorbell -> completion word.  A stream whose queue is shared with a
// still-waiting one never answers (timeout).
//   kind 0: hipStreamCreateWithFlags(NonBlocking)              (the pool of GPU_MAX_HW_QUEUES queues)
//   kind 1: hipExtStreamCreateWithCUMask(all CUs)              (is that a queue of its own?)
//   kind 2: hipStreamCreateWithPriority, priorities dealt round-robin over the range
// Also: does work on the null stream / on a fresh non-blocking stream complete while the waits are parked?
// And the cost of a call without any parked wait: launch + hipStreamWriteValue32 + spin on the word, against
// launch + hipStreamSynchronize and against a kernel that stores the completion word itself.
// hipcc --offload-arch=gfx950 -O2 queue_independence.hip -o queue_independence
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

__global__ void touch(const float *in, float *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = in[i] * 0.5f;
}
// the same, and the last workgroup to finish stores the completion word (system scope)
__global__ void touch_flag(const float *in, float *out, int n, unsigned *count, unsigned *done, unsigned k)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = in[i] * 0.5f;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(count, 1u);
        if (t == gridDim.x - 1) {
            *count = 0;
            __atomic_store_n(done, k, __ATOMIC_RELEASE);
        }
    }
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static bool wait_word(volatile unsigned *w, unsigned k, double limit_us)
{
    const double t0 = now_us();
    while (*w != k) {
        __builtin_ia32_pause();
        if (now_us() - t0 > limit_us)
            return false;
    }
    return true;
}

static hipStream_t make_stream(int kind, int i, int cus)
{
    hipStream_t s = nullptr;
    if (kind == 0) {
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    } else if (kind == 1) {
        std::vector<uint32_t> mask((cus + 31) / 32, 0xFFFFFFFFu);
        if (cus % 32)
            mask.back() = (1u << (cus % 32)) - 1u;
        CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    } else {
        int lo = 0, hi = 0;
        CK(hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo = least (numerically largest)
        const int span = lo - hi + 1;
        CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi + (i % span)));
    }
    return s;
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? std::atoi(argv[1]) : 16;
    const int first_part = argc > 2 ? std::atoi(argv[2]) : 1;
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("device: %s, %d CUs, GPU_MAX_HW_QUEUES=%s\n", prop.name, cus,
                std::getenv("GPU_MAX_HW_QUEUES") ? std::getenv("GPU_MAX_HW_QUEUES") : "(unset)");
    unsigned *bell = nullptr, *done = nullptr;
    CK(hipHostMalloc((void **)&bell, 64 * 256, hipHostMallocCoherent | hipHostMallocMapped));
    CK(hipHostMalloc((void **)&done, 64 * 256, hipHostMallocCoherent | hipHostMallocMapped));
    const int n = 8192;
    float *hin, *hout;
    CK(hipHostMalloc((void **)&hin, n * 4, hipHostMallocCoherent | hipHostMallocMapped));
    CK(hipHostMalloc((void **)&hout, n * 4, hipHostMallocCoherent | hipHostMallocMapped));
    for (int i = 0; i < n; ++i)
        hin[i] = (float)i;
    volatile unsigned *vbell = bell, *vdone = done;

    for (int kind = 0; kind < (first_part ? 3 : 0); ++kind) {
        std::vector<hipStream_t> ss(N);
        for (int i = 0; i < N; ++i)
            ss[i] = make_stream(kind, i, cus);
        unsigned flags0 = 0;
        CK(hipStreamGetFlags(ss[0], &flags0));
        for (int round = 1; round <= 3; ++round) {
            for (int i = 0; i < N; ++i) {
                vbell[16 * i] = 0;
                vdone[16 * i] = 0;
            }
            // park: wait, kernel, completion word -- on every stream
            for (int i = 0; i < N; ++i) {
                CK(hipStreamWaitValue32(ss[i], bell + 16 * i, (unsigned)round, hipStreamWaitValueEq, 0xFFFFFFFFu));
                hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, ss[i], hin, hout, n);
                CK(hipStreamWriteValue32(ss[i], done + 16 * i, (unsigned)round, 0));
            }
            // (a) while all are parked: a fresh non-blocking stream and the null stream
            double t_fresh = -1, t_null = -1;
            if (round == 1) {
                hipStream_t f;
                CK(hipStreamCreateWithFlags(&f, hipStreamNonBlocking));
                vdone[16 * 255] = 0;
                double t0 = now_us();
                hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, f, hin, hout, n);
                CK(hipStreamWriteValue32(f, done + 16 * 255, 7u, 0));
                t_fresh = wait_word(vdone + 16 * 255, 7u, 200e3) ? now_us() - t0 : -1;
                vdone[16 * 254] = 0;
                t0 = now_us();
                hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, 0, hin, hout, n);
                CK(hipStreamWriteValue32(0, done + 16 * 254, 7u, 0));
                t_null = wait_word(vdone + 16 * 254, 7u, 200e3) ? now_us() - t0 : -1;
                std::printf("kind %d (flags %#x) N=%d parked: fresh non-blocking stream %s (%.1f us), null stream %s (%.1f us)\n",
                            kind, flags0, N, t_fresh >= 0 ? "ran" : "BLOCKED", t_fresh, t_null >= 0 ? "ran" : "BLOCKED",
                            t_null);
                // (the blocked work stays queued; it drains when the bells ring)
                (void)f;
            }
            // (b) ring in reverse order, one at a time, and wait for that one
            int ok = 0;
            double worst = 0, sum = 0;
            std::vector<int> stuck;
            for (int i = N - 1; i >= 0; --i) {
                const double t0 = now_us();
                __atomic_store_n(bell + 16 * i, (unsigned)round, __ATOMIC_RELEASE);
                if (wait_word(vdone + 16 * i, (unsigned)round, 100e3)) {
                    const double dt = now_us() - t0;
                    ++ok;
                    sum += dt;
                    worst = std::max(worst, dt);
                } else {
                    stuck.push_back(i);
                }
            }
            std::printf("kind %d round %d: %d of %d answered their own doorbell (mean %.1f us, worst %.1f us); stuck:", kind,
                        round, ok, N, ok ? sum / ok : 0.0, worst);
            for (int i : stuck)
                std::printf(" %d", i);
            std::printf("\n");
            // everything has been rung by now: drain
            for (int i = 0; i < N; ++i)
                CK(hipStreamSynchronize(ss[i]));
            CK(hipDeviceSynchronize());
        }
        for (int i = 0; i < N; ++i)
            CK(hipStreamDestroy(ss[i]));
    }

    // ---- what K parked queues cost the others ---------------------------------------------------------------
    // K streams with a hardware queue of their own wait for doorbells that are not rung; measured next to them:
    // (i) ring -> completion of ONE more armed stream of that kind (armed again while it runs, as the product does),
    // (ii) launch + hipStreamWriteValue32 + spin on an ordinary non-blocking stream.
    // (streams with a queue of their own are made ONCE and never destroyed: making one after another one was
    // destroyed hung this probe -- nothing parked, device idle -- in two runs out of three)
    std::vector<hipStream_t> own(13);
    for (int i = 0; i < 13; ++i)
        own[i] = make_stream(1, i, cus);
    for (int K : {0, 1, 2, 3, 4, 6, 8, 12}) {
        std::vector<hipStream_t> parked(own.begin(), own.begin() + K);
        hipStream_t a = own[12], b;
        CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
        std::printf("[K=%d] parking\n", K);
        for (int i = 0; i < K; ++i) {
            vbell[16 * i] = 0;
            CK(hipStreamWaitValue32(parked[i], bell + 16 * i, 1u, hipStreamWaitValueEq, 0xFFFFFFFFu));
            CK(hipStreamWriteValue32(parked[i], done + 16 * i, 1u, 0));
        }
        const int reps = 1500;
        std::vector<double> ta(reps), tb(reps);
        volatile unsigned *abell = vbell + 16 * 200, *adone = vdone + 16 * 200, *bdone = vdone + 16 * 201;
        *abell = 0;
        *adone = 0;
        *bdone = 0;
        std::printf("[K=%d] parked; measuring\n", K);
        auto arm = [&](unsigned k) {
            CK(hipStreamWaitValue32(a, bell + 16 * 200, k, hipStreamWaitValueEq, 0xFFFFFFFFu));
            hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, a, hin, hout, n);
            CK(hipStreamWriteValue32(a, done + 16 * 200, k, 0));
        };
        unsigned k = 0;
        arm(1);
        int lost_a = 0, lost_b = 0;
        for (int r = 0; r < reps + 50 && lost_a < 5 && lost_b < 5; ++r) {
            ++k;
            double t0 = now_us();
            __atomic_store_n(bell + 16 * 200, k, __ATOMIC_RELEASE);
            arm(k + 1);
            if (!wait_word(adone, k, 50e3)) {
                ++lost_a;
                std::printf("  K=%d r=%d: the armed stream did not answer within 50 ms\n", K, r);
                wait_word(adone, k, 2e6);
            }
            if (r >= 50)
                ta[r - 50] = now_us() - t0;
            t0 = now_us();
            hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, b, hin, hout, n);
            CK(hipStreamWriteValue32(b, done + 16 * 201, k, 0));
            if (!wait_word(bdone, k, 50e3)) {
                ++lost_b;
                std::printf("  K=%d r=%d: the ordinary stream did not answer within 50 ms\n", K, r);
                wait_word(bdone, k, 2e6);
            }
            if (r >= 50)
                tb[r - 50] = now_us() - t0;
        }
        std::printf("[K=%d] measured; ringing everything\n", K);
        __atomic_store_n(bell + 16 * 200, k + 1, __ATOMIC_RELEASE);
        for (int i = 0; i < K; ++i)
            __atomic_store_n(bell + 16 * i, 1u, __ATOMIC_RELEASE);
        CK(hipDeviceSynchronize());
        std::printf("[K=%d] device idle\n", K);
        std::sort(ta.begin(), ta.end());
        std::sort(tb.begin(), tb.end());
        std::printf("%2d queues parked: armed stream ring->done median %.2f us (p99 %.2f); ordinary stream launch+word+spin median %.2f us (p99 %.2f)\n",
                    K, ta[reps / 2], ta[reps * 99 / 100], tb[reps / 2], tb[reps * 99 / 100]);
        CK(hipStreamDestroy(b));
    }

    // ---- a call without a parked wait -------------------------------------------------------------------
    {
        hipStream_t s;
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        unsigned *count;
        CK(hipMalloc((void **)&count, 4));
        CK(hipMemset(count, 0, 4));
        const int reps = 3000;
        for (int variant = 0; variant < 4; ++variant) {
            std::vector<double> t(reps);
            unsigned k = 100;
            vdone[0] = 0;
            for (int r = 0; r < reps + 100; ++r) {
                ++k;
                const double t0 = now_us();
                if (variant == 0) {
                    hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, s, hin, hout, n);
                    CK(hipStreamSynchronize(s));
                } else if (variant == 1) {
                    hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, s, hin, hout, n);
                    CK(hipStreamWriteValue32(s, done, k, 0));
                    wait_word(vdone, k, 1e6);
                } else if (variant == 2) {
                    hipLaunchKernelGGL(touch_flag, dim3(n / 256), dim3(256), 0, s, hin, hout, n, count, done, k);
                    wait_word(vdone, k, 1e6);
                } else {
                    // two kernels (a chain of two stages), the second stores the word
                    hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, s, hin, hout, n);
                    hipLaunchKernelGGL(touch_flag, dim3(n / 256), dim3(256), 0, s, hout, hout, n, count, done, k);
                    wait_word(vdone, k, 1e6);
                }
                if (r >= 100)
                    t[r - 100] = now_us() - t0;
            }
            CK(hipStreamSynchronize(s));
            std::sort(t.begin(), t.end());
            const char *names[] = {"launch + hipStreamSynchronize", "launch + hipStreamWriteValue32 + spin",
                                   "kernel stores the word + spin", "two kernels, the second stores the word + spin"};
            std::printf("no parked wait, %-48s median %.2f us, p90 %.2f, p99 %.2f\n", names[variant], t[reps / 2],
                        t[reps * 9 / 10], t[reps * 99 / 100]);
        }
    }
    return 0;
}
