// Round trips between a host thread and a stream that was armed ahead of time:
//   host stores doorbell = k  ->  hipStreamWaitValue32 (queued earlier) lets the stream go  ->  [kernel]  ->
//   hipStreamWriteValue32 stores done = k  ->  the host, spinning on `done`, sees it.
// against the usual launch + hipStreamSynchronize / hipEventSynchronize of the same kernel.
// hipcc --offload-arch=gfx950 -O2 stream_wait_latency.hip -o stream_wait_latency
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>

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

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? std::atoi(argv[1]) : 2000;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // doorbell: signal memory (what hipStreamWaitValue32 is specified for) or plain coherent pinned memory
    for (int kind = 0; kind < 2; ++kind) {
        unsigned *bell = nullptr, *done = nullptr;
        if (kind == 0) {
            if (hipExtMallocWithFlags((void **)&bell, 8, hipMallocSignalMemory) != hipSuccess) {
                std::printf("signal memory: not available\n");
                (void)hipGetLastError();
                continue;
            }
        } else {
            CK(hipHostMalloc((void **)&bell, 64, hipHostMallocCoherent | hipHostMallocMapped));
        }
        CK(hipHostMalloc((void **)&done, 64, hipHostMallocCoherent | hipHostMallocMapped));
        float *hin, *hout;
        const int n = 8192;  // one 4096 x 2 float32 pipe buffer
        CK(hipHostMalloc((void **)&hin, n * 4, hipHostMallocCoherent | hipHostMallocMapped));
        CK(hipHostMalloc((void **)&hout, n * 4, hipHostMallocCoherent | hipHostMallocMapped));
        for (int i = 0; i < n; ++i)
            hin[i] = (float)i;
        volatile unsigned *vbell = bell, *vdone = done;
        *vbell = 0;
        *vdone = 0;
        for (int with_kernel = 0; with_kernel < 2; ++with_kernel) {
            // warm up + measure: arm k + 1 while k runs (as the product would)
            auto arm = [&](unsigned k) {
                hipError_t e = hipStreamWaitValue32(s, bell, k, hipStreamWaitValueGte, 0xFFFFFFFFu);
                if (e != hipSuccess) {
                    std::printf("kind %d: hipStreamWaitValue32 -> %s\n", kind, hipGetErrorString(e));
                    return false;
                }
                if (with_kernel)
                    hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, s, hin, hout, n);
                CK(hipStreamWriteValue32(s, done, k, 0));
                return true;
            };
            unsigned k = *vdone;
            if (!arm(k + 1))
                break;
            double tot = 0, worst = 0, arm_us = 0;
            for (int r = 0; r < reps + 100; ++r) {
                ++k;
                const double t0 = now_us();
                __atomic_store_n(bell, k, __ATOMIC_RELEASE);
                const double ta = now_us();
                arm(k + 1);  // the next call's work is queued while this one runs
                const double tb = now_us();
                while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != k) {
                }
                const double t1 = now_us();
                if (r >= 100) {
                    tot += t1 - t0;
                    arm_us += tb - ta;
                    worst = t1 - t0 > worst ? t1 - t0 : worst;
                }
            }
            // the same with the arming OFF the timed path (armed two ahead; the next set is queued after `done` is seen,
            // where a helper thread would do it): what the doorbell -> kernel -> done path itself costs
            arm(k + 2);
            double tot2 = 0;
            for (int r = 0; r < reps + 100; ++r) {
                ++k;
                const double t0 = now_us();
                __atomic_store_n(bell, k, __ATOMIC_RELEASE);
                while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != k) {
                }
                const double t1 = now_us();
                if (r >= 100)
                    tot2 += t1 - t0;
                arm(k + 2);
            }
            std::printf("    ... armed ahead of time: %.2f us per round trip\n", tot2 / reps);
            // release the last armed sets
            __atomic_store_n(bell, k + 2, __ATOMIC_RELEASE);
            CK(hipStreamSynchronize(s));
            std::printf("%s doorbell, %s: %.2f us per round trip (worst %.1f), of which arming the next %.2f us\n",
                        kind == 0 ? "signal-memory" : "pinned-coherent", with_kernel ? "with a 32 KiB kernel" : "no kernel",
                        tot / reps, worst, arm_us / reps);
        }
        // baseline: launch + stream synchronise
        {
            double tot = 0;
            for (int r = 0; r < reps + 100; ++r) {
                const double t0 = now_us();
                hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, s, hin, hout, n);
                CK(hipStreamSynchronize(s));
                if (r >= 100)
                    tot += now_us() - t0;
            }
            std::printf("launch + hipStreamSynchronize: %.2f us\n", tot / reps);
            hipEvent_t ev;
            CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            tot = 0;
            for (int r = 0; r < reps + 100; ++r) {
                const double t0 = now_us();
                hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, s, hin, hout, n);
                CK(hipEventRecord(ev, s));
                while (hipEventQuery(ev) == hipErrorNotReady) {
                }
                if (r >= 100)
                    tot += now_us() - t0;
            }
            std::printf("launch + event record + spin on hipEventQuery: %.2f us\n", tot / reps);
        }
    }
    return 0;
}
