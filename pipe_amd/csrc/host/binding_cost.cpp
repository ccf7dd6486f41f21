// What a binding costs per ProcessFunc call BEFORE the device is involved: moving one pipe buffer out of a
// signal.Floating into the pinned staging slice and the result back.  TEST/BENCH harness (libpipe_host.so).
//
// The Go shim's first version (integration/go/hip/hip.go, round 1-4) did it one Sample(i) / SetSample(i, v)
// INTERFACE call at a time -- signal.Floating is an interface (pipe.go:62-64), Go does not inline through it
//  -- 16 384 calls per 4096 x 2 buffer.  No Go toolchain here, so the stand-in is the closest thing
// C++ has: a call through a vtable the optimiser cannot see through (the object is made in this translation unit
// behind an opaque factory, the loops are in another function, -fno-devirtualize is not even needed).  The bulk
// alternative is what signal.ReadFloat64 / signal.WriteFloat64 do (mock/mock_test.go:120,128): one copy.
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pipe_host.h"

namespace {

struct FloatingIface {  // the two accessors of signal.Floating the per-sample path uses (mock.go:100-102)
    virtual ~FloatingIface() = default;
    virtual int Len() const = 0;
    virtual double Sample(int i) const = 0;
    virtual void SetSample(int i, double v) = 0;
};
struct Float64Buffer final : FloatingIface {
    std::vector<double> v;
    explicit Float64Buffer(int n) : v((size_t)n) {}
    int Len() const override { return (int)v.size(); }
    double Sample(int i) const override { return v[(size_t)i]; }
    void SetSample(int i, double x) override { v[(size_t)i] = x; }
};
// (opaque to the optimiser: which implementation sits behind the pointer is decided at run time)
__attribute__((noinline)) FloatingIface *make_buffer(int n, int kind)
{
    struct Other final : FloatingIface {
        int n_;
        explicit Other(int n) : n_(n) {}
        int Len() const override { return n_; }
        double Sample(int) const override { return 0.0; }
        void SetSample(int, double) override {}
    };
    if (kind == 1)
        return new Other(n);
    return new Float64Buffer(n);
}
__attribute__((noinline)) void read_per_sample(const FloatingIface *in, double *dst)
{
    const int n = in->Len();
    for (int i = 0; i < n; ++i)
        dst[i] = in->Sample(i);
}
__attribute__((noinline)) void write_per_sample(const double *src, int n, FloatingIface *out)
{
    for (int i = 0; i < n; ++i)
        out->SetSample(i, src[i]);
}
__attribute__((noinline)) void read_per_sample_f32(const FloatingIface *in, float *dst)
{
    const int n = in->Len();
    for (int i = 0; i < n; ++i)
        dst[i] = (float)in->Sample(i);
}
__attribute__((noinline)) void write_per_sample_f32(const float *src, int n, FloatingIface *out)
{
    for (int i = 0; i < n; ++i)
        out->SetSample(i, (double)src[i]);
}
double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int cmp(const void *a, const void *b)
{
    const double x = *static_cast<const double *>(a), y = *static_cast<const double *>(b);
    return x < y ? -1 : x > y;
}

}  // namespace

// median microseconds per buffer (in + out) of frames x channels float64 samples, over `reps` buffers:
//   out_us[0] per-sample interface calls, float64 staging     out_us[1] bulk copies, float64 staging
//   out_us[2] per-sample interface calls, float32 staging     out_us[3] bulk float64 copy + a plain conversion loop, float32 staging
// `staging_in` / `staging_out` (frames x channels doubles each) may be pinned memory of the caller; NULL: heap.
extern "C" int pipe_host_binding_cost(int32_t frames, int32_t channels, int32_t reps, double *staging_in, double *staging_out,
                                      double *out_us)
{
    if (frames < 1 || channels < 1 || reps < 1 || !out_us)
        return 1;
    const int n = frames * channels;
    const char *k = std::getenv("PIPE_HOST_BINDING_KIND");  // (never set: it only keeps make_buffer's choice a run-time one)
    FloatingIface *in = make_buffer(n, k ? std::atoi(k) : 0), *out = make_buffer(n, k ? std::atoi(k) : 0);
    auto *fin = dynamic_cast<Float64Buffer *>(in);
    auto *fout = dynamic_cast<Float64Buffer *>(out);
    if (!fin || !fout)
        return 1;
    std::vector<double> own_in, own_out, lat((size_t)reps);
    if (!staging_in) {
        own_in.resize((size_t)n);
        staging_in = own_in.data();
    }
    if (!staging_out) {
        own_out.resize((size_t)n);
        staging_out = own_out.data();
    }
    float *s32_in = reinterpret_cast<float *>(staging_in), *s32_out = reinterpret_cast<float *>(staging_out);
    std::vector<double> scratch((size_t)n);
    for (int mode = 0; mode < 4; ++mode) {
        for (int r = 0; r < reps; ++r) {
            for (int i = 0; i < n; ++i)  // a new buffer every call (and the caches see what a pipe's caches see)
                fin->v[(size_t)i] = (double)((r * 131 + i) % 1009) * 1e-3;
            const double t0 = now_us();
            switch (mode) {
            case 0:
                read_per_sample(in, staging_in);
                write_per_sample(staging_out, n, out);
                break;
            case 1:  // signal.ReadFloat64(in, staging) / signal.WriteFloat64(staging, out)
                std::memcpy(staging_in, fin->v.data(), sizeof(double) * (size_t)n);
                std::memcpy(fout->v.data(), staging_out, sizeof(double) * (size_t)n);
                break;
            case 2:
                read_per_sample_f32(in, s32_in);
                write_per_sample_f32(s32_out, n, out);
                break;
            default:  // one bulk read into a float64 scratch slice, then plain loops over slices (no interface calls)
                std::memcpy(scratch.data(), fin->v.data(), sizeof(double) * (size_t)n);
                for (int i = 0; i < n; ++i)
                    s32_in[i] = (float)scratch[(size_t)i];
                for (int i = 0; i < n; ++i)
                    scratch[(size_t)i] = (double)s32_out[i];
                std::memcpy(fout->v.data(), scratch.data(), sizeof(double) * (size_t)n);
                break;
            }
            lat[(size_t)r] = now_us() - t0;
        }
        std::qsort(lat.data(), (size_t)reps, sizeof(double), cmp);
        out_us[mode] = lat[(size_t)reps / 2];
    }
    delete in;
    delete out;
    return 0;
}
