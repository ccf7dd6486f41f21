// Host-side buffer model: the subset of pipelined.dev/signal v0.10.0 (go.mod:3,
// NOT vendored in the reference) that the pipe's hot path touches.  Restated from
// the reference's call sites only:
//   PoolAllocator.Float64 / GetPoolAllocator   pipe.go:394,437,490-492
//   Floating.Length / Slice / Free             pipe.go:401,404-405,431,442,447,464
//   SetSample / FloatingAsFloating / Append    mock/mock.go:100-102,151,185
//   Allocator{Channels,Length,Capacity}.Float64, WriteFloat64, ReadFloat64
//                                              mock/mock_test.go:28-32,119-125,140-141
// Layout: interleaved frames x channels, float64 (what the reference pipe
// allocates for every stage output).  Storage comes from pinned host memory when a
// HIP device is present so that the HIP Processors DMA straight out of pool buffers.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <mutex>
#include <vector>

namespace pipe {
namespace signal {

using Frequency = double;  // signal.Frequency (Hz)

class PoolAllocator;

// backing store shared by every slice of one buffer
struct Storage {
    double *data = nullptr;
    size_t samples = 0;  // capacity in scalar samples
    bool pinned = false;
    std::vector<double> heap;  // used when no pinned memory is available / growable buffers
    ~Storage();
};

// signal.Floating: a (channels, length, capacity) view over shared storage.  Copying
// a Floating copies the view, like a Go slice header.
class Floating {
public:
    Floating() = default;
    bool valid() const { return static_cast<bool>(store_); }
    bool Pinned() const { return store_ && store_->pinned; }  // storage came from the pinned allocator
    int Channels() const { return channels_; }
    int Length() const { return length_; }      // frames
    int Capacity() const { return capacity_; }  // frames
    int Len() const { return length_ * channels_; }  // scalar samples
    double Sample(int i) const { return store_->data[offset_ + (size_t)i]; }
    void SetSample(int i, double v) { store_->data[offset_ + (size_t)i] = v; }
    double *data() { return store_ ? store_->data + offset_ : nullptr; }
    const double *data() const { return store_ ? store_->data + offset_ : nullptr; }
    // Slice(start, end) in frames; shares storage (pipe.go:405,442 use Slice(0, n))
    Floating Slice(int start, int end) const;
    // Append copies src's samples after Length, growing if needed (mock.go:185)
    void Append(const Floating &src);
    // Free returns the buffer to the pool it came from (no-op for foreign buffers)
    void Free(PoolAllocator *pool);

private:
    friend class PoolAllocator;
    friend struct Allocator;
    std::shared_ptr<Storage> store_;
    size_t offset_ = 0;  // scalar samples
    int channels_ = 0, length_ = 0, capacity_ = 0;
};

// signal.Allocator{Channels, Length, Capacity}
struct Allocator {
    int Channels = 0;
    int Length = 0;
    int Capacity = 0;
    Floating Float64() const;  // zero-filled, heap backed (growable)
};

// signal.PoolAllocator: recycles buffers of one geometry.  Thread-safe (async
// mode frees from the downstream goroutine/thread: pipe.go:431).
class PoolAllocator {
public:
    PoolAllocator(int channels, int length, int capacity);
    ~PoolAllocator();
    const int Channels, Length, Capacity;
    Floating Float64();             // Length frames; recycled contents are NOT cleared
    void put(const Floating &f);    // used by Floating::Free
    int64_t allocated() const { return allocated_; }  // buffers ever created

private:
    std::mutex mu_;
    std::vector<std::shared_ptr<Storage>> free_;
    int64_t allocated_ = 0;
};

// signal.GetPoolAllocator(channels, length, capacity)                pipe.go:491
std::shared_ptr<PoolAllocator> GetPoolAllocator(int channels, int length, int capacity);
// buffers created by ALL pools of the process since it started: the steady state of a pipe adds none
int64_t PoolBuffersCreated();

// copies min(src.Len(), dst.Len()) samples; returns FRAMES copied     mock.go:151
int FloatingAsFloating(const Floating &src, Floating &dst);
int WriteFloat64(const std::vector<double> &src, Floating &dst);
int ReadFloat64(const Floating &src, std::vector<double> &dst);

// pinned-memory hooks, installed by the HIP layer when a device is present
using HostAllocFn = int (*)(int64_t bytes, void **ptr);
using HostFreeFn = int (*)(void *ptr);
void SetPinnedAllocator(HostAllocFn alloc, HostFreeFn free);

}  // namespace signal
}  // namespace pipe
