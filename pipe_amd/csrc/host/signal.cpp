#include <atomic>
#include "signal.hpp"

#include <algorithm>
#include <cstring>

namespace pipe {
namespace signal {

namespace {
HostAllocFn g_alloc = nullptr;
HostFreeFn g_free = nullptr;
}  // namespace

void SetPinnedAllocator(HostAllocFn alloc, HostFreeFn free)
{
    g_alloc = alloc;
    g_free = free;
}

Storage::~Storage()
{
    if (pinned && data && g_free)
        g_free(data);
}

static std::shared_ptr<Storage> make_storage(size_t samples, bool try_pinned)
{
    auto s = std::make_shared<Storage>();
    s->samples = samples;
    void *p = nullptr;
    if (try_pinned && g_alloc && samples > 0 && g_alloc((int64_t)(samples * sizeof(double)), &p) == 0 && p) {
        s->data = static_cast<double *>(p);
        s->pinned = true;
    } else {
        s->heap.assign(samples ? samples : 1, 0.0);
        s->data = s->heap.data();
    }
    return s;
}

Floating Floating::Slice(int start, int end) const
{
    Floating f = *this;
    if (start < 0)
        start = 0;
    if (end > capacity_)
        end = capacity_;
    if (end < start)
        end = start;
    f.offset_ = offset_ + (size_t)start * (size_t)channels_;
    f.length_ = end - start;
    f.capacity_ = capacity_ - start;
    return f;
}

void Floating::Append(const Floating &src)
{
    if (!store_) {
        channels_ = src.Channels();
        store_ = make_storage(0, false);
    }
    const size_t need = (size_t)Len() + (size_t)src.Len();
    if (offset_ + need > store_->samples || store_->pinned) {
        // grow like Go's append: new backing array, amortised doubling
        size_t cap = std::max(need, store_->samples * 2);
        auto ns = make_storage(cap, false);
        if (Len())
            std::memcpy(ns->data, store_->data + offset_, sizeof(double) * (size_t)Len());
        store_ = ns;
        offset_ = 0;
        capacity_ = channels_ ? (int)(cap / (size_t)channels_) : 0;
    }
    for (int i = 0; i < src.Len(); ++i)
        store_->data[offset_ + (size_t)Len() + (size_t)i] = src.Sample(i);
    length_ += channels_ ? src.Len() / channels_ : 0;
}

void Floating::Free(PoolAllocator *pool)
{
    if (pool && store_)
        pool->put(*this);
    store_.reset();
    length_ = capacity_ = 0;
}

Floating Allocator::Float64() const
{
    Floating f;
    f.channels_ = Channels;
    f.length_ = Length;
    f.capacity_ = std::max(Capacity, Length);
    f.store_ = make_storage((size_t)f.capacity_ * (size_t)Channels, false);
    return f;
}

static std::atomic<int64_t> g_pool_buffers_created{0};
int64_t PoolBuffersCreated() { return g_pool_buffers_created.load(std::memory_order_relaxed); }

PoolAllocator::PoolAllocator(int channels, int length, int capacity)
    : Channels(channels), Length(length), Capacity(capacity)
{
}

PoolAllocator::~PoolAllocator() = default;

Floating PoolAllocator::Float64()
{
    std::shared_ptr<Storage> s;
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (!free_.empty()) {
            s = std::move(free_.back());
            free_.pop_back();
        } else {
            ++allocated_;
            g_pool_buffers_created.fetch_add(1, std::memory_order_relaxed);
        }
    }
    if (!s)
        s = make_storage((size_t)Capacity * (size_t)Channels, true);
    Floating f;
    f.store_ = std::move(s);
    f.offset_ = 0;
    f.channels_ = Channels;
    f.length_ = Length;
    f.capacity_ = Capacity;
    return f;
}

void PoolAllocator::put(const Floating &f)
{
    // only whole buffers of this geometry come back (a Slice(0,n) still starts at 0)
    if (f.channels_ != Channels || f.offset_ != 0 || !f.store_ ||
        f.store_->samples != (size_t)Capacity * (size_t)Channels)
        return;
    std::lock_guard<std::mutex> lk(mu_);
    free_.push_back(f.store_);
}

std::shared_ptr<PoolAllocator> GetPoolAllocator(int channels, int length, int capacity)
{
    return std::make_shared<PoolAllocator>(channels, length, capacity);
}

int FloatingAsFloating(const Floating &src, Floating &dst)
{
    const int n = std::min(src.Len(), dst.Len());
    for (int i = 0; i < n; ++i)
        dst.SetSample(i, src.Sample(i));
    return src.Channels() ? n / src.Channels() : 0;
}

int WriteFloat64(const std::vector<double> &src, Floating &dst)
{
    const int n = std::min((int)src.size(), dst.Len());
    for (int i = 0; i < n; ++i)
        dst.SetSample(i, src[(size_t)i]);
    return dst.Channels() ? n / dst.Channels() : 0;
}

int ReadFloat64(const Floating &src, std::vector<double> &dst)
{
    const int n = std::min((int)dst.size(), src.Len());
    for (int i = 0; i < n; ++i)
        dst[(size_t)i] = src.Sample(i);
    return src.Channels() ? n / src.Channels() : 0;
}

}  // namespace signal
}  // namespace pipe
