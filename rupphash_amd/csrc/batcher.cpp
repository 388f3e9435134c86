// batcher.cpp -- thread-safe single-image entry point with internal batching.
//
// The reference calls generate_pdq_features once per file from many rayon workers
// (/root/reference/src/scanner.rs:1202-1205, :1410).  One image per GPU call would spend its time in launch and PCIe latency,
// so concurrent callers are coalesced here: the first caller of a batch becomes its leader, waits until the batch is full or
// `max_wait_us` has passed, runs ONE rph_pdq_hash_batch_dev over the pinned staging buffer and wakes the others.
// Callers with a different geometry simply form their own batch.
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <tuple>
#include <vector>

#include "rph_internal.h"

namespace {

// pinned host staging + device buffers of one batch; expensive to create (pinning ~200 MB takes tens of ms), so they are
// pooled in the Batcher and reused by later batches of the same geometry
struct Staging {
    size_t image_bytes = 0;
    uint32_t capacity = 0;
    uint8_t *h_px = nullptr, *h_hash = nullptr, *h_valid = nullptr;
    float *h_q = nullptr, *h_c = nullptr;
    void *d_px = nullptr, *d_hash = nullptr, *d_q = nullptr, *d_c = nullptr, *d_v = nullptr;
    bool alloc(size_t image_bytes_, uint32_t capacity_)
    {
        image_bytes = image_bytes_;
        capacity = capacity_;
        return hipHostMalloc((void **)&h_px, image_bytes * capacity) == hipSuccess && hipHostMalloc((void **)&h_hash, (size_t)capacity * 32) == hipSuccess &&
               hipHostMalloc((void **)&h_q, (size_t)capacity * 4) == hipSuccess && hipHostMalloc((void **)&h_c, (size_t)capacity * 1024) == hipSuccess &&
               hipHostMalloc((void **)&h_valid, capacity) == hipSuccess && hipMalloc(&d_px, image_bytes * capacity) == hipSuccess &&
               hipMalloc(&d_hash, (size_t)capacity * 32) == hipSuccess && hipMalloc(&d_q, (size_t)capacity * 4) == hipSuccess &&
               hipMalloc(&d_c, (size_t)capacity * 1024) == hipSuccess && hipMalloc(&d_v, capacity) == hipSuccess;
    }
    ~Staging()
    {
        for (void *p : {(void *)h_px, (void *)h_hash, (void *)h_valid, (void *)h_q, (void *)h_c})
            if (p) (void)hipHostFree(p);
        for (void *p : {d_px, d_hash, d_q, d_c, d_v})
            if (p) (void)hipFree(p);
    }
};

struct Batch {
    uint32_t w, h, channels;
    uint32_t capacity = 0, count = 0, copying = 0, readers = 0;
    bool closed = false, done = false;
    int status = RPH_OK;
    std::chrono::steady_clock::time_point born;
    std::unique_ptr<Staging> st;
    size_t image_bytes = 0;
    std::condition_variable cv;
};

struct Batcher {
    std::mutex mu;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, std::shared_ptr<Batch>> open;  // batches still accepting images, by geometry
    std::vector<std::unique_ptr<Staging>> pool;                                         // idle staging sets (at most kPoolMax)
    uint32_t max_batch = 256;
    uint32_t max_wait_us = 1000;
    uint64_t n_batches = 0, n_images = 0;
};
constexpr size_t kPoolMax = 4;

std::mutex g_registry_mu;
std::map<rph_ctx *, std::unique_ptr<Batcher>> g_registry;

Batcher &batcher_of(rph_ctx *ctx)
{
    std::lock_guard<std::mutex> lock(g_registry_mu);
    auto &slot = g_registry[ctx];
    if (!slot) slot.reset(new Batcher());
    return *slot;
}

int run_batch(rph_ctx *ctx, Batch &b)
{
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    const uint32_t n = b.count;
    Staging &s = *b.st;
    RPH_HIP_CHECK(hipMemcpyAsync(s.d_px, s.h_px, b.image_bytes * n, hipMemcpyHostToDevice, ctx->stream));
    const int rc = rph_pdq_hash_batch_dev(ctx, s.d_px, n, b.w, b.h, b.channels, (size_t)b.w * b.channels, b.image_bytes, s.d_hash, s.d_q, s.d_c,
                                          nullptr, s.d_v, ctx->stream);
    if (rc != RPH_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    RPH_HIP_CHECK(hipMemcpyAsync(s.h_hash, s.d_hash, (size_t)n * 32, hipMemcpyDeviceToHost, ctx->stream));
    RPH_HIP_CHECK(hipMemcpyAsync(s.h_q, s.d_q, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    RPH_HIP_CHECK(hipMemcpyAsync(s.h_c, s.d_c, (size_t)n * 1024, hipMemcpyDeviceToHost, ctx->stream));
    RPH_HIP_CHECK(hipMemcpyAsync(s.h_valid, s.d_v, n, hipMemcpyDeviceToHost, ctx->stream));
    RPH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RPH_OK;
}

}  // namespace

void rph_batcher_forget(rph_ctx *ctx)
{
    std::lock_guard<std::mutex> lock(g_registry_mu);
    g_registry.erase(ctx);
}

extern "C" int rph_pdq_batcher_config(rph_ctx *ctx, uint32_t max_batch, uint32_t max_wait_us)
{
    return rph_guarded("rph_pdq_batcher_config", [&]() -> int {
        if (!ctx || max_batch == 0 || max_batch > 65536) return RPH_ERR_INVALID_ARG;
        Batcher &B = batcher_of(ctx);
        std::lock_guard<std::mutex> lock(B.mu);
        B.max_batch = max_batch;
        B.max_wait_us = max_wait_us;
        return RPH_OK;
    });
}

extern "C" int rph_pdq_batcher_stats(rph_ctx *ctx, uint64_t *n_batches, uint64_t *n_images)
{
    return rph_guarded("rph_pdq_batcher_stats", [&]() -> int {
        if (!ctx) return RPH_ERR_INVALID_ARG;
        Batcher &B = batcher_of(ctx);
        std::lock_guard<std::mutex> lock(B.mu);
        if (n_batches) *n_batches = B.n_batches;
        if (n_images) *n_images = B.n_images;
        return RPH_OK;
    });
}

extern "C" int rph_pdq_hash_one(rph_ctx *ctx, const uint8_t *px, uint32_t w, uint32_t h, uint32_t channels, size_t row_stride,
                                uint8_t *hash32_out, float *quality_out, float *coeffs_out, uint8_t *valid_out)
{
    return rph_guarded("rph_pdq_hash_one", [&]() -> int {
        if (!ctx || !px || !hash32_out || (channels != 1 && channels != 3 && channels != 4) || row_stride < (size_t)w * channels || w == 0 ||
            h == 0) {
            rph_set_error("rph_pdq_hash_one: invalid argument");
            return RPH_ERR_INVALID_ARG;
        }
        Batcher &B = batcher_of(ctx);
        const auto key = std::make_tuple(w, h, channels);
        std::shared_ptr<Batch> b;
        uint32_t slot;
        bool leader = false;
        {
            std::unique_lock<std::mutex> lock(B.mu);
            auto it = B.open.find(key);
            if (it != B.open.end() && !it->second->closed && it->second->count < it->second->capacity) {
                b = it->second;
            } else {
                b = std::make_shared<Batch>();
                b->w = w;
                b->h = h;
                b->channels = channels;
                b->image_bytes = (size_t)w * h * channels;
                // keep a batch below ~256 MiB of pixels
                const size_t by_bytes = std::max<size_t>(1, ((size_t)256 << 20) / b->image_bytes);
                b->capacity = (uint32_t)std::min<size_t>(B.max_batch, by_bytes);
                b->born = std::chrono::steady_clock::now();
                for (size_t k = 0; k < B.pool.size(); k++)
                    if (B.pool[k]->image_bytes == b->image_bytes && B.pool[k]->capacity == b->capacity) {
                        b->st = std::move(B.pool[k]);
                        B.pool.erase(B.pool.begin() + k);
                        break;
                    }
                if (!b->st) {
                    b->st.reset(new Staging());
                    if (hipSetDevice(ctx->device) != hipSuccess || !b->st->alloc(b->image_bytes, b->capacity)) {
                        rph_set_error("rph_pdq_hash_one: staging allocation failed (%zu bytes x %u)", b->image_bytes, b->capacity);
                        return RPH_ERR_OOM;
                    }
                }
                B.open[key] = b;
                leader = true;
            }
            slot = b->count++;
            b->copying++;
            b->readers++;
            if (b->count == b->capacity) b->cv.notify_all();  // wake the leader: batch is full
        }
        // copy this caller's pixels into its slot (outside the lock: copies of different callers run in parallel)
        {
            uint8_t *dst = b->st->h_px + (size_t)slot * b->image_bytes;
            const size_t line = (size_t)w * channels;
            if (row_stride == line)
                std::memcpy(dst, px, line * h);
            else
                for (uint32_t y = 0; y < h; y++) std::memcpy(dst + (size_t)y * line, px + (size_t)y * row_stride, line);
        }
        {
            std::unique_lock<std::mutex> lock(B.mu);
            b->copying--;
            if (leader) {
                const auto deadline = b->born + std::chrono::microseconds(B.max_wait_us);
                b->cv.wait_until(lock, deadline, [&] { return b->count == b->capacity; });
                b->closed = true;
                auto it = B.open.find(key);
                if (it != B.open.end() && it->second == b) B.open.erase(it);  // later callers start a new batch
                b->cv.wait(lock, [&] { return b->copying == 0; });            // every joined caller has finished copying
                B.n_batches++;
                B.n_images += b->count;
                lock.unlock();
                const int rc = run_batch(ctx, *b);
                lock.lock();
                b->status = rc;
                b->done = true;
                b->cv.notify_all();
            } else {
                b->cv.notify_all();  // the leader may be waiting for copying == 0
                b->cv.wait(lock, [&] { return b->done; });
            }
        }
        const int rc = b->status;
        if (rc == RPH_OK) {
            const Staging &st = *b->st;
            std::memcpy(hash32_out, st.h_hash + (size_t)slot * 32, 32);
            if (quality_out) *quality_out = st.h_q[slot];
            if (coeffs_out) std::memcpy(coeffs_out, st.h_c + (size_t)slot * 256, 1024);
            if (valid_out) *valid_out = st.h_valid[slot];
        }
        {
            std::lock_guard<std::mutex> lock(B.mu);
            if (--b->readers == 0) {  // last caller out: the staging set goes back to the pool
                if (B.pool.size() >= kPoolMax) B.pool.erase(B.pool.begin());
                B.pool.push_back(std::move(b->st));
            }
        }
        return rc;
    });
}
